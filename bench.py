#!/usr/bin/env python3
"""bench.py -- LM iterations/s of the geometric-consistency optimizer on MI355X (driver contract).

Metric (BASELINE.json): Gauss-Newton / Levenberg-Marquardt iterations per second on a 300-frame 384x224
synthetic video (configs[2]: hierarchical flow_list, full LM loop).  One "step" = one LM iteration in Ceres'
counting (Jacobian evaluation + linear solve + candidate-cost evaluation) at the FINAL coarse-to-fine grid
(17x10 bilinear depth grid, 177 unknowns per frame, 53 100 unknowns, ~1.09 M flow constraints), starting
from the state the coarser levels converged to.  Exactly K iterations are timed; they belong to real, naturally
converging solves of that level (when a solve converges early the start state is restored and the next begins).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1 (default `--mode shard`): the SAME 300-frame problem, frame pairs sharded across the ranks
(robust_cvd_amd/sharding.py), regularisers by frame % N; the library all-reduces [g | H_ff | cost] once per
Jacobian evaluation and q once per PCG product over RCCL.  Total work is fixed => "strong" scaling,
`value` = K iterations / max-over-ranks time.  `--mode replicas` instead gives every rank its own video
(no data-path exchange, "weak" scaling, value = N x K / time).

The JSON line also carries
  roofline     : dominant kernel k_matvec_pairs -- algorithmic HBM bytes per launch / average launch duration
                 (HIP events on the solver stream, live in this run) vs the 8 TB/s HBM peak;
  cpu_baseline : the CPU oracle (a port of the reference's Ceres problem: autodiff + exact Cholesky LM)
                 timed on the host cores on a bounded sample, rank 0 at N = 1 only.
"""
import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

FRAMES, WIDTH, HEIGHT = 300, 384, 224
SEED = 1234 + 3
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def prepare(solver, video, params, final_grid=(17, 10), pair_graph=None):
    """Untimed: everything pose_optimization() does before the final coarse-to-fine level."""
    from robust_cvd_amd import synth
    from robust_cvd_amd.ctypes_types import XformDesc
    import torch
    t_up = time.perf_counter()
    synth.load_into(solver, video, params.focal_long)  # the boundary hands over HOST buffers: depth maps + constraints
    torch.cuda.synchronize()
    prepare.upload_seconds = time.perf_counter() - t_up
    if pair_graph is not None:  # pair-sharded mode: the whole problem's frame graph for the coarse preconditioner level
        solver.set_pair_graph(pair_graph)
    solver.reset_depth_xforms(XformDesc.global_depth())
    solver.reset_spatial_xforms(XformDesc.spatial())
    solver.normalize_depth(params)
    # CTF schedule of reference lib/PoseOptimizer.cpp:858-863 for a landscape video: Global -> 6x4 -> 12x7 -> 17x10
    first = True
    for grid in (None, (6, 4), (12, 7)):
        if grid is not None:
            solver.grid_xform_split(XformDesc.grid_depth(*grid))
        solver.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=first)
        first = False
    solver.grid_xform_split(XformDesc.grid_depth(*final_grid))


def cpu_baseline(params, full_constraints):
    """Oracle (kind 'port') on a bounded sample: 32 frames at the same resolution / grid, 6 LM iterations (~10 s of CPU
    work on the GPU box's host cores).  The oracle's linear solve is an exact DENSE Cholesky (cubic in the frame count),
    Ceres' is sparse: the Jacobian-evaluation share, which is a faithful restatement, is reported beside the total."""
    from robust_cvd_amd import synth
    from robust_cvd_amd.ctypes_types import XformDesc
    from oracle.oracle import Oracle
    cores = os.cpu_count() or 1
    threads = min(12, cores)  # reference default numThreads = 12 (lib/PoseOptimizer.h:57)
    sample_frames = 32
    video = synth.make_video(sample_frames, WIDTH, HEIGHT, seed=SEED)
    from robust_cvd_amd.ctypes_types import OptParams
    p = OptParams.defaults()
    p.num_threads = threads
    o = Oracle()
    synth.load_into(o, video, p.focal_long)
    o.reset_depth_xforms(XformDesc.global_depth())
    o.reset_spatial_xforms(XformDesc.spatial())
    o.normalize_depth(p)
    o.grid_xform_split(XformDesc.grid_depth(17, 10))
    p.max_iterations = 6
    t0 = time.perf_counter()
    o.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=True)
    dt = time.perf_counter() - t0
    s = o.summary()
    iters = max(1, s["num_iterations"])
    sample_rate = iters / dt
    scaled = sample_rate * video.num_constraints / float(full_constraints)
    return {
        "value": scaled, "unit": "LM iterations/s", "cores": threads, "kind": "port",
        "sample": (f"oracle (dual-number autodiff + exact dense Cholesky LM) on {sample_frames} frames {WIDTH}x{HEIGHT}, "
                   f"{len(video.pairs)} pairs, {video.num_constraints} constraints, 17x10 grid, {iters} LM iterations in "
                   f"{dt:.2f} s = {sample_rate:.3f} it/s on the sample; scaled linearly in the constraint count to the "
                   f"{full_constraints}-constraint workload (optimistic for the CPU: its solve grows super-linearly)"),
        "sample_it_per_s": sample_rate,
        # the same extrapolation on the residual + Jacobian evaluation time alone (no linear solve at all): an upper bound
        # for any CPU solver built on the reference's autodiff evaluation
        "evaluation_only_it_per_s_scaled": (iters / max(s["evaluate_seconds"], 1e-9)) * video.num_constraints / float(full_constraints),
        "evaluate_seconds": s["evaluate_seconds"], "linear_solve_seconds": s["linear_solve_seconds"],
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES)
    ap.add_argument("--pcg-tol", type=float, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extra-pairs", action="store_true", help="denser pair set (towards the ~4k pairs of BASELINE.json)")
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP-event timing of every kernel class (slower)")
    ap.add_argument("--time-every", type=int, default=4, help="HIP-event pair on every k-th launch of the hot kernel (1 = all)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="development: no HIP-event timing of the hot kernel (roofline fields are then empty)")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard", help="N > 1: pair-sharded (strong) or one video per GPU (weak)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    from robust_cvd_amd import api, synth
    from robust_cvd_amd.ctypes_types import OptParams

    params = OptParams.defaults()
    shard = world > 1 and args.mode == "shard"
    video = synth.make_video(args.frames, WIDTH, HEIGHT, seed=SEED + (0 if shard or world == 1 else rank),
                             extra_offsets=args.extra_pairs)
    solver = api.Solver(local_rank)
    full_pairs, full_constraints = len(video.pairs), video.num_constraints
    if shard:
        # RCCL communicator inside the library: rank 0 mints the id, torch.distributed carries the 128 bytes
        from robust_cvd_amd import sharding
        ids = [api.Solver.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        solver.comm_init(rank, world, ids[0])
        all_pairs = video.pairs.copy()
        mine = sharding.shard_pairs(video.pairs, video.offsets, world)[rank]
        video.pairs, video.offsets, video.loc, video.is_static = sharding.take_pairs(
            video.pairs, video.offsets, video.loc, video.is_static, mine)
    if args.pcg_tol is not None:
        solver.set_options(pcg_relative_tolerance=args.pcg_tol)
    t_prep = time.perf_counter()
    prepare(solver, video, params, pair_graph=all_pairs if shard else None)
    t_prep = time.perf_counter() - t_prep
    prep_summary = solver.summary()

    # State at the start of the final coarse-to-fine level: every measured LM iteration belongs to a real,
    # naturally converging solve from here.  When a solve converges before K iterations are used up, the state is
    # restored and the next solve starts (the restore is two small host->device uploads inside the timed region).
    pose0 = solver.get_pose_params().copy()
    theta0 = solver.get_xform_params().copy()

    def run_iterations(count):
        done, cg, solves, last = 0, 0, 0, None
        while done < count:
            solver.set_pose_params(pose0)
            solver.set_xform_params(theta0)
            params.max_iterations = count - done
            solver.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=False)
            last = solver.summary()
            done += max(1, last["num_iterations"])
            cg += last["total_linear_iterations"]
            solves += 1
        return done, cg, solves, last

    if args.warmup > 0:
        run_iterations(args.warmup)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # HIP-event timing of the dominant kernel only (two event records per timed launch): the other classes are
    # timed in the profiles/ runs, not inside the measured region
    # (default: every 4th launch of the hot kernel carries the start/stop event pair -- a uniform sample of the timed
    # region's launches; the events of hipExtLaunchKernelGGL serialise the dispatch, ~3 % of the rate at every launch)
    sample_every = 1 if args.time_all_kernels else args.time_every
    if not args.no_kernel_timing:
        solver.set_kernel_timing(True, classes=None if args.time_all_kernels else ["matvec_pairs"], sample_every=sample_every)
    barrier()
    t0 = time.perf_counter()
    done, total_cg, n_solves, summ = run_iterations(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    assert done == args.steps, (done, args.steps)
    ktimes = solver.kernel_times()

    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        n_active = solver.num_active_constraints()
        B = solver.block_size()
        mv = ktimes["matvec_pairs"]
        # algorithmic bytes of one k_matvec_pairs launch: the 24 B constraint table entry (ndc 16 B + source
        # depths 8 B) of every constraint + per work item the two frames' x, z, p_old, mask blocks read and
        # the two partial q blocks written (B doubles each).  See DESIGN.md "k_matvec_pairs".
        # work items = undirected frame pairs (both directions share one workgroup), chunked at 768 per direction
        import numpy as np
        cnt = np.diff(video.offsets)
        und = {}
        for (a, b), n in zip(video.pairs.tolist(), cnt.tolist()):
            k = (min(a, b), max(a, b))
            und[k] = max(und.get(k, 0), n)
        n_items = sum(-(-n // 768) for n in und.values())
        bytes_launch = 24.0 * n_active + n_items * (2 * 4 + 2) * B * 8.0
        achieved = (bytes_launch / (mv["avg_ms"] * 1e-3)) / 1e9 if mv["avg_ms"] > 0 else 0.0
        # HBM bytes per launch from the committed PMC run (separate rocprofv3 --pmc passes cannot run inside this
        # process): profiles/pmc_matvec_pairs.json, FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE
        traffic = None
        try:
            with open(os.path.join(_ROOT, "profiles", "pmc_matvec_pairs.json")) as fpm:
                traffic = json.load(fpm)["traffic_bytes_per_launch"] if world == 1 and args.frames == FRAMES else None
        except Exception:
            traffic = None
        out = {
            "metric": "GN/LM iterations/sec (and ms/iter) on 300-frame 384x224 video, 1/2/4/8 GPU",
            "value": (1 if shard else world) * args.steps / dt,
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            # default mode: the SAME problem at every N (pairs sharded) => strong; --mode replicas: one video per GPU => weak
            "scaling": "strong" if args.mode == "shard" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"configs[2]: {args.frames}-frame {WIDTH}x{HEIGHT} synthetic video, hierarchical2 two-way "
                             f"flow_list ({full_pairs} directed pairs, {full_constraints} flow constraints), "
                             f"full LM loop; timed = LM iterations at the final CTF level (17x10 bilinear grid, "
                             f"B={B}, {args.frames * B} unknowns), Cauchy 0.5, PerFrame intrinsics"),
                "pairs": int(full_pairs), "constraints": int(n_active), "unknowns": int(args.frames * B),
                "parallelism": "single-gpu" if world == 1 else (f"pair-sharded dp{world} + RCCL all-reduce" if shard else "video-per-gpu"),
                "linear_solver": "PCG on matrix-free J^T J, two-level preconditioner (per-frame block-Jacobi + pose-graph coarse level)",
                "pcg_iterations_per_lm_iteration": total_cg / max(1, done),
                "solves_in_timed_region": n_solves,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_matvec_pairs", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": bytes_launch, "avg_launch_ms": mv["avg_ms"], "launches": mv["launches"],
                "timed": f"HIP start/stop events on every {sample_every}. launch of the timed region ({mv['launches']} launches timed)",
                "note": "f64 VALU/latency-bound, not HBM-bound: ~24 B and ~1 kflop per constraint (DESIGN.md)",
            },
            "kernels_avg_ms": {k: round(v["avg_ms"], 5) for k, v in ktimes.items()},
            "kernels_launches": {k: v["launches"] for k, v in ktimes.items()},
            "last_timed_solve": {k: summ[k] for k in ("num_iterations", "num_successful_steps", "total_linear_iterations",
                                                      "initial_cost", "final_cost", "termination")},
            "prepare_seconds": t_prep,
            # host -> device hand-over of the inputs (depth maps F*H*W f32 + 16 B per constraint), once per solve sequence;
            # never part of `value` (inputs are resident when the timed region starts)
            "upload_seconds": getattr(prepare, "upload_seconds", None),
            "upload_bytes": int(video.depth.nbytes + video.loc.nbytes + video.is_static.nbytes),
            "prepare_last_level": {k: prep_summary[k] for k in ("num_iterations", "total_linear_iterations", "final_cost",
                                                                "total_seconds")},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, n_active)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
