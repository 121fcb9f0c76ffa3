#!/usr/bin/env python3
"""bench.py -- LM iterations/s of the geometric-consistency optimizer on MI355X (driver contract).

Metric (BASELINE.json): Gauss-Newton / Levenberg-Marquardt iterations per second on a 300-frame 384x224 synthetic
video with ~4k flow pairs, full LM loop (configs[2] as north_star quotes it).  The default workload is the
hierarchical2 flow list of the reference's sampler (utils/frame_sampling.py:77-120) densified to 4140 directed pairs
(`--pairs-level 6`: level l starts every 2^max(0, l-6) frames) = 2.40 M flow constraints; the
reference sampler's own 1766-pair list is timed in the same run as the secondary figure (`secondary_1766_pairs`).

One "step" = one LM iteration in Ceres' counting (Jacobian evaluation + linear solve + candidate-cost evaluation) at
the FINAL coarse-to-fine grid (17x10 bilinear depth grid, 177 unknowns per frame, 53 100 unknowns), starting from the
state the coarser levels converged to, with the DEFAULT solver options -- the ones the parity tests
(tests/test_gpu_baseline_configs.py) run with.  Exactly K iterations are timed; they belong to real, naturally
converging solves of that level (when a solve converges early the start state is restored and the next begins).

    python bench.py --gpus 1 --steps K --warmup W
    python bench.py --config 4 [--robust huber]      # BASELINE configs[4]: 1000 frames 640x384, 16x12 grid (one GPU)
    python bench.py --dense --steps 8 --warmup 1     # configs[2] dense: every masked pixel of 1766 pairs (146 M constraints)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...                     # no launcher: spawns the N ranks itself (re-exec under torch.distributed.run)

N > 1 (default `--mode shard`): the SAME problem, frame pairs sharded across the ranks (robust_cvd_amd/sharding.py),
regularisers by frame % N; the library exchanges over RCCL.  Total work is fixed => "strong" scaling,
`value` = K iterations / max-over-ranks time.  `--mode replicas` gives every rank its own video (weak scaling).
The ranks of the bench itself talk over gloo (barrier, the 128-byte RCCL id, the max over ranks): the library's
communicator is the only RCCL user in the process.

The JSON line also carries
  roofline     : dominant kernel k_matvec_pairs -- algorithmic HBM bytes per launch / average launch duration (HIP
                 events on the solver stream, live in this run) vs the 8 TB/s HBM peak, the same for its counted
                 f64 flops vs the 78.6 TF/s vector peak (`valu`), and the PMC HBM traffic of profiles/pmc_matvec_pairs.json
                 when that file was produced from the kernel sources being benchmarked (hash check);
  cpu_baseline : the CPU oracle (a port of the reference's Ceres problem: autodiff + exact block-sparse Cholesky LM)
                 timed on the host cores on the SAME workload from the SAME state, rank 0 at N = 1 only.
"""
import argparse
import hashlib
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

SEED = 1234 + 3
HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
F64_PEAK_TFLOPS = 78.6  # MI355X f64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz)
# f64 operations per constraint of k_matvec_pairs_fast<4, 256, 1> counted from the COMPILED loop body (FMA = 2): 75 v_mul_f64 +
# 20 v_add_f64 + 93 fused multiply-adds = 281 (round 5, the cross-product form of the rotation derivatives; rounds 2-4: 358 for
# the same product -- the fraction below prices the arithmetic the kernel executes, not the arithmetic an older form needed)
FLOPS_PER_CONSTRAINT = 281.0

CONFIGS = {
    2: dict(frames=300, width=384, height=224, ctf=(17, 10), label="configs[2]"),
    4: dict(frames=1000, width=640, height=384, ctf=(16, 12), label="configs[4]"),
}


def kernel_sources_digest(dense=False):
    """SHA-256 over the headers that define the hot kernel (k_matvec_pairs_fast and the device functions it inlines; dense: + the
    dense mode's walk): ties a committed PMC measurement to the kernel it measured."""
    h = hashlib.sha256()
    d = os.path.join(_ROOT, "robust_cvd_amd", "csrc")
    for fn in ("cvd_device.h", "cvd_kernels.h") + (("cvd_dense_walk.h",) if dense else ()):
        with open(os.path.join(d, fn), "rb") as f:
            h.update(fn.encode())
            h.update(f.read())
    return h.hexdigest()


def ctf_schedule(params, aspect):
    """Grid sizes of the coarse-to-fine levels after the Global one (reference lib/PoseOptimizer.cpp:795-802,858-863)."""
    rows, cols = params.ctf_long, params.ctf_short
    if aspect >= 1.0:
        rows, cols = cols, rows
    out = []
    for step in range(params.num_steps - 1):
        it = (step + 1) / float(params.num_steps - 1)
        out.append((int(1 + (cols - 1) * it + 0.5), int(1 + (rows - 1) * it + 0.5)))
    return out


def prepare(solver, video, params, pair_graph=None):
    """Untimed: everything pose_optimization() does before the final coarse-to-fine level."""
    from robust_cvd_amd import synth
    from robust_cvd_amd.ctypes_types import XformDesc
    import torch
    params.max_iterations = 1000  # (reference default; run_iterations() caps it for the timed solves)
    t_up = time.perf_counter()
    synth.load_into(solver, video, params.focal_long)  # the boundary hands over HOST buffers: depth maps + constraints
    if getattr(video, "dense_flow", None) is not None:     # dense mode: flow / mask images replace the constraint list
        solver.set_pair_flows(video.pairs, video.dense_flow, video.dense_mask)
    torch.cuda.synchronize()
    upload = time.perf_counter() - t_up
    if pair_graph is not None:  # pair-sharded mode: the whole problem's frame graph for the coarse preconditioner level
        solver.set_pair_graph(pair_graph)
    # the whole pipeline once, as pose_optimization.py runs it (normalizeDepth + every coarse-to-fine level): the FIRST run in
    # this process, i.e. including every one-time cost (kernel loading, buffer growth)
    solver.reset_depth_xforms(XformDesc.global_depth())
    solver.reset_spatial_xforms(XformDesc.spatial())
    torch.cuda.synchronize()
    t_pipe = time.perf_counter()
    solver.normalize_depth(params)
    solver.pose_optimization(params)
    pipeline_first = time.perf_counter() - t_pipe
    pipe_summary = solver.summary()
    solver.reset_poses(params.focal_long)
    solver.reset_depth_xforms(XformDesc.global_depth())
    solver.reset_spatial_xforms(XformDesc.spatial())
    solver.normalize_depth(params)
    grids = ctf_schedule(params, video.aspect)
    first = True
    for grid in [None] + grids[:-1]:
        if grid is not None:
            solver.grid_xform_split(XformDesc.grid_depth(*grid))
        solver.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=first)
        first = False
    solver.grid_xform_split(XformDesc.grid_depth(*grids[-1]))
    return upload, grids[-1], {"seconds": pipeline_first, "lm_iterations": pipe_summary["num_iterations"],
                               "pcg_iterations": pipe_summary["total_linear_iterations"], "final_cost": pipe_summary["final_cost"]}


def cpu_baseline(params, video, grid, pose0, theta0, robust, iterations=1):
    """Oracle (kind 'port': dual-number autodiff + Ceres-default LM + exact block-sparse Cholesky on the frame graph) on
    the SAME workload from the SAME state as the timed GPU iterations: one LM iteration, i.e. the initial Jacobian
    evaluation, one factorisation + solve, the candidate cost and (step accepted) the next Jacobian evaluation.  The
    problem construction (one residual-block object per constraint, as the reference builds its Ceres problem in every
    poseOptimizationStep) is reported beside it, not counted."""
    from robust_cvd_amd import synth
    from robust_cvd_amd.ctypes_types import OptParams, XformDesc
    from oracle.oracle import Oracle
    cores = os.cpu_count() or 1
    threads = min(12, cores)  # reference default numThreads = 12 (lib/PoseOptimizer.h:57)
    p = OptParams.defaults()
    for k in ("ctf_long", "ctf_short", "robustness"):
        setattr(p, k, getattr(params, k))
    p.num_threads = threads
    p.max_iterations = iterations
    o = Oracle()
    o.set_robust_loss(robust)
    synth.load_into(o, video, p.focal_long)
    o.reset_depth_xforms(XformDesc.grid_depth(*grid))
    o.reset_spatial_xforms(XformDesc.spatial())
    o.set_pose_params(pose0)
    o.set_xform_params(theta0)
    t0 = time.perf_counter()
    o.pose_optimization_step(p, p.depth_deform_reg_final, convert_poses=False)
    wall = time.perf_counter() - t0
    s = o.summary()
    iters = s["num_iterations"]
    assert iters == iterations, s
    return {
        "value": iters / s["total_seconds"], "unit": "LM iterations/s", "cores": threads, "kind": "port",
        "sample": (f"oracle on the full benchmarked workload ({len(video.pairs)} directed pairs, {video.num_constraints} constraints, "
                   f"{video.num_frames} frames, {grid[0]}x{grid[1]} grid) from the same state as the timed GPU iterations: {iters} LM "
                   f"iteration(s) = {s['total_seconds']:.2f} s ({s['evaluate_seconds']:.2f} s residual + Jacobian evaluation by dual "
                   f"numbers: two Jacobian passes and one cost pass; {s['linear_solve_seconds']:.2f} s exact block-sparse Cholesky "
                   f"step on the frame graph); not counted: {wall - s['total_seconds']:.1f} s problem construction; {threads} "
                   f"threads of {cores} host cores (reference default numThreads = 12); no scaling of any kind"),
        "seconds_per_iteration": s["total_seconds"] / max(1, iters), "evaluate_seconds": s["evaluate_seconds"],
        "linear_solve_seconds": s["linear_solve_seconds"], "problem_construction_seconds": wall - s["total_seconds"],
        # a solver with a free linear solve on top of the reference's autodiff evaluation
        "evaluation_only_it_per_s": iters / max(s["evaluate_seconds"], 1e-9),
        "cost_after_iteration": s["final_cost"],
    }


def run_iterations(solver, params, pose0, theta0, count, records_out=None):
    done, cg, solves, last = 0, 0, 0, None
    while done < count:
        solver.set_pose_params(pose0)
        solver.set_xform_params(theta0)
        params.max_iterations = count - done
        solver.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=False)
        last = solver.summary()
        assert last["num_iterations"] >= 1, last  # (a solve from this start state never terminates at iteration 0)
        if records_out is not None and solves == 0:
            records_out.extend(solver.records())   # (after the timed region's first solve; read outside the hot loop's kernels)
        done += last["num_iterations"]
        cg += last["total_linear_iterations"]
        solves += 1
    return done, cg, solves, last


def matvec_bytes_per_launch(video, n_active, B):
    """Algorithmic bytes of one k_matvec_pairs launch: the 24 B constraint table entry (ndc 16 B + source depths 8 B) of
    every constraint + per work item (undirected frame pair, chunked at 768 constraints per direction) the two frames'
    x, z, p_old, mask blocks read and the two partial q blocks written (B doubles each).  DESIGN.md 3."""
    import numpy as np
    cnt = np.diff(video.offsets)
    und = {}
    for (a, b), n in zip(video.pairs.tolist(), cnt.tolist()):
        k = (min(a, b), max(a, b))
        und[k] = max(und.get(k, 0), n)
    n_items = sum(-(-n // 768) for n in und.values())
    return 24.0 * n_active + n_items * (2 * 4 + 2) * B * 8.0


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU of this node."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n:
        print(f"bench.py: --gpus {n} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(CONFIGS), default=2, help="BASELINE.json configs[k]")
    ap.add_argument("--frames", type=int, default=None, help="development: override the frame count")
    ap.add_argument("--pairs-level", type=int, default=None,
                    help="flow-list density (synth.hierarchical_pairs extra_offsets): 1 = the reference sampler's own list "
                         "(1766 pairs at 300 frames), 6 = the ~4k pairs of north_star (4140, default for --config 2)")
    ap.add_argument("--robust", choices=["cauchy", "huber"], default="cauchy", help="robust loss on the flow constraints")
    ap.add_argument("--dense", action="store_true",
                    help="dense mode (the reference's matchSeparation = 0): flow / mask images of every pair instead of the "
                         "sampled constraint list; the kernels read flow, mask and depth directly, 17 B per pixel pair")
    ap.add_argument("--pcg-tol", type=float, default=None, help="development: PCG forcing value (default: the library's)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="development: any integer field of cvd_solver_options (e.g. coarse_rebuild_excess=8, coarse_level=2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figure on the reference sampler's 1766-pair list")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--time-all-kernels", action="store_true", help="HIP-event timing of every kernel class (slower)")
    ap.add_argument("--time-every", type=int, default=16, help="HIP-event pair on every k-th launch of the hot kernel (1 = all)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="development: no HIP-event timing of the hot kernel (roofline fields are then empty)")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard", help="N > 1: pair-sharded (strong) or one video per GPU (weak)")
    ap.add_argument("--pcg-lockstep", action="store_true", help="profiling: no PCG run-ahead (clean per-launch counter averages)")
    ap.add_argument("--lib-variant", default=None, help="development: load lib/libcvd_hip_<name>.so (robust_cvd_amd.build.build_variant)")
    ap.add_argument("--verify", action="store_true",
                    help="N = 1: the oracle runs the SAME number of LM iterations as the last timed solve from the same state and the "
                         "final costs must agree to 1e-6 (minutes of CPU time)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")

    # (torch first, as every test does: its bundled HIP runtime / librccl.so.1 carry the sonames libcvd_hip.so asks for, so the
    # process holds ONE copy of each -- loading the library first made hipGetDeviceCount fail under the mixed runtimes)
    import torch
    from robust_cvd_amd import api, synth
    from robust_cvd_amd.ctypes_types import OptParams
    if args.lib_variant:
        api.load_library(variant=args.lib_variant)
    if torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} needs device {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    cfg = CONFIGS[args.config]
    frames = args.frames or cfg["frames"]
    width, height = cfg["width"], cfg["height"]
    level = args.pairs_level if args.pairs_level is not None else (6 if args.config == 2 and not args.dense else 1)
    robust = 1 if args.robust == "huber" else 0
    params = OptParams.defaults()
    params.ctf_long, params.ctf_short = cfg["ctf"]
    shard = world > 1 and args.mode == "shard"

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(pairs_level, steps, warmup, timing, seed_offset=0):
        """prepare + warm-up + the timed region for one flow list; returns everything the JSON line needs."""
        video = synth.make_video(frames, width, height, seed=SEED + seed_offset, extra_offsets=pairs_level,
                                 spacing=(1e9 if args.dense else 12.5))  # (dense: the sampled list is not used)
        solver = api.Solver(local_rank)
        if args.dense:
            t_flow = time.perf_counter()
            video.dense_flow, video.dense_mask = synth.make_dense_flows(video)
            print(f"[bench] dense flows of {len(video.pairs)} pairs generated in {time.perf_counter() - t_flow:.1f} s", file=sys.stderr)
        solver.set_options(robust_loss=robust)
        full = dict(pairs=len(video.pairs), constraints=video.num_constraints)
        full_video = video
        all_pairs = None
        if shard:
            # RCCL communicator inside the library: rank 0 mints the id, torch.distributed carries the 128 bytes
            from robust_cvd_amd import sharding
            import copy
            ids = [api.Solver.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            solver.comm_init(rank, world, ids[0])
            all_pairs = video.pairs.copy()
            video = copy.copy(video)
            if args.dense:
                # dense mode shards by pairs as well: every pair is width x height pixel slots, its walk is independent of the others'
                mine = sharding.shard_pairs(full_video.pairs, sharding.uniform_offsets(len(full_video.pairs), width * height), world)[rank]
                _, video.dense_flow, video.dense_mask = sharding.take_pair_flows(
                    full_video.pairs, full_video.dense_flow, full_video.dense_mask, mine)
                video.pairs, video.offsets, video.loc, video.is_static = sharding.take_pairs(   # (the sampled list: not used)
                    full_video.pairs, full_video.offsets, full_video.loc, full_video.is_static, mine)
            else:
                mine = sharding.shard_pairs(video.pairs, video.offsets, world)[rank]
                video.pairs, video.offsets, video.loc, video.is_static = sharding.take_pairs(
                    full_video.pairs, full_video.offsets, full_video.loc, full_video.is_static, mine)
        if args.pcg_tol is not None:
            solver.set_options(pcg_relative_tolerance=args.pcg_tol)
        if args.pcg_lockstep:
            solver.set_options(pcg_lockstep=1)
        for kv in args.opt:
            name, val = kv.split("=")
            solver.set_options(**{name: float(val) if "." in val or "e" in val else int(val)})
        t_prep = time.perf_counter()
        upload, grid, pipeline_first = prepare(solver, video, params, pair_graph=all_pairs)
        t_prep = time.perf_counter() - t_prep
        prep_summary = solver.summary()
        # State at the start of the final coarse-to-fine level: every measured LM iteration belongs to a real, naturally
        # converging solve from here (restoring it is two small host->device uploads inside the timed region).
        pose0 = solver.get_pose_params().copy()
        theta0 = solver.get_xform_params().copy()
        cold = None
        if warmup > 0:
            # (the warm-up is also the COLD solve of this level on this handle: it builds its coarse level in line, whereas the
            # timed solves below start from the one the previous solve left behind -- reported beside the headline)
            torch.cuda.synchronize()
            t_cold = time.perf_counter()
            wdone, wcg, _, _ = run_iterations(solver, params, pose0, theta0, warmup)
            torch.cuda.synchronize()
            cold = {"lm_iterations": wdone, "ms_per_iteration": (time.perf_counter() - t_cold) / max(1, wdone) * 1e3,
                    "pcg_iterations_per_lm_iteration": wcg / max(1, wdone)}
        # HIP-event timing of the dominant kernel only (two event records per timed launch, on every 4th launch: a uniform
        # sample of the timed region; the events of hipExtLaunchKernelGGL serialise the dispatch)
        sample_every = 1 if args.time_all_kernels else args.time_every
        if timing and not args.no_kernel_timing:
            # every class, sampled (every --time-every-th launch of each class): the line carries SURVEY 8(d)'s per-phase split
            classes = None
            solver.set_kernel_timing(True, classes=classes, sample_every=sample_every)
        barrier()
        t0 = time.perf_counter()
        records = []
        done, total_cg, n_solves, summ = run_iterations(solver, params, pose0, theta0, steps, records)
        barrier()
        dt = time.perf_counter() - t0
        assert done == steps, (done, steps)
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        comm = solver.comm_times() if shard else None
        ktimes = solver.kernel_times()
        if args.dense:
            ktimes.update(solver.dense_times())
        solver.set_kernel_timing(False)
        # the whole pipeline again on the warm handle (steady state of a process that optimises video after video)
        from robust_cvd_amd.ctypes_types import XformDesc
        solver.reset_poses(params.focal_long)
        solver.reset_depth_xforms(XformDesc.global_depth())
        solver.reset_spatial_xforms(XformDesc.spatial())
        params.max_iterations = 1000
        barrier()
        t_pipe = time.perf_counter()
        solver.normalize_depth(params)
        solver.pose_optimization(params)
        barrier()
        pipeline_warm = time.perf_counter() - t_pipe
        pipe_sum = solver.summary()
        pipeline = {"first_run_in_process_seconds": pipeline_first["seconds"], "seconds": pipeline_warm,
                    "lm_iterations": pipe_sum["num_iterations"], "pcg_iterations": pipe_sum["total_linear_iterations"],
                    "final_cost": pipe_sum["final_cost"],
                    "what": "normalizeDepth + poseOptimization (all coarse-to-fine levels) on the benchmarked video, inputs resident"}
        equiv = None
        if shard and timing:
            # N = 1 equivalence: one full solve of the final level from the common start state, sharded over all ranks and,
            # on rank 0, unsharded on its own GPU -- the sharded mode must reproduce the single-GPU result
            solver.set_pose_params(pose0)
            solver.set_xform_params(theta0)
            params.max_iterations = 1000
            solver.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=False)
            sh_sum, sh_pose, sh_theta = solver.summary(), solver.get_poses(), solver.get_xform_params().copy()
            if rank == 0:
                import numpy as np
                single = api.Solver(local_rank)
                single.set_options(robust_loss=robust)
                if args.pcg_tol is not None:
                    single.set_options(pcg_relative_tolerance=args.pcg_tol)
                synth.load_into(single, full_video, params.focal_long)
                if args.dense:
                    single.set_pair_flows(full_video.pairs, full_video.dense_flow, full_video.dense_mask)
                from robust_cvd_amd.ctypes_types import XformDesc
                single.reset_depth_xforms(XformDesc.grid_depth(*grid))
                single.reset_spatial_xforms(XformDesc.spatial())
                single.set_pose_params(pose0)
                single.set_xform_params(theta0)
                single.pose_optimization_step(params, params.depth_deform_reg_final, convert_poses=False)
                s1, p1 = single.summary(), single.get_poses()
                perr, rerr = synth.relative_pose_error(sh_pose["position"], sh_pose["orientation"], p1["position"], p1["orientation"])
                equiv = {"final_cost_sharded": sh_sum["final_cost"], "final_cost_single_gpu": s1["final_cost"],
                         "final_cost_rel_diff": abs(sh_sum["final_cost"] - s1["final_cost"]) / abs(s1["final_cost"]),
                         "lm_iterations": [sh_sum["num_iterations"], s1["num_iterations"]],
                         "pose_err": perr, "rot_err": rerr,
                         "theta_rel_diff": float(np.abs(sh_theta - single.get_xform_params()).max() / np.abs(sh_theta).max())}
                single.close()
        return dict(video=full_video, local_video=video, solver=solver, full=full, dt=dt, total_cg=total_cg, n_solves=n_solves, summ=summ, cold=cold,
                    pipeline=pipeline, records=records,
                    comm=comm, equivalence=equiv,
                    t_prep=t_prep, upload=upload, prep_summary=prep_summary, pose0=pose0, theta0=theta0, grid=grid,
                    sample_every=sample_every, ktimes=ktimes, n_active=solver.num_active_constraints(),
                    B=solver.block_size())

    m = measure(level, args.steps, args.warmup, timing=True, seed_offset=(0 if shard or world == 1 else rank))

    if rank == 0:
        video, B, n_active = m["video"], m["B"], m["n_active"]
        mv = m["ktimes"]["matvec_pairs"]
        dense_explicit = args.dense and not any(kv.replace(" ", "") == "dense_matrix_free=1" for kv in args.opt)
        if args.dense:
            npx = width * height
            slots = len(video.pairs) * npx
            und = {(min(a, b), max(a, b)) for a, b in video.pairs.tolist()}
            if dense_explicit:
                # The dominant kernel of a dense-mode step is the ONE walk over the pixels of the Jacobian evaluation
                # (k_dense_walk, cvd_dense_walk.h): SURVEY.md 8d's 17 B per pixel pair -- flow 8 B + mask 1 B + d_src0 4 B +
                # gathered d_src1 4 B, every slot of every directed pair read once -- + the 8 B per slot it leaves for the
                # grid x grid kernel + one record per directed pair.  (The PCG product streams the assembled blocks: dense_kernels.)
                mv = m["ktimes"]["dense_walk"]
                G = B - 7
                bytes_launch = 17.0 * slots + 8.0 * slots + len(video.pairs) * (256 + 40 * G + 8) * 8.0
            else:
                # flow 8 B + mask 1 B + d_src0 4 B + gathered d_src1 4 B = 17 B per pixel pair (every pixel slot
                # of every pair is read) + per work item (8192 slots per direction) the frame blocks as in the list mode
                bytes_launch = 17.0 * slots + len(und) * (-(-npx // 8192)) * (2 * 4 + 2) * B * 8.0
        else:
            bytes_launch = matvec_bytes_per_launch(m["local_video"], n_active, B)
        achieved = (bytes_launch / (mv["avg_ms"] * 1e-3)) / 1e9 if mv["avg_ms"] > 0 else 0.0
        # (explicit blocks: y_a = X p_b and y_b = X^T p_a, 4 B^2 flop per pair)
        # dense walk: ~780 f64 VALU flop per pixel constraint counted from the compiled loop (mul 209 + add 102 + 2 x 172 fma, the
        # row loop three times), beside 48 v_mfma_f64_16x16x4 per 64 constraints on the matrix pipe (1536 flop per constraint issued)
        DENSE_WALK_FLOPS = 780.0
        flops_launch = (DENSE_WALK_FLOPS * n_active) if (args.dense and dense_explicit) else FLOPS_PER_CONSTRAINT * n_active
        tflops = (flops_launch / (mv["avg_ms"] * 1e-3)) / 1e12 if mv["avg_ms"] > 0 else 0.0
        # HBM bytes per launch from the committed PMC passes (separate rocprofv3 --pmc runs cannot happen inside this
        # process): only when profiles/pmc_matvec_pairs.json was produced from the kernel sources benchmarked here and
        # on this workload; FETCH_SIZE doubled per the gfx950 correction, + WRITE_SIZE (tools/pmc_to_json.py)
        traffic, traffic_note = None, "no PMC file"
        pmc_name = "profiles/pmc_dense_walk.json" if (args.dense and dense_explicit) else "profiles/pmc_matvec_pairs.json"
        try:
            with open(os.path.join(_ROOT, pmc_name)) as fpm:
                pmc = json.load(fpm)
            if pmc.get("kernel_sources_sha256") != kernel_sources_digest(dense=pmc_name.endswith("pmc_dense_walk.json")):
                traffic_note = f"{pmc_name} is stale (kernel sources changed since it was measured)"
            elif world != 1 or pmc.get("constraints") != int(n_active):
                traffic_note = f"{pmc_name} was measured on another workload"
            else:
                traffic, traffic_note = pmc["traffic_bytes_per_launch"], f"{pmc_name} (same kernel sources, same workload)"
        except Exception:
            pass
        out = {
            "metric": "GN/LM iterations/sec (and ms/iter) on 300-frame 384x224 video, 1/2/4/8 GPU",
            "value": (1 if shard else world) * args.steps / m["dt"],
            "unit": "LM iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["dt"] / args.steps * 1e3,
            "higher_is_better": True,
            # default mode: the SAME problem at every N (pairs sharded) => strong; --mode replicas: one video per GPU => weak
            "scaling": "strong" if args.mode == "shard" else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (f"{cfg['label']}{' dense' if args.dense else ''}: {frames}-frame {width}x{height} synthetic video, hierarchical2 two-way flow_list "
                             f"densified to level {level} ({m['full']['pairs']} directed pairs, "
                             + (f"DENSE mode: every masked pixel is a constraint, {n_active} flow constraints read from the flow / mask / "
                                f"depth images" if args.dense else f"{m['full']['constraints']} flow constraints") + f"), full LM loop; timed = LM iterations at the final CTF level ({m['grid'][0]}x{m['grid'][1]} "
                             f"bilinear grid, B={B}, {frames * B} unknowns), {'Huber' if robust else 'Cauchy'} {params.robustness}, "
                             f"PerFrame intrinsics, default solver options"),
                "pairs": int(m["full"]["pairs"]), "constraints": int(n_active), "unknowns": int(frames * B),
                "parallelism": "single-gpu" if world == 1 else (f"pair-sharded dp{world} + RCCL" if shard else "video-per-gpu"),
                "linear_solver": ("PCG on matrix-free J^T J, additive three-level preconditioner: per-frame block-Jacobi + pose-graph level (8 modes per frame; exact "
                             "sparse factor or, for this pair graph, its temporally coarse form) + temporally coarse depth-grid level"),
                "pcg_iterations_per_lm_iteration": m["total_cg"] / max(1, args.steps),
                "solves_in_timed_region": m["n_solves"],
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_dense_walk" if (args.dense and dense_explicit) else "k_matvec_pairs", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                "bytes_per_launch": bytes_launch, "avg_launch_ms": mv["avg_ms"], "launches": mv["launches"],
                "timed": f"HIP start/stop events on every {m['sample_every']}. launch of the timed region ({mv['launches']} launches timed)",
                "valu": {"achieved": tflops, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_PEAK_TFLOPS,
                         "flops_per_launch": flops_launch,
                         "note": (f"{DENSE_WALK_FLOPS:.0f} f64 VALU flop per pixel constraint counted from the compiled loop; the pose Gram "
                                  "tile runs beside it on the matrix pipe (48 v_mfma_f64_16x16x4 per 64 constraints)" if (args.dense and dense_explicit) else
                                  f"{FLOPS_PER_CONSTRAINT:.0f} f64 flop per constraint counted from the kernel source (DESIGN.md 3)")},
                "note": ("dense mode: the Jacobian evaluation's one walk over the pixels (flow / mask / depth read once per evaluation); "
                         "f64 VALU / matrix-pipe / LDS-atomic bound, not HBM-bound: both fractions are reported; the PCG product of "
                         "this mode streams the assembled cross blocks (dense_kernels.product)" if (args.dense and dense_explicit)
                         else "f64 VALU/latency-bound, not HBM-bound (DESIGN.md 3): both fractions are reported"),
            },
            **({"dense_kernels": {
                # the image-reading kernels, 17 B per pixel slot and pass (the cost class also holds the small per-frame
                # regulariser kernels; the assemble class reads every slot twice for the frame-diagonal blocks -- once per side --
                # and, with explicit cross blocks, 1 + panels more times for X_ab: reported against the two diagonal passes)
                "cost": {"avg_ms": m["ktimes"]["cost"]["avg_ms"], "GB/s": 17.0 * slots / max(m["ktimes"]["cost"]["avg_ms"], 1e-9) * 1e-6,
                         "frac_hbm": 17.0 * slots / max(m["ktimes"]["cost"]["avg_ms"], 1e-9) * 1e-6 / HBM_PEAK_GBS},
                # the whole Jacobian evaluation: walk + per-frame fold + cross-block fold + grid x grid kernel (+ the small launches)
                "assemble": {"avg_ms": m["ktimes"]["evaluate_assemble"]["avg_ms"],
                             "GB/s": 17.0 * slots / max(m["ktimes"]["evaluate_assemble"]["avg_ms"], 1e-9) * 1e-6,
                             "frac_hbm": 17.0 * slots / max(m["ktimes"]["evaluate_assemble"]["avg_ms"], 1e-9) * 1e-6 / HBM_PEAK_GBS,
                             "TFLOP/s": 780.0 * n_active / max(m["ktimes"]["evaluate_assemble"]["avg_ms"], 1e-9) * 1e-9,
                             "frac_valu": 780.0 * n_active / max(m["ktimes"]["evaluate_assemble"]["avg_ms"], 1e-9) * 1e-9 / F64_PEAK_TFLOPS,
                             "note": "rounds 2-5: three walks, ~1600 flop per constraint (the chain re-derived per side and per cross block), "
                                     "38.2 ms; round 6: one walk, ~780 VALU flop per constraint + the Gram tile on the matrix pipe"},
                **({"walk": {"avg_ms": m["ktimes"]["dense_walk"]["avg_ms"], "launches": m["ktimes"]["dense_walk"]["launches"]},
                    "grid_x_grid": {"avg_ms": m["ktimes"]["dense_gg"]["avg_ms"], "launches": m["ktimes"]["dense_gg"]["launches"],
                                    "GB/s": 16.0 * slots / max(m["ktimes"]["dense_gg"]["avg_ms"], 1e-9) * 1e-6,
                                    "note": "8 B scalar + 8 B flow per pixel slot, read once: panels of SOURCE vertices, one launch per "
                                            "direction (the pixels of the one cell row two panels share are read twice: + 1/9 at 17 x 10)"},
                    # (one block per undirected pair + -- one GPU -- the frames' own blocks H_ff, streamed by the same kernel since round 6)
                    "product": {"kernel": "k_cross_matvec", "avg_ms": m["ktimes"]["matvec_pairs"]["avg_ms"],
                                "GB/s": (len(und) + (frames if world == 1 else 0)) * (B * B * 8.0 + (2 * 3 + 2) * B * 8.0) / max(m["ktimes"]["matvec_pairs"]["avg_ms"], 1e-9) * 1e-6,
                                "frac_hbm": (len(und) + (frames if world == 1 else 0)) * (B * B * 8.0 + (2 * 3 + 2) * B * 8.0) / max(m["ktimes"]["matvec_pairs"]["avg_ms"], 1e-9) * 1e-6 / HBM_PEAK_GBS}}
                   if dense_explicit else {}),
            }} if args.dense else {}),
            **({"rccl": {"ranks": world, "frames_owned_per_rank": -(-frames // world),
                         "exchange_avg_ms": {k: round(v["avg_ms"], 5) for k, v in m["comm"].items()},
                         "exchange_counts": {k: v["count"] for k, v in m["comm"].items()},
                         "n1_equivalence": m["equivalence"],
                         "note": "per Jacobian evaluation: all-reduce g + cost, reduce-scatter H_ff to the frames' owners, all-gather "
                                 "diag(H) and the f32 block inverses; per PCG iteration (owner-sharded update, the default): reduce-scatter q + "
                                 "all-reduce [Z^T q | p.q] after the product, all-gather z / c / r^T z shares after the update -- two "
                                 "grouped collectives (rank 0's timings)"}} if shard else {}),
            "kernels_avg_ms": {k: round(v["avg_ms"], 5) for k, v in m["ktimes"].items()},
            "kernels_launches": {k: v["launches"] for k, v in m["ktimes"].items()},
            "kernels_note": ("HIP events on a uniform sample of every class's launches (every --time-every-th); matvec_pairs: the "
                             "kernel's own start / stop stamps, the other classes: event pairs around the launch (dispatch gap "
                             "included).  With the fused PCG tail (k_pcg_tail: finish + update in one launch, the default where its "
                             "scope allows) the finish class is empty and cg_update is the fused kernel; block_inverse includes the "
                             "in-line rebuilds of the coarse level; SURVEY 8(d)'s split per LM iteration = evaluate_assemble + "
                             "block_inverse + pcg_iterations_per_lm_iteration x (matvec_pairs + matvec_finish + cg_update) + cost"),
            "last_timed_solve": {k: m["summ"][k] for k in ("num_iterations", "num_successful_steps", "total_linear_iterations",
                                                           "initial_cost", "final_cost", "termination")},
            # every solve builds its coarse level in line (one persistent kernel) and carries nothing over from the solve before:
            # the timed solves ARE cold solves; the warm-up solve additionally pays first-launch costs of the final level's kernels
            "cold_first_solve": ({**m["cold"], "note": "the warm-up solve = the first solve of the final level on this handle"}
                                 if m["cold"] else None),
            "pipeline": m["pipeline"],
            "prepare_seconds": m["t_prep"],
            # host -> device hand-over of the inputs (depth maps F*H*W f32 + 16 B per constraint), once per solve sequence;
            # never part of `value` (inputs are resident when the timed region starts)
            "upload_seconds": m["upload"],
            "upload_bytes": int(video.depth.nbytes + video.loc.nbytes + video.is_static.nbytes),
            "prepare_last_level": {k: m["prep_summary"][k] for k in ("num_iterations", "total_linear_iterations", "final_cost",
                                                                     "total_seconds")},
        }
    solver_main = m.pop("solver")
    solver_main.close()
    if args.config == 2 and level != 1 and not args.no_secondary and args.frames is None and not args.dense and world == 1:
        # secondary figure: the reference sampler's own flow list (1766 directed pairs at 300 frames), same recipe (single-GPU
        # runs only: a multi-rank run times the headline workload and nothing else)
        m2 = measure(1, args.secondary_steps, min(args.warmup, 2), timing=False)
        if rank == 0:
            out["secondary_1766_pairs"] = {
                "value": (1 if shard else world) * args.secondary_steps / m2["dt"], "unit": "LM iterations/s",
                "ms_per_step": m2["dt"] / args.secondary_steps * 1e3, "steps": args.secondary_steps,
                "pairs": int(m2["full"]["pairs"]), "constraints": int(m2["full"]["constraints"]),
                "pcg_iterations_per_lm_iteration": m2["total_cg"] / max(1, args.secondary_steps),
                "workload": "the reference sampler's hierarchical2 flow list (utils/frame_sampling.py:77-120), otherwise identical",
            }
        m2.pop("solver").close()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.dense:
            cb = cpu_baseline(params, m["video"], m["grid"], m["pose0"], m["theta0"], robust)
            # results, not only speed: the oracle's cost after ITS first LM iteration (exact sparse Cholesky step) against the
            # GPU's after the first iteration of the first timed solve (PCG step to eta), same start state
            recs = m["records"]
            if len(recs) >= 2:
                g1 = recs[1]["cost"]
                cb["gpu_cost_after_iteration"] = g1
                cb["cost_rel_diff_after_iteration"] = abs(g1 - cb["cost_after_iteration"]) / abs(cb["cost_after_iteration"])
                if cb["cost_rel_diff_after_iteration"] > 1e-4:
                    print(f"[bench] WARNING: GPU and oracle disagree after one LM iteration: {g1} vs {cb['cost_after_iteration']}", file=sys.stderr)
            out["cpu_baseline"] = cb
            if args.verify:
                k = m["summ"]["num_iterations"]
                cv = cpu_baseline(params, m["video"], m["grid"], m["pose0"], m["theta0"], robust, iterations=k)
                rel = abs(cv["cost_after_iteration"] - m["summ"]["final_cost"]) / abs(cv["cost_after_iteration"])
                out["verify"] = {"lm_iterations": k, "oracle_final_cost": cv["cost_after_iteration"], "gpu_final_cost": m["summ"]["final_cost"],
                                 "rel_diff": rel, "oracle_seconds": cv["seconds_per_iteration"] * k}
                assert rel <= 1e-6, out["verify"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
