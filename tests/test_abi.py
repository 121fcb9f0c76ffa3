"""C-ABI library: loads without a GPU, exports every symbol include/cvd_hip.h (the drop-in boundary) and include/cvd_hip_debug.h (test /
measurement hooks) declare, struct sizes agree."""
import ctypes as C
import os
import re

import pytest

from robust_cvd_amd import api
from robust_cvd_amd.ctypes_types import FramePose, IterationRecord, OptParams, SolveSummary, XformDesc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(headers=("cvd_hip.h", "cvd_hip_debug.h")):
    names = set()
    for hname in headers:
        text = open(os.path.join(ROOT, "include", hname)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(cvd_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_header_symbols_are_exported():
    lib = api.load_library()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported"
    assert sorted(api.EXPORTED_SYMBOLS) == names
    # the product header holds no test hook: debug options, simulated ranks and parity hooks of internal solves live in cvd_hip_debug.h
    product = declared_symbols(("cvd_hip.h",))
    for n in ("cvd_set_debug_options", "cvd_comm_init_local_group", "cvd_comm_init_phantom", "cvd_set_generic_kernels",
              "cvd_block_inverse_debug", "cvd_dense_inverse_debug", "cvd_coarse_debug", "cvd_temporal_debug"):
        assert n not in product and n in names
    text = open(os.path.join(ROOT, "include", "cvd_hip.h")).read()
    for hook in ("force_iterations", "force_sharded_path", "pcg_lockstep", "stall_fused_tail_once"):
        assert not re.search(r"int32_t\s+" + hook, text), hook


def test_struct_sizes_match_compiled_library():
    lib = api.load_library()
    out = (C.c_int32 * 6)()
    lib.cvd_abi_sizes(out)
    assert list(out) == [C.sizeof(XformDesc), C.sizeof(OptParams), C.sizeof(FramePose), C.sizeof(IterationRecord),
                         C.sizeof(SolveSummary), C.sizeof(api.SolverOptions)]


def test_default_params_match_reference_defaults():
    """reference lib/PoseOptimizer.h:55-103"""
    lib = api.load_library()
    p = OptParams()
    lib.cvd_opt_params_default(C.byref(p))
    d = OptParams.defaults()
    for name, _ in OptParams._fields_:
        if name == "frame_range":
            continue
        assert getattr(p, name) == getattr(d, name), name
    assert p.max_iterations == 1000 and p.num_threads == 12 and p.num_steps == 4 and p.robustness == 0.5
    assert p.ctf_long == 17 and p.ctf_short == 10 and p.dso_long == 4 and p.dso_short == 3
    assert p.focal_long == 0.3461538376301239 and p.intr_opt == 2 and p.static_loss_type == 1


def test_default_solver_options():
    """The defaults bench.py times and the parity tests run with (include/cvd_hip.h cvd_solver_options); nothing is read from
    the environment, in the library or in the drop-in module."""
    lib = api.load_library()
    o = api.SolverOptions()
    lib.cvd_solver_options_default(C.byref(o))
    assert o.pcg_relative_tolerance == 1e-3 and o.pcg_max_iterations == 300 and o.coarse_level == 1 and o.robust_loss == 0
    assert o.coarse_rebuild_excess == 16 and o.coarse_rebuild_excess_dense == 0   # (0 = the constant 32; -1 = measured on the handle)
    assert o.coarse_dense_max_unknowns == 4096 and o.coarse_update_budget == 40000 and o.coarse_dense_shift == 1e-5
    assert o.constraint_order == 1
    assert (o.temporal_level, o.temporal_step, o.temporal_grid_x, o.temporal_grid_y) == (1, 32, 0, 0)
    assert (o.coarse_temporal_step, o.coarse_over_budget, o.coarse_temporal_min_frames, o.temporal_weight) == (8, 0, 128, 0.7)
    assert (o.dense_matrix_free, o.block_inverse_variant, o.verbose) == (0,) * 3
    d = api.DebugOptions()
    lib.cvd_debug_options_default(C.byref(d))
    assert (d.force_sharded_path, d.pcg_lockstep, d.force_iterations, d.stall_fused_tail_once) == (0,) * 4
    assert d.struct_size == C.sizeof(api.DebugOptions) | (api.ABI_REVISION << 32)
    csrc = os.path.join(ROOT, "robust_cvd_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h", ".cpp")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f"{f} reads the environment"


def test_no_cpu_fallback(have_gpu):
    """Without a GPU the product path must fail loudly (no CPU fallback, no oracle routing)."""
    if have_gpu:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device|no CPU path"):
        api.Solver(0)


def test_product_package_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/ (tier rule 3)."""
    pkg = os.path.join(ROOT, "robust_cvd_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|libcvd_oracle|cvdo_[a-z]+\s*\(|dlopen.*oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f"{f} references the oracle"


@pytest.mark.gpu
def test_solver_options_are_validated():
    """cvd_set_solver_options copies the struct whole: a caller built against another revision of the header (struct_size) and
    values no code path is defined for are refused before anything is stored."""
    s = api.Solver(0)
    lib = api.load_library()
    o = api.SolverOptions()
    lib.cvd_solver_options_default(C.byref(o))
    # sizeof in the low half, the header's revision in the high half (a removed int32 can hide behind the padding)
    assert o.struct_size == C.sizeof(api.SolverOptions) | (api.ABI_REVISION << 32)
    assert lib.cvd_set_solver_options(s._h, C.byref(o)) == 0
    for stale in (o.struct_size - 8, C.sizeof(api.SolverOptions), C.sizeof(api.SolverOptions) | ((api.ABI_REVISION - 1) << 32)):
        o.struct_size = stale
        assert lib.cvd_set_solver_options(s._h, C.byref(o)) != 0
        assert b"struct_size" in lib.cvd_last_error(s._h)
    lib.cvd_solver_options_default(C.byref(o))
    o.pcg_fused_tail = 2
    assert lib.cvd_set_solver_options(s._h, C.byref(o)) != 0   # (the stall hook moved to cvd_debug_options)
    for field, bad in (("pcg_relative_tolerance", 0.0), ("pcg_relative_tolerance", float("nan")), ("pcg_max_iterations", 0),
                       ("coarse_dense_shift", -1.0), ("coarse_rebuild_excess", -1), ("coarse_dense_max_unknowns", 1 << 20),
                       ("coarse_level", 4), ("coarse_temporal_step", 1), ("coarse_over_budget", 2), ("coarse_dense_row_split", 9), ("temporal_level", 3), ("temporal_step", 1), ("temporal_grid_x", 1)):
        lib.cvd_solver_options_default(C.byref(o))
        setattr(o, field, bad)
        assert lib.cvd_set_solver_options(s._h, C.byref(o)) != 0, field
    s.close()
