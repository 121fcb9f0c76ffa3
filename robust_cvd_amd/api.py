"""Python host mirror of the optimizer path over the C ABI of libcvd_hip.so (include/cvd_hip.h).

`Solver` has the method surface of the reference's DepthVideoProcessor / DepthVideoPoseOptimizer calls used by
`pose_optimization.py` (reference pose_optimization.py:177-240): reset_depth_xforms, reset_spatial_xforms,
normalize_depth, pose_optimization (= optimizePoses), grid_xform_split.  There is NO fallback: if the HIP
library is missing, or no GPU is usable, construction raises.
"""
import ctypes as C
import os
import sys

from . import build as _build
from .binding import Binding
from .ctypes_types import OptParams, SolveSummary, XformDesc  # noqa: F401 (re-exported)

_lib = None
_variant = None


class SolverOptions(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint64),
        ("pcg_relative_tolerance", C.c_double),
        ("pcg_max_iterations", C.c_int32),
        ("verbose", C.c_int32),
        ("coarse_level", C.c_int32),
        ("robust_loss", C.c_int32),
        ("dense_matrix_free", C.c_int32),
        ("block_inverse_variant", C.c_int32),
        ("coarse_dense_max_unknowns", C.c_int32),
        ("coarse_rebuild_excess", C.c_int32),
        ("coarse_update_budget", C.c_int64),
        ("coarse_dense_shift", C.c_double),
        ("constraint_order", C.c_int32),
        ("coarse_rebuild_excess_dense", C.c_int32),
        ("pcg_fused_tail", C.c_int32),
        ("coarse_dense_row_split", C.c_int32),
        ("dist_owner_update", C.c_int32),
        ("temporal_level", C.c_int32),
        ("temporal_step", C.c_int32),
        ("temporal_grid_x", C.c_int32),
        ("temporal_grid_y", C.c_int32),
        ("coarse_temporal_step", C.c_int32),
        ("coarse_over_budget", C.c_int32),
        ("coarse_temporal_min_frames", C.c_int32),
        ("temporal_weight", C.c_double),
    ]


class DebugOptions(C.Structure):
    """include/cvd_hip_debug.h cvd_debug_options: test / measurement hooks, not part of the product interface."""
    _fields_ = [
        ("struct_size", C.c_uint64),
        ("force_iterations", C.c_int32),
        ("force_sharded_path", C.c_int32),
        ("pcg_lockstep", C.c_int32),
        ("stall_fused_tail_once", C.c_int32),
    ]


ABI_REVISION = 6  # include/cvd_hip.h: CVD_ABI_REVISION


def load_library(variant=None):
    """dlopen the in-tree libcvd_hip.so. Raises ImportError (never falls back) when it is not built.  Nothing is read from the
    environment.  `variant` (development tools only, before anything else loaded the library): a profile build
    lib/libcvd_hip_<variant>.so made by robust_cvd_amd.build.build_variant; the chosen path is reported on stderr."""
    global _lib, _variant
    if _lib is not None and variant:
        raise RuntimeError("load_library(variant=...) must be the first load of the library in the process")
    if _lib is None:
        path = _build.LIB
        if variant:
            path = os.path.join(os.path.dirname(path), f"libcvd_hip_{variant}.so")
            sys.stderr.write(f"[robust_cvd_amd] loading the development variant {path}\n")
        if not os.path.exists(path):
            raise ImportError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). robust_cvd_amd has no CPU fallback.")
        lib = C.CDLL(path)
        lib.cvd_create.restype = C.c_void_p
        lib.cvd_create.argtypes = [C.c_int32]
        lib.cvd_last_error.restype = C.c_char_p
        lib.cvd_last_error.argtypes = [C.c_void_p]
        lib.cvd_num_active_constraints.restype = C.c_int64
        lib.cvd_abi_revision.restype = C.c_int32
        if lib.cvd_abi_revision() != ABI_REVISION:
            raise ImportError(f"{path} was built with ABI revision {lib.cvd_abi_revision()}, this binding is written against "
                              f"{ABI_REVISION} (include/cvd_hip.h: CVD_ABI_REVISION): rebuild the library")
        _lib = lib
        _variant = variant
    return _lib


def loaded_variant():
    """None for the product library, else the development variant this process loaded (`det`: the bit-reproducible build)."""
    return _variant


EXPORTED_SYMBOLS = [
    "cvd_create", "cvd_destroy", "cvd_last_error", "cvd_abi_sizes", "cvd_opt_params_default",
    "cvd_solver_options_default", "cvd_set_solver_options", "cvd_debug_options_default", "cvd_set_debug_options", "cvd_set_generic_kernels", "cvd_comm_unique_id", "cvd_comm_init", "cvd_comm_init_local_group", "cvd_comm_init_phantom", "cvd_set_pair_graph", "cvd_set_video", "cvd_set_depth", "cvd_set_depth_all",
    "cvd_set_pair_constraints", "cvd_set_pair_flows", "cvd_dense_mode_supported", "cvd_set_triplet_constraints", "cvd_set_poses", "cvd_get_poses",
    "cvd_reset_poses", "cvd_reset_depth_xforms", "cvd_reset_spatial_xforms", "cvd_grid_xform_split",
    "cvd_get_xform_desc", "cvd_num_xform_params", "cvd_get_xform_params", "cvd_set_xform_params",
    "cvd_get_pose_params", "cvd_set_pose_params", "cvd_block_size", "cvd_normalize_depth", "cvd_pose_optimization",
    "cvd_pose_optimization_step", "cvd_evaluate", "cvd_sample_pair_constraints", "cvd_get_sampled_constraints", "cvd_sample_triplet_constraints", "cvd_get_sampled_triplet_constraints", "cvd_set_dynamic_masks", "cvd_corner_min_eigenval", "cvd_dynamic_distance", "cvd_apply_depth_xforms", "cvd_depth_param_maps", "cvd_spatial_warp_maps", "cvd_flow_guided_filter", "cvd_get_summary", "cvd_num_records", "cvd_get_records",
    "cvd_get_kernel_times", "cvd_get_comm_times", "cvd_get_dense_times", "cvd_set_kernel_timing", "cvd_num_active_constraints", "cvd_coarse_debug", "cvd_temporal_debug", "cvd_path_info", "cvd_abi_revision",
    "cvd_block_inverse_debug", "cvd_dense_inverse_debug",
]

KERNEL_CLASSES = ["evaluate_assemble", "matvec_pairs", "matvec_finish", "cg_update", "block_inverse", "cost"]


class Solver(Binding):
    def __init__(self, device=0):
        lib = load_library()
        handle = lib.cvd_create(C.c_int32(device))
        if not handle:
            raise RuntimeError("cvd_create failed: " + (lib.cvd_last_error(None) or b"").decode())
        super().__init__(lib, "cvd_", handle)
        self._options = None
        self._debug = None

    def set_options(self, pcg_relative_tolerance=None, pcg_max_iterations=None, verbose=None,
                    coarse_level=None, robust_loss=None, **variants):
        """Options persist per handle: only the fields given change (robust_loss: 0 Cauchy = reference, 1 Huber).  Names of
        cvd_debug_options (force_iterations, force_sharded_path, pcg_lockstep, stall_fused_tail_once: test / measurement hooks,
        include/cvd_hip_debug.h) are routed to cvd_set_debug_options."""
        debug = {k: variants.pop(k) for k in list(variants) if k in dict(DebugOptions._fields_) and k != "struct_size"}
        if debug:
            d = self._debug
            if d is None:
                d = DebugOptions()
                self._lib.cvd_debug_options_default(C.byref(d))
                self._debug = d
            for k, v in debug.items():
                setattr(d, k, int(v))
            self._check(self._fn("set_debug_options")(self._h, C.byref(d)))
        o = self._options
        if o is None:
            o = SolverOptions()
            self._lib.cvd_solver_options_default(C.byref(o))
            self._options = o
        if pcg_relative_tolerance is not None:
            o.pcg_relative_tolerance = pcg_relative_tolerance
        if pcg_max_iterations is not None:
            o.pcg_max_iterations = pcg_max_iterations
        if verbose is not None:
            o.verbose = int(verbose)
        if coarse_level is not None:
            o.coarse_level = int(coarse_level)
        if robust_loss is not None:
            o.robust_loss = int(robust_loss)
        for k, v in variants.items():  # dense_matrix_free, block_inverse_variant, coarse_*, temporal_*, ...
            if k not in dict(SolverOptions._fields_):
                raise TypeError(f"unknown solver option {k!r}")
            setattr(o, k, float(v) if k in ("coarse_dense_shift", "temporal_weight") else int(v))
        self._check(self._fn("set_solver_options")(self._h, C.byref(o)))

    def set_robust_loss(self, kind):
        """0: ceres::CauchyLoss(robustness) (the reference, lib/PoseOptimizer.cpp:1220); 1: ceres::HuberLoss(robustness)."""
        self.set_options(robust_loss=kind)

    @staticmethod
    def comm_unique_id():
        buf = (C.c_uint8 * 128)()
        load_library().cvd_comm_unique_id(buf)
        return bytes(buf)

    def comm_init(self, rank, world, unique_id: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._fn("comm_init")(self._h, C.c_int32(rank), C.c_int32(world), buf))

    def comm_init_local_group(self, rank, world, group_key):
        """Test backend of the exchange layer: `world` handles of this process (one host thread each) form a group."""
        self._check(self._fn("comm_init_local_group")(self._h, C.c_int32(rank), C.c_int32(world), C.c_uint64(group_key)))

    def comm_init_phantom(self, rank, world):
        """Measurement aid (tools/shard_sim.py): rank `rank` of a `world`-rank run whose other ranks do not exist."""
        self._check(self._fn("comm_init_phantom")(self._h, C.c_int32(rank), C.c_int32(world)))

    def set_pair_graph(self, pair_frames):
        """Frame pairs of the whole problem (pair-sharded multi-GPU mode): same array on every rank."""
        import numpy as np
        pf = np.ascontiguousarray(pair_frames, dtype=np.int32).reshape(-1, 2)
        self._check(self._fn("set_pair_graph")(self._h, C.c_int32(pf.shape[0]), pf.ctypes.data_as(C.POINTER(C.c_int32))))

    def set_pair_flows(self, pair_frames, flow, mask):
        """Dense mode (the reference's matchSeparation = 0): flow [P, H, W, 2] f32 pixels and mask [P, H, W] u8 of every
        directed pair instead of a constraint list; the kernels read the images directly."""
        import numpy as np
        pf = np.ascontiguousarray(pair_frames, dtype=np.int32).reshape(-1, 2)
        fl = np.ascontiguousarray(flow, dtype=np.float32)
        mk = np.ascontiguousarray(mask, dtype=np.uint8)
        assert fl.shape == (pf.shape[0], self.height, self.width, 2) and mk.shape == fl.shape[:3], (fl.shape, mk.shape)
        self._check(self._fn("set_pair_flows")(self._h, C.c_int32(pf.shape[0]), pf.ctypes.data_as(C.POINTER(C.c_int32)),
                                               fl.ctypes.data_as(C.POINTER(C.c_float)), mk.ctypes.data_as(C.POINTER(C.c_uint8))))

    def set_generic_kernels(self, enabled=True):
        self._check(self._fn("set_generic_kernels")(self._h, C.c_int32(int(enabled))))

    def set_kernel_timing(self, enabled=True, classes=None, sample_every=1):
        """enabled=True times every kernel class; classes=[names] only those (see KERNEL_CLASSES); sample_every=k attaches
        the hot kernel's start/stop event pair to every k-th launch only (a uniform sample of the launches)."""
        mask = int(bool(enabled))
        if classes is not None:
            mask = 0
            for c in classes:
                mask |= 1 << KERNEL_CLASSES.index(c)
            if mask == 1:
                mask |= 1 << 6  # keep it a mask (bit 0 alone would read as 'all')
        mask |= (max(1, min(256, int(sample_every))) - 1) << 8
        self._check(self._fn("set_kernel_timing")(self._h, C.c_int32(mask)))

    def kernel_times(self):
        ms = (C.c_double * 6)()
        n = (C.c_int64 * 6)()
        self._check(self._fn("get_kernel_times")(self._h, ms, n))
        return {k: {"avg_ms": ms[i], "launches": n[i]} for i, k in enumerate(KERNEL_CLASSES)}

    def dense_times(self):
        """Average ms / launches of the dense mode's two pixel-walking kernels of a Jacobian evaluation (see cvd_get_dense_times)."""
        ms = (C.c_double * 2)()
        n = (C.c_int64 * 2)()
        self._check(self._fn("get_dense_times")(self._h, ms, n))
        return {k: {"avg_ms": ms[i], "launches": n[i]} for i, k in enumerate(("dense_walk", "dense_gg"))}

    def comm_times(self):
        """Average ms / counts of the sharded mode's exchange steps (see cvd_get_comm_times)."""
        ms = (C.c_double * 3)()
        n = (C.c_int64 * 3)()
        self._check(self._fn("get_comm_times")(self._h, ms, n))
        return {k: {"avg_ms": ms[i], "count": n[i]} for i, k in enumerate(("evaluate_exchange", "product_exchange", "coarse_exchange"))}

    def num_active_constraints(self):
        return int(self._lib.cvd_num_active_constraints(self._h))

    def block_inverse_debug(self, blocks, variant=0):
        """f32 inverses of SPD f64 blocks [n, B, B] through the block-Jacobi kernel (variant 0 MFMA blocked sweep = the
        default path, 1 scalar sweep, 2 LDS Cholesky) and the number of failed pivots."""
        import numpy as np
        a = np.ascontiguousarray(blocks, dtype=np.float64)
        n, B, B2 = a.shape
        assert B == B2
        out = np.zeros((n, B, B), dtype=np.float32)
        fl = C.c_int32(0)
        self._check(self._fn("block_inverse_debug")(self._h, C.c_int32(n), C.c_int32(B), a.ctypes.data_as(C.POINTER(C.c_double)),
                                                    C.c_int32(variant), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(fl)))
        return out, fl.value

    def dense_inverse_debug(self, a):
        """f64 inverse of one dense SPD f64 matrix [n, n] through the dense coarse level's kernel (k_dense_spd_inverse) and
        its failure word (1: non-positive pivot, bit 30: barrier timeout)."""
        import numpy as np
        a = np.ascontiguousarray(a, dtype=np.float64)
        n = a.shape[0]
        assert a.shape == (n, n)
        out = np.zeros((n, n), dtype=np.float64)
        fl = C.c_int32(0)
        self._check(self._fn("dense_inverse_debug")(self._h, C.c_int32(n), a.ctypes.data_as(C.POINTER(C.c_double)),
                                                    out.ctypes.data_as(C.POINTER(C.c_double)), C.byref(fl)))
        return out, fl.value

    def temporal_debug(self):
        """Third preconditioner level after the last solve: None when it was off, else its dimensions, Galerkin matrix, the inverse
        in use and the damping vector of the last LM iteration."""
        import numpy as np
        dims = (C.c_int32 * 6)()
        self._check(self._fn("temporal_debug")(self._h, dims, None, None, None, None))
        if dims[0] == 0:
            return None
        n = dims[0]
        a = np.zeros((n, n))
        ai = np.zeros((n, n))
        lam = np.zeros(self.num_frames * self.block_size())
        fl = C.c_int32(0)
        dp = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self._fn("temporal_debug")(self._h, dims, dp(a), dp(ai), dp(lam), C.byref(fl)))
        return {"NT": n, "S": dims[1], "nn": dims[2], "step": dims[3], "Sx": dims[4], "Sy": dims[5], "a_t": a, "a_t_inverse": ai,
                "lam": lam, "failed": fl.value}

    def path_info(self):
        """Which variant of the linear solver the last solve ran (cvd_path_info)."""
        out = (C.c_int32 * 8)()
        self._check(self._fn("path_info")(self._h, out))
        form = {-1: "none", 0: "exact sparse factor", 1: "exact dense inverse", 2: "temporal pose level"}[out[1]]
        return {"pose_graph_level": form, "depth_grid_level": bool(out[2]), "fused_tail": bool(out[3]), "tail_disabled": bool(out[4]),
                "taps": out[5], "work_items": out[6], "cross_blocks": bool(out[7])}

    def coarse_debug(self):
        """(A_c, A_c^-1 as applied, pivot failures) of the coarse preconditioner level after the last solve."""
        import numpy as np
        n = C.c_int32(0)
        self._check(self._fn("coarse_debug")(self._h, C.byref(n), None, None, None))
        if n.value == 0:
            return None
        a = np.zeros((n.value, n.value))
        ai = np.zeros((n.value, n.value))
        fl = C.c_int32(0)
        self._check(self._fn("coarse_debug")(self._h, C.byref(n), a.ctypes.data_as(C.POINTER(C.c_double)),
                                             ai.ctypes.data_as(C.POINTER(C.c_double)), C.byref(fl)))
        return {"a_c": a, "a_c_inverse": ai, "failed": fl.value}
