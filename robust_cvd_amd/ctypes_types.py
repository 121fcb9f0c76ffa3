"""ctypes mirrors of include/cvd_types.h (one definition, used by the product binding and by tests).

Field order and widths must match the header exactly; tests/test_abi.py checks sizeof() against the
values the compiled library reports.
"""
import ctypes as C
import enum


class ValueXformType(enum.IntEnum):  # reference lib/ValueTransform.h:16-20
    NONE = 0
    Scale = 1
    ScaleShift = 2


class XformType(enum.IntEnum):  # reference lib/DepthMapTransform.h:25-28
    Depth = 0
    Spatial = 1


class DepthXformType(enum.IntEnum):  # reference lib/DepthMapTransform.h:31-36
    NONE = 0
    Identity = 1
    Global = 2
    Grid = 3


class SpatialXformType(enum.IntEnum):  # reference lib/DepthMapTransform.h:39-46
    NONE = 0
    Identity = 1
    VerticalLinear = 2
    CornersBilinear = 3
    BilinearGrid = 4
    BicubicGrid = 5


class StaticLossType(enum.IntEnum):  # reference lib/PoseOptimizer.h:22-27
    Euclidean = 0
    ReproDisparity = 1
    ReproDepthRatio = 2
    ReproLogDepth = 3


class SmoothLossType(enum.IntEnum):  # reference lib/PoseOptimizer.h:37-42
    EuclideanLaplacian = 0
    ReproDisparityLaplacian = 1
    ReproDepthRatioConsistency = 2
    ReproLogDepthConsistency = 3


class IntrinsicsOptimization(enum.IntEnum):  # reference lib/PoseOptimizer.h:46-50
    Fixed = 0
    Shared = 1
    PerFrame = 2


class XformDesc(C.Structure):
    _fields_ = [
        ("type", C.c_int32),
        ("depth_type", C.c_int32),
        ("spatial_type", C.c_int32),
        ("value_xform", C.c_int32),
        ("cubic_interpolation", C.c_int32),
        ("grid_size", C.c_int32 * 3),
        ("depth_min_max", C.c_double * 2),
    ]

    @staticmethod
    def identity_depth():
        return XformDesc(type=XformType.Depth, depth_type=DepthXformType.Identity)

    @staticmethod
    def global_depth(value=ValueXformType.Scale):
        return XformDesc(type=XformType.Depth, depth_type=DepthXformType.Global, value_xform=value)

    @staticmethod
    def grid_depth(cols, rows, value=ValueXformType.Scale, cubic=False, depth=1, dmin=0.0, dmax=0.0):
        d = XformDesc(type=XformType.Depth, depth_type=DepthXformType.Grid, value_xform=value,
                      cubic_interpolation=int(cubic))
        d.grid_size[0], d.grid_size[1], d.grid_size[2] = cols, rows, depth
        d.depth_min_max[0], d.depth_min_max[1] = dmin, dmax
        return d

    @staticmethod
    def spatial(kind=SpatialXformType.Identity, cols=0, rows=0):
        d = XformDesc(type=XformType.Spatial, depth_type=DepthXformType.NONE, spatial_type=kind)
        d.grid_size[0], d.grid_size[1], d.grid_size[2] = cols, rows, 0
        return d

    def copy(self):
        d = XformDesc()
        C.memmove(C.byref(d), C.byref(self), C.sizeof(XformDesc))
        return d


class OptParams(C.Structure):
    """DepthVideoPoseOptimizer::Params (reference lib/PoseOptimizer.h:54-108), defaults identical."""
    _fields_ = [
        ("frame_range", C.POINTER(C.c_int32)),
        ("num_range_frames", C.c_int32),
        ("max_iterations", C.c_int32),
        ("num_threads", C.c_int32),
        ("num_steps", C.c_int32),
        ("robustness", C.c_double),
        ("static_loss_type", C.c_int32),
        ("static_spatial_weight", C.c_double),
        ("static_depth_weight", C.c_double),
        ("smooth_loss_type", C.c_int32),
        ("smooth_static_weight", C.c_double),
        ("smooth_dynamic_weight", C.c_double),
        ("position_reg", C.c_double),
        ("scale_reg", C.c_double),
        ("scale_reg_grid_size", C.c_int32),
        ("depth_deform_reg_initial", C.c_double),
        ("depth_deform_reg_final", C.c_double),
        ("adaptive_deformation_cost", C.c_double),
        ("spatial_deform_reg", C.c_double),
        ("graduate_depth_deform_reg", C.c_int32),
        ("focal_reg", C.c_double),
        ("coarse_to_fine", C.c_int32),
        ("ctf_long", C.c_int32),
        ("ctf_short", C.c_int32),
        ("deferred_spatial_opt", C.c_int32),
        ("dso_long", C.c_int32),
        ("dso_short", C.c_int32),
        ("focal_long", C.c_double),
        ("intr_opt", C.c_int32),
        ("fix_poses", C.c_int32),
        ("fix_depth_xforms", C.c_int32),
        ("fix_spatial_xforms", C.c_int32),
        ("normalize_depth_from_first_frame", C.c_int32),
    ]

    @staticmethod
    def defaults():
        p = OptParams()
        p.frame_range = None
        p.num_range_frames = 0
        p.max_iterations = 1000
        p.num_threads = 12
        p.num_steps = 4
        p.robustness = 0.5
        p.static_loss_type = StaticLossType.ReproDisparity
        p.static_spatial_weight = 1.0
        p.static_depth_weight = 1.0
        p.smooth_loss_type = SmoothLossType.ReproDisparityLaplacian
        p.smooth_static_weight = 0.0
        p.smooth_dynamic_weight = 0.0
        p.position_reg = 0.0
        p.scale_reg = 1.0
        p.scale_reg_grid_size = 10
        p.depth_deform_reg_initial = 1.0
        p.depth_deform_reg_final = 0.1
        p.adaptive_deformation_cost = 0.0
        p.spatial_deform_reg = 1.0
        p.graduate_depth_deform_reg = 0
        p.focal_reg = 1.0
        p.coarse_to_fine = 1
        p.ctf_long = 17
        p.ctf_short = 10
        p.deferred_spatial_opt = 0
        p.dso_long = 4
        p.dso_short = 3
        p.focal_long = 0.3461538376301239
        p.intr_opt = IntrinsicsOptimization.PerFrame
        p.fix_poses = 0
        p.fix_depth_xforms = 0
        p.fix_spatial_xforms = 0
        p.normalize_depth_from_first_frame = 1
        return p

    def set_frame_range(self, frames):
        """Keeps the backing array alive on the struct instance."""
        if frames is None:
            self._range_keepalive = None
            self.frame_range = None
            self.num_range_frames = 0
        else:
            arr = (C.c_int32 * len(frames))(*[int(f) for f in frames])
            self._range_keepalive = arr
            self.frame_range = C.cast(arr, C.POINTER(C.c_int32))
            self.num_range_frames = len(frames)


class FramePose(C.Structure):
    _fields_ = [
        ("position", C.c_float * 3),
        ("orientation", C.c_float * 4),
        ("vfov", C.c_float),
        ("hfov", C.c_float),
    ]


class IterationRecord(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32),
        ("step_is_successful", C.c_int32),
        ("linear_iterations", C.c_int32),
        ("reserved", C.c_int32),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]


class SolveSummary(C.Structure):
    _fields_ = [
        ("num_iterations", C.c_int32),
        ("num_successful_steps", C.c_int32),
        ("termination", C.c_int32),
        ("num_residual_blocks", C.c_int32),
        ("num_parameters", C.c_int32),
        ("total_linear_iterations", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("total_seconds", C.c_double),
        ("evaluate_seconds", C.c_double),
        ("linear_solve_seconds", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}
