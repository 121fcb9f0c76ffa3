/*
 * cvd_types.h -- plain-C value types shared by the C-ABI of the MI355X geometric-consistency
 * optimizer (include/cvd_hip.h) and by the CPU oracle (oracle/cvd_oracle.cpp).
 *
 * Every enum / struct mirrors a type of the reference (facebookresearch/robust_cvd, paths relative
 * to the reference root):
 *   cvd_value_xform_type   <- lib/ValueTransform.h:16-20      (ValueXformType)
 *   cvd_xform_type         <- lib/DepthMapTransform.h:25-28   (XformType)
 *   cvd_depth_xform_type   <- lib/DepthMapTransform.h:31-36   (DepthXformType)
 *   cvd_spatial_xform_type <- lib/DepthMapTransform.h:39-46   (SpatialXformType)
 *   cvd_static_loss_type   <- lib/PoseOptimizer.h:22-27       (StaticLossType)
 *   cvd_smooth_loss_type   <- lib/PoseOptimizer.h:37-42       (SmoothLossType)
 *   cvd_intrinsics_opt     <- lib/PoseOptimizer.h:46-50       (IntrinsicsOptimization)
 *   cvd_xform_desc         <- lib/DepthMapTransform.h:50-84   (XformDescriptor)
 *   cvd_opt_params         <- lib/PoseOptimizer.h:54-108      (DepthVideoPoseOptimizer::Params)
 *   cvd_frame_pose         <- lib/DepthPhoto.h Extrinsics{position,orientation}+Intrinsics{vFov,hFov}
 *
 * Only plain pointers, sizes and PODs: no C++ / torch / Eigen types cross this boundary.
 */
#ifndef CVD_TYPES_H_
#define CVD_TYPES_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cvd_value_xform_type {
  CVD_VALUE_NONE = 0,
  CVD_VALUE_SCALE = 1,
  CVD_VALUE_SCALE_SHIFT = 2
} cvd_value_xform_type;

typedef enum cvd_xform_type { CVD_XFORM_DEPTH = 0, CVD_XFORM_SPATIAL = 1 } cvd_xform_type;

typedef enum cvd_depth_xform_type {
  CVD_DEPTH_NONE = 0,
  CVD_DEPTH_IDENTITY = 1,
  CVD_DEPTH_GLOBAL = 2,
  CVD_DEPTH_GRID = 3
} cvd_depth_xform_type;

typedef enum cvd_spatial_xform_type {
  CVD_SPATIAL_NONE = 0,
  CVD_SPATIAL_IDENTITY = 1,
  CVD_SPATIAL_VERTICAL_LINEAR = 2,
  CVD_SPATIAL_CORNERS_BILINEAR = 3,
  CVD_SPATIAL_BILINEAR_GRID = 4,
  CVD_SPATIAL_BICUBIC_GRID = 5
} cvd_spatial_xform_type;

typedef enum cvd_static_loss_type {
  CVD_STATIC_EUCLIDEAN = 0,
  CVD_STATIC_REPRO_DISPARITY = 1,
  CVD_STATIC_REPRO_DEPTH_RATIO = 2,
  CVD_STATIC_REPRO_LOG_DEPTH = 3
} cvd_static_loss_type;

typedef enum cvd_smooth_loss_type {
  CVD_SMOOTH_EUCLIDEAN_LAPLACIAN = 0,
  CVD_SMOOTH_REPRO_DISPARITY_LAPLACIAN = 1,
  CVD_SMOOTH_REPRO_DEPTH_RATIO_CONSISTENCY = 2,
  CVD_SMOOTH_REPRO_LOG_DEPTH_CONSISTENCY = 3
} cvd_smooth_loss_type;

typedef enum cvd_intrinsics_opt {
  CVD_INTR_FIXED = 0,
  CVD_INTR_SHARED = 1,
  CVD_INTR_PER_FRAME = 2
} cvd_intrinsics_opt;

/* XformDescriptor (lib/DepthMapTransform.h:50-84). grid_size = (cols, rows, depth-wise). */
typedef struct cvd_xform_desc {
  int32_t type;                /* cvd_xform_type */
  int32_t depth_type;          /* cvd_depth_xform_type */
  int32_t spatial_type;        /* cvd_spatial_xform_type */
  int32_t value_xform;         /* cvd_value_xform_type */
  int32_t cubic_interpolation; /* bool */
  int32_t grid_size[3];
  double depth_min_max[2];
} cvd_xform_desc;

/* DepthVideoPoseOptimizer::Params (lib/PoseOptimizer.h:54-108); defaults = cvd_opt_params_default(). */
typedef struct cvd_opt_params {
  const int32_t* frame_range; /* sorted frame ids; NULL => every frame (FrameRange) */
  int32_t num_range_frames;
  int32_t max_iterations;
  int32_t num_threads;
  int32_t num_steps;
  double robustness;
  int32_t static_loss_type;
  double static_spatial_weight;
  double static_depth_weight;
  int32_t smooth_loss_type;
  double smooth_static_weight;
  double smooth_dynamic_weight;
  double position_reg;
  double scale_reg;
  int32_t scale_reg_grid_size;
  double depth_deform_reg_initial;
  double depth_deform_reg_final;
  double adaptive_deformation_cost;
  double spatial_deform_reg;
  int32_t graduate_depth_deform_reg;
  double focal_reg;
  int32_t coarse_to_fine;
  int32_t ctf_long;
  int32_t ctf_short;
  int32_t deferred_spatial_opt;
  int32_t dso_long;
  int32_t dso_short;
  double focal_long;
  int32_t intr_opt;
  int32_t fix_poses;
  int32_t fix_depth_xforms;
  int32_t fix_spatial_xforms;
  int32_t normalize_depth_from_first_frame;
} cvd_opt_params;

/* DepthFrame pose as the reference stores it: float Extrinsics + float FOV (lib/DepthStream.h). */
typedef struct cvd_frame_pose {
  float position[3];
  float orientation[4]; /* quaternion coefficients x, y, z, w (Eigen storage order) */
  float vfov;
  float hfov;
} cvd_frame_pose;

/* One record per trust-region iteration (what Ceres prints with minimizer_progress_to_stdout). */
typedef struct cvd_iteration_record {
  int32_t iteration;
  int32_t step_is_successful;
  int32_t linear_iterations; /* inner PCG iterations (0 for a direct solve) */
  int32_t reserved;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} cvd_iteration_record;

/* Summary of one solve (ceres::Solver::Summary subset + timing split of SURVEY.md 8d). */
typedef struct cvd_solve_summary {
  int32_t num_iterations; /* LM iterations, successful + unsuccessful, Ceres' counting */
  int32_t num_successful_steps;
  int32_t termination; /* 0 convergence, 1 no-convergence (max iter), 2 failure */
  int32_t num_residual_blocks;
  int32_t num_parameters;
  int32_t total_linear_iterations;
  double initial_cost;
  double final_cost;
  double total_seconds;
  double evaluate_seconds;
  double linear_solve_seconds;
} cvd_solve_summary;

#ifdef __cplusplus
}
#endif

#endif /* CVD_TYPES_H_ */
