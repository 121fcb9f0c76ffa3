// cvd_setup.hip -- problem -> device layout: poses / transforms on the host, the Layout of a solve, the compiled constraint table and
// work decomposition, the coarse level's symbolic plan, state transfers, buffers, frame medians.
#include "cvd_host.h"
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace cvd {

// ---- rotation conversions on the host (ceres/rotation.h + Eigen semantics) -------------
static void quatToMatrix(const double q[4] /*x,y,z,w*/, double R[3][3]) {
  // columns = q * e_x, q * e_y, q * e_z with Eigen's v + w*uv + qv x uv, uv = 2 qv x v
  for (int c = 0; c < 3; ++c) {
    double v[3] = {0, 0, 0};
    v[c] = 1.0;
    const double uv[3] = {2.0 * (q[1] * v[2] - q[2] * v[1]), 2.0 * (q[2] * v[0] - q[0] * v[2]),
                          2.0 * (q[0] * v[1] - q[1] * v[0])};
    R[0][c] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
    R[1][c] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
    R[2][c] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
  }
}
static void matrixToAngleAxis(const double R[3][3], double aa[3]) {
  double q0, q1, q2, q3;  // w, x, y, z
  const double tr = R[0][0] + R[1][1] + R[2][2];
  if (tr >= 0.0) {
    double t = std::sqrt(tr + 1.0);
    q0 = 0.5 * t;
    t = 0.5 / t;
    q1 = (R[2][1] - R[1][2]) * t;
    q2 = (R[0][2] - R[2][0]) * t;
    q3 = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    double qq[4];
    qq[i + 1] = 0.5 * t;
    t = 0.5 / t;
    qq[0] = (R[k][j] - R[j][k]) * t;
    qq[j + 1] = (R[j][i] + R[i][j]) * t;
    qq[k + 1] = (R[k][i] + R[i][k]) * t;
    q0 = qq[0]; q1 = qq[1]; q2 = qq[2]; q3 = qq[3];
  }
  const double s2 = q1 * q1 + q2 * q2 + q3 * q3;
  if (s2 > 0.0) {
    const double s = std::sqrt(s2);
    const double two = 2.0 * ((q0 < 0.0) ? std::atan2(-s, -q0) : std::atan2(s, q0));
    const double k = two / s;
    aa[0] = q1 * k; aa[1] = q2 * k; aa[2] = q3 * k;
  } else {
    aa[0] = q1 * 2.0; aa[1] = q2 * 2.0; aa[2] = q3 * 2.0;
  }
}
static void angleAxisToMatrix(const double aa[3], double R[3][3]) {
  const double th2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (th2 > std::numeric_limits<double>::epsilon()) {
    const double th = std::sqrt(th2);
    const double wx = aa[0] / th, wy = aa[1] / th, wz = aa[2] / th;
    const double c = std::cos(th), s = std::sin(th);
    R[0][0] = c + wx * wx * (1 - c);       R[1][0] = wz * s + wx * wy * (1 - c);  R[2][0] = -wy * s + wx * wz * (1 - c);
    R[0][1] = wx * wy * (1 - c) - wz * s;  R[1][1] = c + wy * wy * (1 - c);       R[2][1] = wx * s + wy * wz * (1 - c);
    R[0][2] = wy * s + wx * wz * (1 - c);  R[1][2] = -wx * s + wy * wz * (1 - c); R[2][2] = c + wz * wz * (1 - c);
  } else {
    R[0][0] = 1;      R[1][0] = aa[2];  R[2][0] = -aa[1];
    R[0][1] = -aa[2]; R[1][1] = 1;      R[2][1] = aa[0];
    R[0][2] = aa[1];  R[1][2] = -aa[0]; R[2][2] = 1;
  }
}
static void matrixToQuat(const double R[3][3], double q[4] /*x,y,z,w*/) {
  double t = R[0][0] + R[1][1] + R[2][2];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (R[2][1] - R[1][2]) * t;
    q[1] = (R[0][2] - R[2][0]) * t;
    q[2] = (R[1][0] - R[0][1]) * t;
  } else {
    int i = 0;
    if (R[1][1] > R[0][0]) i = 1;
    if (R[2][2] > R[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(R[i][i] - R[j][j] - R[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (R[k][j] - R[j][k]) * t;
    q[j] = (R[j][i] + R[i][j]) * t;
    q[k] = (R[k][i] + R[i][k]) * t;
  }
}
std::vector<int> rangeOf(const cvd_opt_params& p, int F) {
  std::vector<int> r;
  if (!p.frame_range || p.num_range_frames <= 0) {
    for (int i = 0; i < F; ++i) r.push_back(i);
  } else {
    r.assign(p.frame_range, p.frame_range + p.num_range_frames);
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    for (int f : r)
      if (f < 0 || f >= F) throw std::runtime_error("frame range out of bounds");
  }
  return r;
}

// DepthVideoPoseOptimizer ctor, reference lib/PoseOptimizer.cpp:753-782
void posesToParams(cvd_handle* h) {
  h->poseParams.resize(h->F);
  for (int f = 0; f < h->F; ++f) {
    const cvd_frame_pose& p = h->poses[f];
    auto& pose = h->poseParams[f];
    pose[0] = p.position[0];
    pose[1] = p.position[1];
    pose[2] = p.position[2];
    const double q[4] = {p.orientation[0], p.orientation[1], p.orientation[2], p.orientation[3]};
    double R[3][3];
    quatToMatrix(q, R);  // columns right, up, -front == q*ex, q*ey, q*ez
    matrixToAngleAxis(R, &pose[3]);
    pose[6] = std::tan(p.vfov / 2.0);
  }
  h->poseParamsValid = true;
}

// pose write-back, reference lib/PoseOptimizer.cpp:964-987
void paramsToPoses(cvd_handle* h, const cvd_opt_params& params) {
  for (int f : rangeOf(params, h->F)) {
    const auto& pose = h->poseParams[f];
    cvd_frame_pose& p = h->poses[f];
    p.position[0] = static_cast<float>(pose[0]);
    p.position[1] = static_cast<float>(pose[1]);
    p.position[2] = static_cast<float>(pose[2]);
    double R[3][3], q[4];
    angleAxisToMatrix(&pose[3], R);
    matrixToQuat(R, q);
    for (int i = 0; i < 4; ++i) p.orientation[i] = static_cast<float>(q[i]);
    const double fsrc = (params.intr_opt == CVD_INTR_SHARED) ? h->poseParams[0][6] : pose[6];
    p.vfov = static_cast<float>(std::atan(fsrc) * 2.f);
    p.hfov = static_cast<float>(std::atan(fsrc * h->aspect) * 2.f);
  }
}

void resetXforms(cvd_handle* h, const cvd_xform_desc& d, bool spatial) {
  const int nb = xformNumBlocks(d), bs = xformBlockSize(d);
  if (!spatial) {
    if (d.type != CVD_XFORM_DEPTH) throw std::runtime_error("Transform has the wrong type.");
    if (d.depth_type == CVD_DEPTH_GRID && d.grid_size[2] > 1) {
      // GridDepthXform ctor, reference lib/DepthMapTransform.cpp:709-717
      if (d.depth_min_max[0] <= 0.0 || d.depth_min_max[1] <= 0.0) throw std::runtime_error("Depth values must be positive.");
      if (d.depth_min_max[1] - d.depth_min_max[0] <= 0.0) throw std::runtime_error("Depth range must be positive.");
      if (d.cubic_interpolation)
        throw std::runtime_error("Cubic interpolation of depth-wise grids is not defined (reference lib/DepthMapTransform.cpp:944 "
                                 "never applies the depth-wise weights).");
    }
    if (d.depth_type == CVD_DEPTH_GRID && !d.cubic_interpolation && bs != 1 && (d.grid_size[0] > 1 || d.grid_size[2] > 1))
      throw std::runtime_error(
          "Linear grid gather is only defined for 1-parameter value transforms (reference "
          "lib/DepthMapTransform.cpp:829 aliases the blocks otherwise).");
    h->ddesc = d;
    h->dparams.assign(static_cast<size_t>(h->F) * nb * bs, 1.0);
  } else {
    if (d.type != CVD_XFORM_SPATIAL) throw std::runtime_error("Transform has the wrong type.");
    h->sdesc = d;
    h->sparams.assign(static_cast<size_t>(h->F) * nb * bs, 0.0);
  }
}

// DepthVideoProcessor::gridXformSplit, reference lib/Processor.cpp:888-985
void gridXformSplit(cvd_handle* h, const cvd_xform_desc& nd) {
  if (nd.depth_type != CVD_DEPTH_GRID) throw std::runtime_error("Transform type must be a grid type.");
  const cvd_xform_desc prev = h->ddesc;
  if (prev.depth_type != CVD_DEPTH_GLOBAL && prev.depth_type != CVD_DEPTH_GRID)
    throw std::runtime_error("Can only split global or grid type transforms.");
  if (nd.value_xform != prev.value_xform)
    throw std::runtime_error("Old and new transforms must use same value transform.");
  if (prev.depth_type != CVD_DEPTH_GLOBAL &&
      (prev.grid_size[0] > nd.grid_size[0] || prev.grid_size[1] > nd.grid_size[1]))
    throw std::runtime_error(
        "New transform must have at least the same number of rows and columns as the old transform.");
  const std::vector<double> old = h->dparams;
  const int oldN = h->nD();
  resetXforms(h, nd, false);
  const int N = xformBlockSize(nd);
  const int newCols = nd.grid_size[0], newRows = nd.grid_size[1];
  const int newN = h->nD();
  for (int f = 0; f < h->F; ++f) {
    const double* po = &old[static_cast<size_t>(f) * oldN];
    double* pn = &h->dparams[static_cast<size_t>(f) * newN];
    for (int row = 0; row < newRows; ++row) {
      for (int col = 0; col < newCols; ++col) {
        double* dst = pn + static_cast<size_t>(col + row * newCols) * N;
        if (prev.depth_type == CVD_DEPTH_GLOBAL) {
          for (int i = 0; i < N; ++i) dst[i] = po[i];
        } else {
          const int prevRows = prev.grid_size[1], prevCols = prev.grid_size[0];
          const double maxx = std::nextafter(static_cast<double>(prevCols - 1), 0.0);
          const double maxy = std::nextafter(static_cast<double>(prevRows - 1), 0.0);
          const double sx = std::min(col / double(newCols - 1) * (prevCols - 1), maxx);
          const double sy = std::min(row / double(newRows - 1) * (prevRows - 1), maxy);
          const int ix = static_cast<int>(sx), iy = static_cast<int>(sy);
          const double rx = sx - ix, ry = sy - iy;
          const double* b0 = po + static_cast<size_t>(ix + iy * prevCols) * N;
          const double* b1 = po + static_cast<size_t>((ix + 1) + iy * prevCols) * N;
          const double* b2 = po + static_cast<size_t>(ix + (iy + 1) * prevCols) * N;
          const double* b3 = po + static_cast<size_t>((ix + 1) + (iy + 1) * prevCols) * N;
          const double w0 = (1.f - rx) * (1.f - ry), w1 = rx * (1.f - ry), w2 = (1.f - rx) * ry, w3 = rx * ry;
          for (int i = 0; i < N; ++i) dst[i] = b0[i] * w0 + b1[i] * w1 + b2[i] * w2 + b3[i] * w3;
        }
      }
    }
  }
}
// ---- problem -> device layout -------------------------------------------------------------------------
Layout makeLayout(cvd_handle* h, const cvd_opt_params& p, double depthDeformReg, ProblemKind kind) {
  if ((p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0) && kind == PK_POSE_STEP) {
    if (p.smooth_loss_type < CVD_SMOOTH_EUCLIDEAN_LAPLACIAN || p.smooth_loss_type > CVD_SMOOTH_REPRO_LOG_DEPTH_CONSISTENCY)
      throw std::runtime_error("Invalid loss type.");
    if (!h->haveTriplets) throw std::runtime_error("Missing triplet constraints.");
  }
  Layout L{};
  L.F = h->F;
  L.B = h->Bsz();
  L.depthType = h->ddesc.depth_type;
  L.N = xformBlockSize(h->ddesc);
  L.cubic = h->ddesc.cubic_interpolation ? 1 : 0;
  L.gx = h->ddesc.depth_type == CVD_DEPTH_GRID ? h->ddesc.grid_size[0] : 1;
  L.gy = h->ddesc.depth_type == CVD_DEPTH_GRID ? h->ddesc.grid_size[1] : 1;
  L.maxcx = std::nextafter(static_cast<double>(L.gx - 1), 0.0);
  L.maxcy = std::nextafter(static_cast<double>(L.gy - 1), 0.0);
  L.gz = h->ddesc.depth_type == CVD_DEPTH_GRID ? std::max(1, h->ddesc.grid_size[2]) : 1;
  L.maxcz = std::nextafter(static_cast<double>(L.gz - 1), 0.0);
  L.dispMin = 0.0;
  L.dispInterval = 1.0;
  if (L.gz > 1) {  // reference lib/DepthMapTransform.cpp:719-729
    const double dmin = 1.0 / h->ddesc.depth_min_max[1], dmax = 1.0 / h->ddesc.depth_min_max[0];
    L.dispMin = dmin;
    L.dispInterval = (dmax - dmin) / (L.gz - 1);
  }
  L.nD = h->nD();
  L.spatialType = h->sdesc.spatial_type;
  L.sgx = h->sdesc.grid_size[0];
  L.sgy = h->sdesc.grid_size[1];
  L.smaxcx = std::nextafter(static_cast<double>(L.sgx - 1), 0.0);
  L.smaxcy = std::nextafter(static_cast<double>(L.sgy - 1), 0.0);
  L.nS = h->nS();
  L.aspect = h->aspect;
  L.vFocal = (h->aspect >= 1.f ? p.focal_long / static_cast<double>(h->aspect) : p.focal_long);
  L.intrOpt = p.intr_opt;
  L.lossType = p.static_loss_type;
  L.ws = p.static_spatial_weight;
  L.wd = p.static_depth_weight;
  L.cauchyB = p.robustness * p.robustness;
  L.cauchyC = 1.0 / L.cauchyB;
  L.robustA = p.robustness;
  if (h->opt.robust_loss != 0 && h->opt.robust_loss != 1) throw std::runtime_error("robust_loss must be 0 (Cauchy) or 1 (Huber)");
  L.robustKind = h->opt.robust_loss;
  // scale regulariser sample grid, reference lib/PoseOptimizer.cpp:1347-1351
  int gX = p.scale_reg_grid_size;
  int gY = static_cast<int>(std::round(static_cast<float>(gX) * h->invAspect));
  if (h->aspect <= 1.f) std::swap(gX, gY);
  L.sregX = gX;
  L.sregY = gY;
  if (kind == PK_POSE_STEP) {
    L.includeStatic = 1;
    L.scaleRegSqrt = (!p.fix_depth_xforms && p.scale_reg > 0.0) ? std::sqrt(p.scale_reg) : 0.0;
    L.focalRegSqrt = (p.focal_reg > 0.0 && p.intr_opt != CVD_INTR_FIXED) ? std::sqrt(p.focal_reg) : 0.0;
    L.depthDeformW = depthDeformReg > 0.0 ? depthDeformReg : 0.0;
    L.spatialDeformW = p.spatial_deform_reg > 0.0 ? p.spatial_deform_reg : 0.0;
  } else {
    // normalizeDepthFromFirstFrame (default): no pairs at all (reference lib/PoseOptimizer.cpp:1014-1018); otherwise the
    // pair loop of :1020-1105: one DisparityDissimilarityCost with CauchyLoss(robustness) per constraint
    L.includeStatic = p.normalize_depth_from_first_frame ? 0 : 1;
    if (L.includeStatic) {
      L.lossType = kLossNormalizeDisparity;
      L.robustKind = kRobustCauchy;  // (the reference hard-wires CauchyLoss here, :1080)
    }
    L.scaleRegSqrt = p.scale_reg > 0.0 ? std::sqrt(p.scale_reg) : 0.0;
    L.focalRegSqrt = 0.0;
    L.depthDeformW = p.depth_deform_reg_initial > 0.0 ? p.depth_deform_reg_initial : 0.0;
    L.spatialDeformW = 0.0;
  }
  {
    const std::vector<int> rg = rangeOf(p, h->F);
    L.firstFrame = rg.empty() ? 0 : rg.front();
    L.lastFrame = rg.empty() ? 0 : rg.back();
    L.positionRegSqrt = (kind == PK_POSE_STEP && p.position_reg > 0.0) ? std::sqrt(p.position_reg) : 0.0;
    L.rank = h->rank;
    L.world = h->world;
  }
  if (L.scaleRegSqrt > 0.0 && (L.sregX < 2 || L.sregY < 2))
    throw std::runtime_error("scaleRegGridSize too small for this aspect ratio.");
  // AdaptiveDeformationCost replaces DeformationCost for the depth transforms that have deformation residuals, i.e.
  // grids (reference lib/PoseOptimizer.cpp:1465-1484: the others are skipped before the cost is constructed)
  L.adaptW = nullptr;
  L.adaptive = 0.0;
  if (p.adaptive_deformation_cost > 0.0 && L.depthDeformW > 0.0 && L.depthType == CVD_DEPTH_GRID) {
    if (!h->haveDynMasks) throw std::runtime_error("Adaptive smoothness requires a dynamic mask stream.");
    if (L.gz > 1) throw std::runtime_error("AdaptiveDeformationCost with a depth-wise grid is not implemented on the device path.");
    if (L.gx < 2 || L.gy < 2) throw std::runtime_error("Adaptive deformation cost needs a grid of at least 2 x 2 vertices.");
    if (h->adaptGx != L.gx || h->adaptGy != L.gy) {
      const size_t G = static_cast<size_t>(L.gx) * L.gy;
      h->dAdaptW.ensure(G * h->F);
      hipLaunchKernelGGL(k_adaptive_weights, dim3(h->F), dim3(256), 2 * G * sizeof(double), h->stream, h->dDynMask.p, h->dynW,
                         h->dynH, L.gx, L.gy, h->dAdaptW.p);
      HIP_CHECK(hipGetLastError());
      h->adaptGx = L.gx;
      h->adaptGy = L.gy;
    }
    L.adaptW = h->dAdaptW.p;
    L.adaptive = p.adaptive_deformation_cost;
  }
  return L;
}
void checkFrameBlock(size_t B, const char* what) {
  if (B > static_cast<size_t>(kMaxFrameBlock))
    throw std::runtime_error(fmt("%s: %zu unknowns per frame (7 + depth-transform + spatial-transform parameters) exceed the "
                                 "%d this build supports", what, B, kMaxFrameBlock));
}

void tapCounts(const Layout& L, int& KD, int& KS) {
  KD = (L.depthType == CVD_DEPTH_GRID) ? (L.cubic ? 16 : 4) : 1;
  // depth-wise grids: 8 taps (spatial x depth-wise) run in the 16-slot instantiation, 2 taps (depth-wise only) in the 4-slot
  if (L.depthType == CVD_DEPTH_GRID && L.gz > 1) KD = L.gx > 1 ? 16 : 4;
  switch (L.spatialType) {
    case CVD_SPATIAL_IDENTITY: KS = 0; break;
    case CVD_SPATIAL_BICUBIC_GRID: KS = 16; break;
    default: KS = 4;
  }
}

// Scope of the specialised fast kernels: identity spatial transform and the three reprojection losses.
bool fastLoss(const Layout& L) {
  if (L.gz > 1) return false;  // (the fast kernels gather 2-D grids only)
  if (L.B > 256) return false;  // (and hold one element of a frame block per thread)
  return L.lossType == CVD_STATIC_REPRO_DISPARITY || L.lossType == CVD_STATIC_REPRO_DEPTH_RATIO || L.lossType == CVD_STATIC_REPRO_LOG_DEPTH;
}
Table makeTable(cvd_handle* h) {
  Table T{};
  T.ndc = h->dense ? nullptr : (h->tableOrdered ? h->dNdcOrd.p : h->dNdc.p);
  T.dsrc = h->dense ? nullptr : (h->tableOrdered ? h->dDsrcOrd.p : h->dDsrc.p);
  T.pairA = h->dPairA.p;
  T.pairB = h->dPairB.p;
  T.pairOff = h->dPairOff.p;
  T.flow = h->dense ? h->dFlow.p : nullptr;
  T.fmask = h->dense ? h->dFMask.p : nullptr;
  T.depth = h->dDepth.p;
  T.W = h->W;
  T.H = h->H;
  T.sx = 1.f / static_cast<float>(h->W);                 // reference lib/FlowConstraints.cpp:371: Vector2f scale(1.f / w, invAspect / h)
  T.sy = h->invAspect / static_cast<float>(h->H);
  T.invAspect = h->invAspect;
  return T;
}
// Scope of the dense mode as a function of the CALLER's parameters (cvd_dense_mode_supported: lib_python asks before it hands
// a matchSeparation = 0 collection over as images instead of a list) -- the same conditions checkDenseScope
// enforces at solve time, for every step of the schedule the parameters describe.
bool denseFastBlockFits(long long B);
bool denseModeSupported(const cvd_opt_params& p, const cvd_xform_desc& dd, const cvd_xform_desc& sd, bool haveTriplets, int world,
                        bool normalize) {
  (void)world;  // (pair-sharded runs hand every rank the flow images of ITS pairs: same scope)
  if (normalize) return p.normalize_depth_from_first_frame != 0;  // (the pair loop's DisparityDissimilarityCost is a generic-kernel loss)
  if (sd.spatial_type != CVD_SPATIAL_IDENTITY || p.deferred_spatial_opt) return false;
  if (p.static_loss_type != CVD_STATIC_REPRO_DISPARITY && p.static_loss_type != CVD_STATIC_REPRO_DEPTH_RATIO &&
      p.static_loss_type != CVD_STATIC_REPRO_LOG_DEPTH)
    return false;
  (void)haveTriplets;
  if (p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0) return false;  // (scene-flow triplets: generic kernels)
  if (p.intr_opt == CVD_INTR_SHARED) return false;
  // an Identity depth transform has no value parameter (N = 0): the fast kernels and checkDenseScope need N == 1
  if (dd.depth_type == CVD_DEPTH_IDENTITY) return false;
  if (dd.value_xform != CVD_VALUE_SCALE) return false;
  if (dd.depth_type == CVD_DEPTH_GRID && (dd.cubic_interpolation || dd.grid_size[2] > 1)) return false;
  // the largest frame block of the schedule must stay within what the image-reading kernels hold in LDS (199 unknowns)
  long long g = dd.depth_type == CVD_DEPTH_GRID ? static_cast<long long>(dd.grid_size[0]) * dd.grid_size[1] : 1;
  if (p.coarse_to_fine && p.num_steps > 1) g = std::max(g, static_cast<long long>(p.ctf_long) * p.ctf_short);
  return denseFastBlockFits(7 + g);
}
// The image-reading kernels keep a frame's packed lower triangle in LDS (k_assemble_fast; the fold of the walk's records): B <= 199 at
// 160 KB.  Beyond that a dense-mode solve takes the list route like every other configuration outside their scope -- the generic
// list kernels walk the triangle in panels.  (Until round 6 such a solve -- e.g. a 19 x 13 grid, B = 254 -- reached the generic
// kernels WITHOUT a list: a memory fault, found by tests/test_gpu_dense_mode.py's grid19x13 case.)
bool denseFastBlockFits(long long B) {
  return B >= 1 && B <= 256 && (static_cast<size_t>(B) * (B + 1) / 2 + 2 * B + 4 * 36) * 8 <= kMaxLds;
}

// Dense mode: the specialised kernels (pixel walk, image-reading product / cost) cover the default residual configuration of the
// reference pipeline; every other configuration runs on the list the images stand for, materialised on the device (DenseListScope).
bool denseFastScope(const cvd_handle* h, const Layout& L, int KS, bool trip) {
  return !(KS != 0 || !fastLoss(L) || L.N != 1 || trip || L.intrOpt == CVD_INTR_SHARED || h->forceGeneric || L.cubic ||
           !denseFastBlockFits(L.B));
}
void checkDenseScope(cvd_handle* h, const Layout& L, int KS, bool trip) {
  if (!h->dense) return;
  if (!denseFastScope(h, L, KS, trip))
    throw std::logic_error("dense mode: a configuration outside the fast kernels' scope reached them (DenseListScope should have taken it)");
}
DenseListScope::DenseListScope(cvd_handle* hh, bool needList) : h(hh) {
  if (!h->dense || !needList) return;
  hipStream_t s = h->stream;
  const int npx = h->W * h->H;
  const int chunks = (npx + kDlChunk - 1) / kDlChunk;
  const size_t nwg = static_cast<size_t>(h->P) * chunks;
  if (!h->denseListValid) {
    const Table T = makeTable(h);
    h->dDlCounts.ensure(std::max<size_t>(1, nwg));
    h->dDlOffsets.ensure(std::max<size_t>(1, nwg));
    std::vector<int> counts(nwg, 0);
    if (nwg > 0) {
      hipLaunchKernelGGL(k_dense_list, dim3(static_cast<unsigned>(nwg)), dim3(256), 0, s, T, chunks, h->dDlCounts.p, nullptr, nullptr, nullptr, nullptr);
      HIP_CHECK(hipGetLastError());
      h->dDlCounts.download(counts.data(), nwg, s);
      HIP_CHECK(hipStreamSynchronize(s));
    }
    std::vector<long long> offs(std::max<size_t>(1, nwg), 0);
    h->denseListOff.assign(h->P + 1, 0);
    long long o = 0;
    for (int p = 0; p < h->P; ++p) {
      h->denseListOff[p] = o;
      for (int c = 0; c < chunks; ++c) { offs[static_cast<size_t>(p) * chunks + c] = o; o += counts[static_cast<size_t>(p) * chunks + c]; }
    }
    h->denseListOff[h->P] = o;
    h->dLoc.ensure(std::max<long long>(o, 1));
    h->dCPair.ensure(std::max<long long>(o, 1));
    h->dStatic.ensure(std::max<long long>(o, 1));
    if (nwg > 0) {
      h->dDlOffsets.upload(offs.data(), offs.size(), s);
      hipLaunchKernelGGL(k_dense_list, dim3(static_cast<unsigned>(nwg)), dim3(256), 0, s, T, chunks, h->dDlCounts.p, h->dDlOffsets.p, h->dLoc.p,
                         h->dCPair.p, h->dStatic.p);
      HIP_CHECK(hipGetLastError());
    }
    h->denseListValid = true;
  }
  // the handle runs this solve as a list-mode handle
  h->dense = false;
  h->denseAsList = true;
  h->pairOff = h->denseListOff;
  h->C = h->denseListOff[h->P];
  h->dPairOff.upload(h->pairOff.data(), h->pairOff.size(), s);
  h->tableValid = false;
  entered = true;
}
DenseListScope::~DenseListScope() {
  if (!entered) return;
  const long long npx = static_cast<long long>(h->W) * h->H;
  h->dense = true;
  h->denseAsList = false;
  for (int p = 0; p <= h->P; ++p) h->pairOff[p] = static_cast<long long>(p) * npx;
  h->C = static_cast<long long>(h->P) * npx;
  try { h->dPairOff.upload(h->pairOff.data(), h->pairOff.size(), h->stream); } catch (...) {}
  h->tableValid = false;
}
// Row panels of the packed lower triangle of a B x B frame block that fit `capDoubles` of LDS each (AsmPanels,
// cvd_kernels.h): one panel up to B = 199, two at the reference's default deferred-spatial block B = 201.
AsmPanels makePanels(int B, size_t capDoubles, int& panelCap) {
  AsmPanels P{};
  P.n = 0;
  P.row[0] = 0;
  size_t biggest = 0;
  int r0 = 0;
  while (r0 < B) {
    if (P.n >= 8) throw std::runtime_error(fmt("frame block of %d unknowns is too large for the assembly kernels", B));
    const size_t base = static_cast<size_t>(r0) * (r0 + 1) / 2;
    int r1 = r0;
    while (r1 < B && static_cast<size_t>(r1 + 1) * (r1 + 2) / 2 - base <= capDoubles) ++r1;
    if (r1 < std::max(r0 + 1, 7) && r1 < B) throw std::runtime_error("LDS panel too small for the assembly kernels");
    biggest = std::max(biggest, static_cast<size_t>(r1) * (r1 + 1) / 2 - base);
    P.row[++P.n] = r1;
    r0 = r1;
  }
  panelCap = static_cast<int>(biggest);
  return P;
}
// ---- coarse level: symbolic block-sparse Cholesky of the frame graph (cvd_coarse.h) --------------------------
// Greedy minimum-degree ordering on the frame graph, column structures with fill, left-looking update lists and
// the level schedule (level of a column = 1 + the highest level among the columns that update it).
static void buildCoarsePlan(cvd_handle* h, const std::vector<std::pair<int, int>>& edgeList,
                            const std::vector<int>& itemEdge) {
  auto& C = h->coarse;
  const int F = h->F;
  hipStream_t s = h->stream;
  std::vector<std::set<int>> adj(F);
  for (const auto& e : edgeList) {
    adj[e.first].insert(e.second);
    adj[e.second].insert(e.first);
  }
  // Multilevel independent-set ordering: every round eliminates a maximal independent set of low-degree frames
  // (degree <= 2 * current minimum + 2), which become one level of the factorisation; the elimination graph
  // receives the fill.  Far fewer levels than plain minimum degree on these near-chain graphs (34 vs 73 for the
  // 300-frame hierarchical pair set) at ~15% more fill.
  std::vector<int> order, pos(F, -1);
  std::vector<std::vector<int>> structFrames(F);  // by position: neighbours (frames) alive at elimination
  {
    std::vector<std::set<int>> g = adj;
    std::vector<char> alive(F, 1);
    int remaining = F;
    while (remaining > 0) {
      size_t minDeg = std::numeric_limits<size_t>::max();
      for (int v = 0; v < F; ++v)
        if (alive[v]) minDeg = std::min(minDeg, g[v].size());
      const size_t cap = 2 * minDeg + 2;
      std::vector<int> cand;
      for (int v = 0; v < F; ++v)
        if (alive[v] && g[v].size() <= cap) cand.push_back(v);
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return g[a].size() < g[b].size(); });
      std::vector<char> blocked(F, 0);
      std::vector<int> chosen;
      for (int v : cand) {
        if (blocked[v]) continue;
        chosen.push_back(v);
        blocked[v] = 1;
        for (int u : g[v]) blocked[u] = 1;
      }
      for (int v : chosen) {
        pos[v] = static_cast<int>(order.size());
        order.push_back(v);
        std::vector<int> nb(g[v].begin(), g[v].end());
        structFrames[pos[v]] = nb;
        for (int a2 : nb) {
          g[a2].erase(v);
          for (int b2 : nb)
            if (a2 != b2) g[a2].insert(b2);
        }
        alive[v] = 0;
        g[v].clear();
        --remaining;
      }
    }
  }
  // column structures by position (sorted), block ids
  std::vector<int> colPtr(F + 1, 0), blkCol, blkRow;
  std::vector<std::vector<int>> colRows(F);
  for (int j = 0; j < F; ++j) {
    for (int fr : structFrames[j]) colRows[j].push_back(pos[fr]);
    std::sort(colRows[j].begin(), colRows[j].end());
    colPtr[j + 1] = colPtr[j] + static_cast<int>(colRows[j].size());
  }
  const int nnz = colPtr[F];
  const int nBlocks = F + nnz;
  blkCol.assign(nBlocks, 0);
  blkRow.assign(nBlocks, 0);
  std::map<std::pair<int, int>, int> blockOf;  // (row position, column position) -> block id
  for (int j = 0; j < F; ++j) {
    blkCol[j] = j;
    blkRow[j] = j;
    blockOf[{j, j}] = j;
    for (size_t e = 0; e < colRows[j].size(); ++e) {
      const int b = F + colPtr[j] + static_cast<int>(e);
      blkCol[b] = j;
      blkRow[b] = colRows[j][e];
      blockOf[{colRows[j][e], j}] = b;
    }
  }
  // row structures and levels
  std::vector<std::vector<int>> rowBlks(F);
  for (int b = F; b < nBlocks; ++b) rowBlks[blkRow[b]].push_back(b);
  std::vector<int> level(F, 0);
  for (int j = 0; j < F; ++j)
    for (int b : rowBlks[j]) level[j] = std::max(level[j], level[blkCol[b]] + 1);
  const int nLevels = F ? *std::max_element(level.begin(), level.end()) + 1 : 0;
  std::vector<int> rowPtr(F + 1, 0), rowBlk;
  for (int j = 0; j < F; ++j) {
    std::sort(rowBlks[j].begin(), rowBlks[j].end(), [&](int a, int b) { return blkCol[a] < blkCol[b]; });
    rowPtr[j + 1] = rowPtr[j] + static_cast<int>(rowBlks[j].size());
    rowBlk.insert(rowBlk.end(), rowBlks[j].begin(), rowBlks[j].end());
  }
  // update lists: column k contributes L(i,k) L(j,k)^T to block (i, j) for every i >= j in struct(k)
  std::vector<std::vector<std::pair<int, int>>> upd(nBlocks);
  for (int k = 0; k < F; ++k) {
    const auto& rows = colRows[k];
    for (size_t a = 0; a < rows.size(); ++a)
      for (size_t b = a; b < rows.size(); ++b) {
        const int j = rows[a], i = rows[b];  // i >= j
        const int target = blockOf.at({i, j});
        upd[target].push_back({F + colPtr[k] + static_cast<int>(b), F + colPtr[k] + static_cast<int>(a)});
      }
  }
  std::vector<int> updPtr(nBlocks + 1, 0), updA, updB, updBlk;
  for (int b = 0; b < nBlocks; ++b) {
    updPtr[b + 1] = updPtr[b] + static_cast<int>(upd[b].size());
    for (const auto& u : upd[b]) { updA.push_back(u.first); updB.push_back(u.second); updBlk.push_back(b); }
  }
  std::vector<int> levelPtr(nLevels + 1, 0), levelCols, lvlBlkPtr(nLevels + 1, 0), lvlBlks;
  for (int lv = 0; lv < nLevels; ++lv) {
    for (int j = 0; j < F; ++j)
      if (level[j] == lv) {
        levelCols.push_back(j);
        lvlBlks.push_back(j);
        for (int e = colPtr[j]; e < colPtr[j + 1]; ++e) lvlBlks.push_back(F + e);
      }
    levelPtr[lv + 1] = static_cast<int>(levelCols.size());
    lvlBlkPtr[lv + 1] = static_cast<int>(lvlBlks.size());
  }
  // W = L^-1: column j is non-zero on the elimination-tree path j -> root (parent = first row below the diagonal)
  std::vector<int> wPtr(F + 1, 0), wRow;
  for (int j = 0; j < F; ++j) {
    for (int i = j; i >= 0; i = colRows[i].empty() ? -1 : colRows[i][0]) wRow.push_back(i);
    wPtr[j + 1] = static_cast<int>(wRow.size());
  }
  const int nW = static_cast<int>(wRow.size());
  std::vector<std::vector<std::pair<int, int>>> wt(F);  // row -> (column, W block id), columns ascending
  for (int j = 0; j < F; ++j)
    for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) wt[wRow[t]].push_back({j, t});
  std::vector<int> wtPtr(F + 1, 0), wtBlk, wtCol, wtFrame;
  std::vector<int> wSlot(nW, 0);  // W block -> its slot in the row lists (coarseColumnProducts writes there)
  for (int i = 0; i < F; ++i) {
    wtPtr[i + 1] = wtPtr[i] + static_cast<int>(wt[i].size());
    for (const auto& e : wt[i]) {
      wSlot[e.second] = static_cast<int>(wtBlk.size());
      wtCol.push_back(e.first); wtBlk.push_back(e.second); wtFrame.push_back(order[e.first]);
    }
  }
  std::vector<int> wuPtr(nW + 1, 0), wuL, wuW;
  {
    std::vector<int> mark(F, -1);
    for (int j = 0; j < F; ++j) {
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) mark[wRow[t]] = t;
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) {
        const int i = wRow[t];
        if (t > wPtr[j])
          for (int b : rowBlks[i]) {
            const int wk = mark[blkCol[b]];
            if (wk >= 0 && wk < t) { wuL.push_back(b); wuW.push_back(wk); }
          }
        wuPtr[t + 1] = static_cast<int>(wuL.size());
      }
      for (int t = wPtr[j]; t < wPtr[j + 1]; ++t) mark[wRow[t]] = -1;
    }
  }
  std::vector<int> edgeBlk, edgeFa, edgeFb;
  for (const auto& e : edgeList) {
    const int pa = pos[e.first], pb = pos[e.second];
    // stored rows = fa, columns = fb; block (i, j), i > j, has rows = frame of position i
    const int b = blockOf.at({std::max(pa, pb), std::min(pa, pb)});
    edgeBlk.push_back((b << 1) | (pa > pb ? 0 : 1));
    edgeFa.push_back(e.first);
    edgeFb.push_back(e.second);
  }
  C.nEdges = static_cast<int>(edgeList.size());
  C.edgeFaHost = edgeFa;
  C.edgeFbHost = edgeFb;
  C.nBlocks = nBlocks;
  C.nLevels = nLevels;
  C.itemEdge = itemEdge;
  auto up = [&](DevBuf<int>& d, const std::vector<int>& v) { d.upload(v.data(), v.size(), s); };
  if (h->opt.verbose >= 3) {  // development aid: shape of the elimination levels
    for (int lv = 0; lv < nLevels; ++lv) {
      long long nupd = 0, maxCol = 0, nblk = 0, maxChain = 0;
      for (int q = levelPtr[lv]; q < levelPtr[lv + 1]; ++q) {
        const int j = levelCols[q];
        long long colUpd = 0, chain = 0;
        const int nOff = colPtr[j + 1] - colPtr[j];
        for (int k = 0; k <= nOff; ++k) {
          const int b = (k == 0) ? j : F + colPtr[j] + k - 1;
          const long long u = updPtr[b + 1] - updPtr[b];
          colUpd += u;
          chain += u ? (u + 15) / 16 + 1 : 0;
        }
        nblk += nOff + 1;
        nupd += colUpd;
        maxCol = std::max(maxCol, colUpd);
        maxChain = std::max(maxChain, chain);
      }
      std::printf("level %2d: cols %3d blocks %5lld updates %6lld  max updates/col %5lld  max serial steps/col %4lld\n", lv,
                  levelPtr[lv + 1] - levelPtr[lv], nblk, nupd, maxCol, maxChain);
    }
  }
  up(C.order, order); up(C.pos, pos); up(C.levelPtr, levelPtr); up(C.levelCols, levelCols);
  up(C.lvlBlkPtr, lvlBlkPtr); up(C.lvlBlks, lvlBlks); up(C.blkCol, blkCol); up(C.blkRow, blkRow);
  up(C.colPtr, colPtr); up(C.rowPtr, rowPtr); up(C.rowBlk, rowBlk); up(C.updPtr, updPtr); up(C.updA, updA);
  up(C.updB, updB); up(C.updBlk, updBlk); up(C.edgeBlk, edgeBlk); up(C.edgeFa, edgeFa); up(C.edgeFb, edgeFb);
  up(C.wPtr, wPtr); up(C.wRow, wRow); up(C.wtPtr, wtPtr); up(C.wtBlk, wtBlk); up(C.wtCol, wtCol); up(C.wtFrame, wtFrame);
  up(C.wuPtr, wuPtr); up(C.wuL, wuL); up(C.wuW, wuW); up(C.wSlot, wSlot);
  C.nW = nW;
  up(C.itemEdgeDev, itemEdge);
  const size_t n = static_cast<size_t>(F) * kCB;
  C.edges.ensure(static_cast<size_t>(std::max(C.nEdges, 1)) * kCBB);
  C.diag.ensure(static_cast<size_t>(h->framesPadded()) * kCBB);
  C.dropDiag.ensure(static_cast<size_t>(F) * kCBB);
  C.Lb.ensure(static_cast<size_t>(nBlocks) * kCBB);
  C.Linv.ensure(static_cast<size_t>(F) * kCBB);
  C.Wb.ensure(static_cast<size_t>(nW) * kCBB);
  C.Wb2.ensure(static_cast<size_t>(nW) * kCBB);
  C.fail2.ensure(1);
  C.rc.ensure(n);
  C.qc.ensure(n);
  C.wq.ensure(static_cast<size_t>(nW) * kCB);
  C.fdotY.ensure(F);
  C.y.ensure(n);
  C.c.ensure(static_cast<size_t>(h->framesPadded()) * kCB);  // (all-gathered in place by the owner-sharded PCG iteration)
  C.dotPart.ensure(static_cast<size_t>(F) * 2);  // (dense level: the dense-level workgroups' rows' shares, then the frame workgroups')
  C.modeActive.ensure(n);
  C.fail.ensure(1);
  HIP_CHECK(hipStreamSynchronize(s));
  C.plan = CoarsePlan{F, nBlocks, nLevels, C.nEdges, C.order.p, C.pos.p, C.levelPtr.p, C.levelCols.p, C.lvlBlkPtr.p,
                      C.lvlBlks.p, C.blkCol.p, C.blkRow.p, C.colPtr.p, C.rowPtr.p, C.rowBlk.p, C.updPtr.p, C.updA.p,
                      C.updB.p, C.edgeBlk.p, C.edgeFa.p, C.edgeFb.p, C.wPtr.p, C.wRow.p, C.wtPtr.p, C.wtBlk.p, C.wtCol.p, C.wtFrame.p,
                      C.wuPtr.p, C.wuL.p, C.wuW.p, nW, C.updBlk.p};
  C.valid = true;
  C.denseReady = false;
}

// ---- coarse graph sparsification ---------------------------------------------------------------------------
// The coarse factorisation is sparse-direct on the frame graph: its cost follows the FILL of that graph.  The reference
// sampler's hierarchical list (utils/frame_sampling.py:77-120: distance 2^l from every 2^(l-1)-th frame) eliminates with
// ~10 k block updates at 300 frames; the densified "~4k pairs" list of BASELINE.json (long-range pairs from nearly every
// frame) needs ~10^6 and a 44 ms factorisation per rebuild.  The coarse level is only a preconditioner, so it may be built
// on a SUBGRAPH: when the full graph's elimination exceeds a budget, a pair {a, b} at distance d stays in the coarse graph
// iff both frames are multiples of s(d) = the largest power of two <= d / 8 (>= 1) -- a nested, multi-scale subgraph,
// sparse for any flow list.  Dropped pairs are removed from the coarse operator altogether (k_coarse_edges: dropDiag),
// which keeps it the Galerkin operator of a sub-problem: SPD and consistent on the smooth drift modes.  Measured on the
// 4140-pair list (ms per LM iteration at the final level / PCG iterations per LM iteration): full graph 30.0 / 36
// (44 ms per factorisation), s(d) <= d/2 (the reference sampler's own 883-edge skeleton) 16.9 / 123, d/4 12.9 / 88,
// d/8 10.8 / 64, d/16 11.7 / 47.
static long long coarseEliminationUpdates(int F, const std::vector<std::pair<int, int>>& edgeList) {
  std::vector<std::set<int>> g(F);
  for (const auto& e : edgeList) { g[e.first].insert(e.second); g[e.second].insert(e.first); }
  std::set<std::pair<int, int>> queue;
  for (int v = 0; v < F; ++v) queue.insert({static_cast<int>(g[v].size()), v});
  long long updates = 0;
  while (!queue.empty()) {
    const int v = queue.begin()->second;
    queue.erase(queue.begin());
    const std::vector<int> nb(g[v].begin(), g[v].end());
    const long long sN = static_cast<long long>(nb.size());
    updates += sN * (sN + 1) / 2;
    if (updates > (1ll << 40)) break;
    for (int a : nb) queue.erase({static_cast<int>(g[a].size()), a});
    for (int a : nb) {
      g[a].erase(v);
      for (int b : nb)
        if (a != b) g[a].insert(b);
    }
    for (int a : nb) queue.insert({static_cast<int>(g[a].size()), a});
    g[v].clear();
  }
  return updates;
}
static bool coarseKeepsPair(int a, int b) {
  constexpr int shift = 3;  // (d/2: 123 PCG iterations per LM iteration on the 4140-pair list, d/4: 88, d/8: 64, d/16: 47 but a costlier factor)
  const int d = std::abs(a - b);
  int s2 = 1;
  while (s2 * 2 <= (d >> shift)) s2 *= 2;
  return (a % s2) == 0 && (b % s2) == 0;
}

// ---- compile the constraint table + work decomposition for a frame range -------------------------------
void compileTable(cvd_handle* h, const std::vector<int>& range, bool withTriplets, bool ignoreStatic) {
  std::vector<unsigned char> inRange(h->F, 0);
  for (int f : range) inRange[f] = 1;
  if (h->tableValid && inRange == h->tableRange && withTriplets == h->tableWithTriplets && ignoreStatic == h->tableIgnoresStatic) return;
  h->tableIgnoresStatic = ignoreStatic;
  hipStream_t s = h->stream;
  h->dInRange.upload(inRange.data(), inRange.size(), s);
  {
    std::vector<unsigned char> owner(h->F, 0);
    for (int f = 0; f < h->F; ++f) owner[f] = inRange[f] && (f % h->world == h->rank);
    h->dRegOwner.upload(owner.data(), owner.size(), s);
  }
  h->dCount.ensure(1);
  HIP_CHECK(hipMemsetAsync(h->dCount.p, 0, sizeof(unsigned long long), s));
  if (h->dense) {
    // no table: the kernels read the images; only the number of valid constraints is needed here
    if (h->C > 0) {
      const unsigned grid = static_cast<unsigned>((h->C + 255) / 256);
      hipLaunchKernelGGL(k_dense_count, dim3(grid), dim3(256), 0, s, makeTable(h), h->P, h->dInRange.p, h->dCount.p);
      HIP_CHECK(hipGetLastError());
    }
  } else {
    h->dNdc.ensure(std::max<long long>(h->C, 1));
    h->dDsrc.ensure(std::max<long long>(h->C, 1));
  }
  h->orderGx = h->orderGy = -1;  // (the ordered copy, if any, is stale: orderTable)
  h->tableOrdered = false;
  if (h->C > 0 && !h->dense) {
    const int bs = 256;
    const unsigned grid = static_cast<unsigned>((h->C + bs - 1) / bs);
    hipLaunchKernelGGL(k_build_table, dim3(grid), dim3(bs), 0, s, h->W, h->H, h->invAspect, h->C, h->dLoc.p,
                       h->dStatic.p, h->dCPair.p, h->dPairA.p, h->dPairB.p, h->dInRange.p, h->dDepth.p,
                       h->dNdc.p, h->dDsrc.p, h->dCount.p, ignoreStatic ? 1 : 0);
    HIP_CHECK(hipGetLastError());
  }
  if (h->dist()) commAllReduce(h, h->dCount.p, 1, CT_U64, s);
  unsigned long long nv = 0;
  HIP_CHECK(hipMemcpyAsync(&nv, h->dCount.p, sizeof(nv), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  h->numValid = static_cast<long long>(nv);

  // work items: one per UNDIRECTED pair {a < b} and chunk; each carries a slice of a->b and of b->a
  h->itemFa.clear();
  h->itemFb.clear();
  h->itemRange.clear();
  std::vector<std::vector<int>> frameItems(h->F), framePairs(h->F);
  std::map<std::pair<int, int>, std::array<int, 2>> edges;  // (min, max) -> {pair min->max, pair max->min}
  for (int p = 0; p < h->P; ++p) {
    const int a = h->pairA[p], b = h->pairB[p];
    if (!inRange[a] || !inRange[b] || a == b) continue;
    const long long n = h->pairOff[p + 1] - h->pairOff[p];
    if (n <= 0) continue;
    framePairs[a].push_back(p * 2 + 0);
    framePairs[b].push_back(p * 2 + 1);
    auto it = edges.find({std::min(a, b), std::max(a, b)});
    if (it == edges.end()) it = edges.insert({{std::min(a, b), std::max(a, b)}, {-1, -1}}).first;
    it->second[a < b ? 0 : 1] = p;
  }
  struct ItemDesc { int fa, fb; long long b0, e0, b1, e1; };
  std::vector<ItemDesc> itemList;
  // Constraints per direction and work item (list mode): 768 -- a whole pair of the sampled lists -- unless that would leave
  // workgroup slots idle (three 256-thread workgroups per CU): a rank of a pair-sharded run holds 1 / world of the pairs (259
  // of the benchmark's 2070 at 8 ranks: a third of the slots, every workgroup walking a whole pair), and small problems the
  // same.  The pairs are then cut finer, down to 128 constraints, until the items fill the device once.
  long long listChunk = kListChunk;
  if (!h->dense) {
    long long longest = 0;
    for (const auto& e : edges) {
      long long n0 = 0, n1 = 0;
      if (e.second[0] >= 0) n0 = h->pairOff[e.second[0] + 1] - h->pairOff[e.second[0]];
      if (e.second[1] >= 0) n1 = h->pairOff[e.second[1] + 1] - h->pairOff[e.second[1]];
      longest += std::max(n0, n1);
    }
    const long long slots = 3ll * h->numCU;
    if (longest / kListChunk + static_cast<long long>(edges.size()) < slots) {
      const long long want = (longest + slots - 1) / slots;
      listChunk = std::min<long long>(kListChunk, std::max<long long>(128, (want + 63) / 64 * 64));
    }
  }
  for (const auto& e : edges) {
    const int fa = e.first.first, fb = e.first.second;
    long long n0 = 0, n1 = 0, o0 = 0, o1 = 0;
    if (e.second[0] >= 0) { o0 = h->pairOff[e.second[0]]; n0 = h->pairOff[e.second[0] + 1] - o0; }
    if (e.second[1] >= 0) { o1 = h->pairOff[e.second[1]]; n1 = h->pairOff[e.second[1] + 1] - o1; }
    const long long chunk = h->dense ? kDenseChunk : listChunk;
    const long long nItems = std::max<long long>(1, (std::max(n0, n1) + chunk - 1) / chunk);
    const long long c0 = (n0 + nItems - 1) / nItems, c1 = (n1 + nItems - 1) / nItems;
    for (long long k = 0; k < nItems; ++k) {
      const long long b0 = o0 + std::min(n0, k * c0), e0 = o0 + std::min(n0, (k + 1) * c0);
      const long long b1 = o1 + std::min(n1, k * c1), e1 = o1 + std::min(n1, (k + 1) * c1);
      if (b0 >= e0 && b1 >= e1) continue;
      itemList.push_back({fa, fb, b0, e0, b1, e1});
    }
  }
  // Longest items first: the pair-major kernels run one workgroup per item in launch order, ~2.7 rounds of the device at the
  // benchmark's 2070 items -- with the short items last the final, partly filled round is short too.  (Stable: equal sizes
  // keep the frame-pair order.)
  std::stable_sort(itemList.begin(), itemList.end(), [](const ItemDesc& a, const ItemDesc& b) {
      return (a.e0 - a.b0) + (a.e1 - a.b1) > (b.e0 - b.b0) + (b.e1 - b.b1);
    });
  // (Measured and rejected, tools/mv_profile.py: the last, partly filled round of the 2070 items is 17 % of the hot product's
  // launch, but neither a persistent workgroup per slot walking a balanced item list -- the item loop costs the kernel
  // registers at its 170-VGPR budget: 52 -> 61 us -- nor halving the largest items up to a whole number of rounds -- 51.8 ->
  // 50.9 us: the workgroups of a round do not finish together anyway -- pays.)
  for (const ItemDesc& d : itemList) {
    const int item = static_cast<int>(h->itemFa.size());
    h->itemFa.push_back(d.fa);
    h->itemFb.push_back(d.fb);
    h->itemRange.insert(h->itemRange.end(), {d.b0, d.e0, d.b1, d.e1});
    frameItems[d.fa].push_back(item * 2 + 0);
    frameItems[d.fb].push_back(item * 2 + 1);
  }
  // ---- explicit-block mode of the dense mode (cvd_cross.h): one entry per undirected pair with both directions' whole
  // pixel ranges, two partial rows each, rows grouped by frame
  h->xFa.clear();
  h->xFb.clear();
  if (h->dense) {
    std::vector<long long> xRange;
    std::vector<std::vector<int>> frameRows(h->F);
    for (const auto& e : edges) {
      const int fa = e.first.first, fb = e.first.second;
      long long b0 = 0, e0 = 0, b1 = 0, e1 = 0;
      if (e.second[0] >= 0) { b0 = h->pairOff[e.second[0]]; e0 = h->pairOff[e.second[0] + 1]; }
      if (e.second[1] >= 0) { b1 = h->pairOff[e.second[1]]; e1 = h->pairOff[e.second[1] + 1]; }
      if (b0 >= e0 && b1 >= e1) continue;
      const int k = static_cast<int>(h->xFa.size());
      h->xFa.push_back(fa);
      h->xFb.push_back(fb);
      xRange.insert(xRange.end(), {b0, e0, b1, e1});
      frameRows[fa].push_back(k * 2 + 0);
      frameRows[fb].push_back(k * 2 + 1);
    }
    // one-walk assembly (cvd_dense_walk.h): one record per directed pair of the pair graph, and the two directions of every block
    {
      std::vector<int> dwPair, recOff(h->P + 1, 0), xDir;
      for (const auto& e : edges) {
        bool any = false;
        for (int dir = 0; dir < 2; ++dir) any = any || (e.second[dir] >= 0 && h->pairOff[e.second[dir] + 1] > h->pairOff[e.second[dir]]);
        if (!any) continue;
        for (int dir = 0; dir < 2; ++dir) {
          const int p = e.second[dir];
          const bool has = p >= 0 && h->pairOff[p + 1] > h->pairOff[p];
          xDir.push_back(has ? p : -1);
        }
      }
      for (int p = 0; p < h->P; ++p) {
        const int a = h->pairA[p], b = h->pairB[p];
        const long long n = h->pairOff[p + 1] - h->pairOff[p];
        recOff[p] = static_cast<int>(dwPair.size());
        if (!inRange[a] || !inRange[b] || a == b || n <= 0) continue;
        dwPair.push_back(p);
      }
      recOff[h->P] = static_cast<int>(dwPair.size());
      h->nDwRecords = static_cast<int>(dwPair.size());
      if (xDir.empty()) xDir.push_back(-1);
      if (dwPair.empty()) dwPair.push_back(0);
      h->dDwPair.upload(dwPair.data(), dwPair.size(), s);
      h->dDwRecOff.upload(recOff.data(), recOff.size(), s);
      h->dXDir.upload(xDir.data(), xDir.size(), s);
    }
    std::vector<int> xFiOff(h->F + 1, 0), xSlot(std::max<size_t>(1, h->xFa.size() * 2), 0), xDiagSlot(std::max(1, h->F), 0);
    // (one GPU: a frame's own block H_ff is one more row of its range, produced by the product kernel -- k_cross_matvec; a
    // pair-sharded run keeps H_ff p_f in the finish half of the frame's owner)
    h->xDiagRows = !h->dist();
    int row = 0;
    for (int f = 0; f < h->F; ++f) {
      for (int code : frameRows[f]) xSlot[code] = row++;
      if (h->xDiagRows) xDiagSlot[f] = row++;
      xFiOff[f + 1] = row;
    }
    h->xRows = row;
    h->dXDiagSlot.upload(xDiagSlot.data(), xDiagSlot.size(), s);
    h->dXFa.upload(h->xFa.data(), h->xFa.size(), s);
    h->dXFb.upload(h->xFb.data(), h->xFb.size(), s);
    h->dXRange.upload(xRange.data(), xRange.size(), s);
    h->dXSlot.upload(xSlot.data(), xSlot.size(), s);
    h->dXFiOff.upload(xFiOff.data(), xFiOff.size(), s);
  }
  // ---- scene-flow smoothness triplets: table, active groups (all three frames in range; groups are sharded
  // over the ranks like the per-frame regularisers), per-frame (group, role) lists and their partial-product rows
  h->tripActive.clear();
  h->numValidTrip = 0;
  std::vector<std::vector<int>> frameTrips(h->F);
  if (withTriplets) {
    std::map<int, int> groupOf;
    for (size_t g = 0; g < h->tripCenter.size(); ++g) groupOf[h->tripCenter[g]] = static_cast<int>(g);
    if (!range.empty()) {
      // reference lib/PoseOptimizer.cpp:1255-1262: every in-range consecutive triple needs its constraints
      for (int fr = range.front(); fr < range.back() - 1; ++fr) {
        if (!inRange[fr] || !inRange[fr + 1] || !inRange[fr + 2]) continue;
        auto itg = groupOf.find(fr + 1);
        if (itg == groupOf.end()) throw std::runtime_error("Missing triplet constraints.");
        const int g = itg->second;
        h->tripActive.push_back(g);
      }
    }
    h->dTNdc.ensure(static_cast<size_t>(std::max<long long>(h->tripC, 1)) * 3);
    h->dTDsrc.ensure(static_cast<size_t>(std::max<long long>(h->tripC, 1)) * 3);
    HIP_CHECK(hipMemsetAsync(h->dCount.p, 0, sizeof(unsigned long long), s));
    if (h->tripC > 0) {
      const unsigned grid = static_cast<unsigned>((h->tripC + 255) / 256);
      hipLaunchKernelGGL(k_build_triplet_table, dim3(grid), dim3(256), 0, s, h->W, h->H, h->invAspect, h->tripC,
                         h->dTLoc.p, h->dTGroupOfC.p, h->dTCenterAll.p, h->F, h->dInRange.p, h->dDepth.p, h->dTNdc.p,
                         h->dTDsrc.p, h->dCount.p);
      HIP_CHECK(hipGetLastError());
    }
    unsigned long long nvt = 0;
    HIP_CHECK(hipMemcpyAsync(&nvt, h->dCount.p, sizeof(nvt), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    h->numValidTrip = static_cast<long long>(nvt);
    // this rank's share of the groups
    std::vector<int> mine;
    for (size_t k = 0; k < h->tripActive.size(); ++k)
      if (static_cast<int>(k) % h->world == h->rank) mine.push_back(h->tripActive[k]);
    h->tripActive.swap(mine);
    const int pairCodes = static_cast<int>(h->itemFa.size()) * 2;
    for (size_t k = 0; k < h->tripActive.size(); ++k) {
      const int f1 = h->tripCenter[h->tripActive[k]];
      for (int role = 0; role < 3; ++role) {
        frameItems[f1 - 1 + role].push_back(pairCodes + static_cast<int>(k) * 3 + role);
        frameTrips[f1 - 1 + role].push_back((static_cast<int>(k) << 2) | role);
      }
    }
  }
  h->tableWithTriplets = withTriplets;
  h->coarse.valid = false;
  // The coarse level needs the frame graph of the whole problem.  One rank: the local items are the whole problem.
  // Several ranks: only with cvd_set_pair_graph (identical on all ranks); otherwise the level stays off.
  if (static_cast<size_t>(h->F) * kCB <= kCoarseMaxUnknowns &&
      (!h->dist() ? !h->itemFa.empty() : h->haveGlobalEdges)) {  // (rank-independent decision when sharded)
    std::map<std::pair<int, int>, int> edgeId;
    std::vector<std::pair<int, int>> edgeList;
    if (h->haveGlobalEdges) {
      for (const auto& e : h->globalEdges) {
        if (!inRange[e.first] || !inRange[e.second]) continue;
        edgeId.insert({e, static_cast<int>(edgeList.size())});
        edgeList.push_back(e);
      }
    }
    std::vector<int> itemEdge(h->itemFa.size());
    for (size_t i = 0; i < h->itemFa.size(); ++i) {
      const std::pair<int, int> key{h->itemFa[i], h->itemFb[i]};
      auto it = edgeId.find(key);
      if (it == edgeId.end()) {
        if (h->haveGlobalEdges) throw std::runtime_error("cvd_set_pair_graph: a frame pair with constraints is missing from the graph");
        it = edgeId.insert({key, static_cast<int>(edgeList.size())}).first;
        edgeList.push_back(key);
      }
      itemEdge[i] = it->second;
    }
    // sparsify the coarse graph when its elimination is too expensive (a function of the whole problem's pair graph
    // only: identical on all ranks of a sharded run)
    const long long updateBudget = h->opt.coarse_update_budget;
    h->coarse.sparsified = false;
    h->coarse.denseMode = false;
    const int denseMaxUnknowns = h->opt.coarse_dense_max_unknowns;
    const bool overBudget = coarseEliminationUpdates(h->F, edgeList) > updateBudget;
    // coarse_level 3: the temporal pose level (cvd_temporal.h) -- every pair kept, the node-reduced matrix inverted densely
    {
      const int stepP = std::max(2, h->opt.coarse_temporal_step);
      const bool fits = ((h->F - 1 + stepP - 1) / stepP + 1) * kCB <= kDenseCoarseMaxUnknowns;
      // (... and where it brings the PCG iteration inside the fused tail kernel's scope: cvd_solver_options::coarse_temporal_min_frames)
      const bool fusedScope = !h->dist() && h->opt.pcg_fused_tail != 0 && !h->forceGeneric && h->Bsz() <= 256 &&
                              h->opt.coarse_temporal_min_frames > 0 && h->F >= h->opt.coarse_temporal_min_frames &&
                              h->F + 2 * kTlMaxS <= 2 * h->numCU;
      h->coarse.temporalPose = fits && (h->opt.coarse_level == 3 || (overBudget && h->opt.coarse_over_budget == 0) ||
                                        (h->opt.coarse_over_budget == 0 && fusedScope));
    }
    if (h->coarse.temporalPose) {
      h->coarse.denseMode = true;  // (same exchange layout and in-line build as the dense exact level)
    } else if (overBudget && h->F * kCB <= denseMaxUnknowns) {
      h->coarse.denseMode = true;  // small enough to invert as a dense matrix: keeps every pair (cvd_coarse.h)
    } else if (overBudget) {
      std::vector<int> newId(edgeList.size(), -1);
      std::vector<std::pair<int, int>> kept;
      for (size_t e = 0; e < edgeList.size(); ++e)
        if (coarseKeepsPair(edgeList[e].first, edgeList[e].second)) {
          newId[e] = static_cast<int>(kept.size());
          kept.push_back(edgeList[e]);
        }
      for (auto& ie : itemEdge) ie = newId[ie];
      h->coarse.sparsified = kept.size() != edgeList.size();
      edgeList.swap(kept);
    }
    buildCoarsePlan(h, edgeList, itemEdge);
    if (h->coarse.temporalPose) poseTemporalPlan(h);
    if (h->dense) {  // edge block of every cross pair (cvd_cross.h: k_coarse_edges_cross)
      std::map<std::pair<int, int>, int> edgeOfPair;
      for (size_t i = 0; i < h->itemFa.size(); ++i) edgeOfPair[{h->itemFa[i], h->itemFb[i]}] = itemEdge[i];
      std::vector<int> pairEdge(std::max<size_t>(1, h->xFa.size()), -1);
      for (size_t k = 0; k < h->xFa.size(); ++k) {
        auto it = edgeOfPair.find({h->xFa[k], h->xFb[k]});
        if (it != edgeOfPair.end()) pairEdge[k] = it->second;
      }
      h->dXPairEdge.upload(pairEdge.data(), pairEdge.size(), s);
    }
  }
  std::vector<int> fiOff(h->F + 1, 0), fiList, fpOff(h->F + 1, 0), fpList;
  for (int f = 0; f < h->F; ++f) {
    fiOff[f + 1] = fiOff[f] + static_cast<int>(frameItems[f].size());
    fiList.insert(fiList.end(), frameItems[f].begin(), frameItems[f].end());
    fpOff[f + 1] = fpOff[f] + static_cast<int>(framePairs[f].size());
    fpList.insert(fpList.end(), framePairs[f].begin(), framePairs[f].end());
  }
  h->dItemFa.upload(h->itemFa.data(), h->itemFa.size(), s);
  h->dItemFb.upload(h->itemFb.data(), h->itemFb.size(), s);
  h->dItemRange.upload(h->itemRange.data(), h->itemRange.size(), s);
  std::vector<int> itemSlot(fiList.size(), 0);  // [item * 2 + side | pairs * 2 + group * 3 + role] -> row
  for (size_t e = 0; e < fiList.size(); ++e) itemSlot[fiList[e]] = static_cast<int>(e);
  h->qRows = static_cast<int>(fiList.size());
  h->dItemSlot.upload(itemSlot.data(), itemSlot.size(), s);
  if (withTriplets) {
    // compact per-rank group arrays in tripActive order: offsets / centre / rows; per-frame (group, role) lists
    const size_t nG = h->tripActive.size();
    std::vector<long long> tOff(2 * std::max<size_t>(nG, 1), 0);
    std::vector<int> tCen(std::max<size_t>(nG, 1), 0), tSlot(3 * std::max<size_t>(nG, 1), 0);
    const int pairCodes = static_cast<int>(h->itemFa.size()) * 2;
    for (size_t k = 0; k < nG; ++k) {
      const int g = h->tripActive[k];
      tOff[2 * k] = h->tripOff[g];
      tOff[2 * k + 1] = h->tripOff[g + 1];
      tCen[k] = h->tripCenter[g];
      for (int role = 0; role < 3; ++role) tSlot[3 * k + role] = itemSlot[pairCodes + static_cast<int>(k) * 3 + role];
    }
    std::vector<int> ftOff(h->F + 1, 0), ftList;
    for (int f = 0; f < h->F; ++f) {
      ftOff[f + 1] = ftOff[f] + static_cast<int>(frameTrips[f].size());
      ftList.insert(ftList.end(), frameTrips[f].begin(), frameTrips[f].end());
    }
    h->dTOff.upload(tOff.data(), tOff.size(), s);
    h->dTCenter.upload(tCen.data(), tCen.size(), s);
    h->dTSlot.upload(tSlot.data(), tSlot.size(), s);
    h->dFtOff.upload(ftOff.data(), ftOff.size(), s);
    h->dFtList.upload(ftList.data(), ftList.size(), s);
    h->dCostTrip.ensure(std::max<size_t>(nG, 1));
  }
  {
    // k_assemble_fast work list: units of <= kAsmUnit constraints, parts of <= capU units.  capU = the mean units
    // per frame (or per CU when there are fewer frames than CUs), so a frame of average size stays whole and
    // only the long-range hub frames of the hierarchical flow list are split.
    std::vector<int2> units;
    std::vector<int> fuOff(h->F + 1, 0);
    for (int f = 0; f < h->F; ++f) {
      for (const int code : framePairs[f]) {
        const long long n = h->pairOff[(code >> 1) + 1] - h->pairOff[code >> 1];
        for (long long o = 0; o < n; o += (h->dense ? kAsmUnitDense : kAsmUnit)) units.push_back(make_int2(code, static_cast<int>(o)));
      }
      fuOff[f + 1] = static_cast<int>(units.size());
    }
    int activeFrames = 0;
    for (int f = 0; f < h->F; ++f) activeFrames += inRange[f] ? 1 : 0;
    constexpr double partsPerCU = 1.0;  // (1.5 / 2 / 3 / 4 parts per CU measured on the 4140-pair set: 0.44 / 0.42 / 0.46 / 0.51 ms against 0.38)
    const long long denom = std::max<long long>(1, std::max<long long>(activeFrames, static_cast<long long>(partsPerCU * h->numCU)));
    const int capU = static_cast<int>(std::max<long long>(kAsmThreads / 64, (static_cast<long long>(units.size()) + denom - 1) / denom));
    std::vector<AsmPart> parts;
    int slots = 0;
    for (int f = 0; f < h->F; ++f) {
      // (frames outside the range have no entries: their part writes the zero block / gradient / cost)
      const int nu = fuOff[f + 1] - fuOff[f];
      const int np = std::max(1, (nu + capU - 1) / capU);
      const int per = (nu + np - 1) / np;
      for (int q = 0; q < np; ++q) {
        AsmPart a;
        a.frame = f;
        a.u0 = fuOff[f] + std::min(nu, q * per);
        a.u1 = fuOff[f] + std::min(nu, (q + 1) * per);
        a.part = q;
        a.nParts = np;
        a.slot0 = np > 1 ? slots : 0;
        parts.push_back(a);
      }
      if (np > 1) slots += np;
    }
    std::stable_sort(parts.begin(), parts.end(),
                     [](const AsmPart& a, const AsmPart& b) { return a.u1 - a.u0 > b.u1 - b.u0; });
    h->nAsmParts = static_cast<int>(parts.size());
    h->nAsmSlots = slots;
    h->dAsmParts.upload(parts.data(), parts.size(), s);
    h->dAsmUnits.upload(units.data(), units.size(), s);
    if (h->dAsmCount.n < static_cast<size_t>(h->F)) {
      h->dAsmCount.ensure(h->F);
      HIP_CHECK(hipMemsetAsync(h->dAsmCount.p, 0, sizeof(unsigned int) * h->F, s));
    }
  }
  h->dFiOff.upload(fiOff.data(), fiOff.size(), s);
  h->dFiList.upload(fiList.data(), fiList.size(), s);
  h->dFpOff.upload(fpOff.data(), fpOff.size(), s);
  h->dFpList.upload(fpList.data(), fpList.size(), s);
  HIP_CHECK(hipStreamSynchronize(s));
  h->tableRange = inRange;
  h->tableValid = true;
}
// The table in the order the pair-major kernels want for this problem's depth grid (k_order_table): a second copy of the
// table, rebuilt when the table or the grid changes (coarse-to-fine levels refine the grid).  Grids only -- a Global
// transform keeps its depth sums in registers -- and never the dense mode (no table).
void orderTable(cvd_handle* h, const Layout& L, int KD) {
  int gx = 0, gy = 0;
  if (!h->dense && h->C > 0 && h->opt.constraint_order != 0 && L.depthType == kDepthGrid && L.gz <= 1 && (KD == 4 || KD == 16) &&
      L.gx * L.gy <= kOrderMaxCells) {
    gx = L.gx;
    gy = L.gy;
  }
  if (h->orderGx == gx && h->orderGy == gy) return;
  h->orderGx = gx;
  h->orderGy = gy;
  h->tableOrdered = false;
  if (gx == 0) return;
  h->dNdcOrd.ensure(h->C);
  h->dDsrcOrd.ensure(h->C);
  const size_t lds = static_cast<size_t>(gx * gy + 1) * sizeof(int) + 3 * kOrderCap * sizeof(unsigned short);
  allowLds(k_order_table, lds);
  hipLaunchKernelGGL(k_order_table, dim3(h->P), dim3(64), lds, h->stream, h->dPairOff.p, gx, gy, L.maxcx, L.maxcy, h->dNdc.p,
                     h->dDsrc.p, h->dNdcOrd.p, h->dDsrcOrd.p);
  HIP_CHECK(hipGetLastError());
  h->tableOrdered = true;
}

// Pinned staging buffer `which` with room for n doubles (pageable transfers of the F x B vectors cost ~1 ms each).
double* pinnedStage(cvd_handle* h, int which, size_t n) {
  if (h->hStageN[which] < n) {
    if (h->hStage[which]) HIP_CHECK(hipHostFree(h->hStage[which]));
    h->hStage[which] = nullptr;
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hStage[which]), std::max<size_t>(n, 1) * sizeof(double)));
    h->hStageN[which] = n;
  }
  return h->hStage[which];
}

void uploadState(cvd_handle* h, const Layout& L, DevBuf<double>& dst) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  double* x = pinnedStage(h, 1, nAll);
  const int nD = L.nD, nS = L.nS;
  for (int f = 0; f < L.F; ++f) {
    double* xf = &x[static_cast<size_t>(f) * L.B];
    for (int i = 0; i < 7; ++i) xf[i] = h->poseParams[f][i];
    for (int i = 0; i < nD; ++i) xf[7 + i] = h->dparams[static_cast<size_t>(f) * nD + i];
    for (int i = 0; i < nS; ++i) xf[7 + nD + i] = h->sparams[static_cast<size_t>(f) * nS + i];
  }
  dst.upload(x, nAll, h->stream);
  HIP_CHECK(hipStreamSynchronize(h->stream));
}

void downloadState(cvd_handle* h, const Layout& L, const DevBuf<double>& src) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  double* x = pinnedStage(h, 0, nAll);
  src.download(x, nAll, h->stream);
  HIP_CHECK(hipStreamSynchronize(h->stream));
  const int nD = L.nD, nS = L.nS;
  for (int f = 0; f < L.F; ++f) {
    const double* xf = &x[static_cast<size_t>(f) * L.B];
    for (int i = 0; i < 7; ++i) h->poseParams[f][i] = xf[i];
    for (int i = 0; i < nD; ++i) h->dparams[static_cast<size_t>(f) * nD + i] = xf[7 + i];
    for (int i = 0; i < nS; ++i) h->sparams[static_cast<size_t>(f) * nS + i] = xf[7 + nD + i];
  }
}

void buildMask(cvd_handle* h, const Layout& L, const cvd_opt_params& p, ProblemKind kind,
                      const std::vector<int>& range) {
  const size_t nAll = static_cast<size_t>(L.F) * L.B;
  HIP_CHECK(hipStreamSynchronize(h->stream));  // an earlier transfer may still read the staging buffer
  double* m = pinnedStage(h, 0, nAll);
  std::fill(m, m + nAll, 0.0);
  for (int f : range) {
    double* mf = &m[static_cast<size_t>(f) * L.B];
    const bool poseFree = (kind == PK_POSE_STEP) && !p.fix_poses;
    for (int i = 0; i < 6; ++i) mf[i] = poseFree ? 1.0 : 0.0;
    mf[6] = (kind == PK_POSE_STEP && p.intr_opt != CVD_INTR_FIXED) ? 1.0 : 0.0;
    const bool depthFree = (kind == PK_NORMALIZE) || !p.fix_depth_xforms;
    for (int i = 0; i < L.nD; ++i) mf[7 + i] = depthFree ? 1.0 : 0.0;
    const bool spatialFree = (kind == PK_POSE_STEP) && !p.fix_spatial_xforms;
    for (int i = 0; i < L.nS; ++i) mf[7 + L.nD + i] = spatialFree ? 1.0 : 0.0;
  }
  h->dMask.upload(m, nAll, h->stream);
}

void ensureBuffers(Ctx& c) {
  cvd_handle* h = c.h;
  const size_t n = c.n;
  const size_t B = c.L.B;
  h->dX.ensure(n); h->dXc.ensure(n); h->dG.ensure(n); h->dLam.ensure(n); h->dScale.ensure(n);
  // (sharded: the vectors the owner-sharded PCG iteration reduce-scatters / all-gathers in place hold world x chunk frames)
  const size_t nVec = static_cast<size_t>(h->framesPadded()) * B;
  if (h->dist() && (h->dDx.n < nVec || h->dZ.n < nVec)) {
    h->dDx.ensure(nVec); h->dR.ensure(nVec); h->dZ.ensure(nVec);
    HIP_CHECK(hipMemsetAsync(h->dDx.p, 0, nVec * sizeof(double), h->stream));
    HIP_CHECK(hipMemsetAsync(h->dR.p, 0, nVec * sizeof(double), h->stream));
    HIP_CHECK(hipMemsetAsync(h->dZ.p, 0, nVec * sizeof(double), h->stream));
  }
  h->dDx.ensure(nVec); h->dR.ensure(nVec); h->dR1.ensure(n); h->dZ.ensure(nVec); h->dP0.ensure(n); h->dP1.ensure(n);
  // (+ [Z^T q | p.q | third level's restricted products]: the fused exchange of the pair-sharded mode)
  h->dQ.ensure(nVec + static_cast<size_t>(c.L.F) * (kCB + kTlMaxS) + 8);
  h->dOwnerScal.ensure(2 * static_cast<size_t>(std::max(1, h->world)));
  h->dHd.ensure(n);
  {
    // (sharded mode: room for world x chunk frames so that the reduce-scatter / all-gather chunks are equal; the tail
    // frames are zero and stay zero)
    const size_t nPad = static_cast<size_t>(h->framesPadded()) * B;
    const bool grow = h->dH.n < nPad * B;
    h->dH.ensure(nPad * B); h->dMinv.ensure(nPad * B + 4); h->dHd.ensure(nPad);  // (+4: k_cg_update's 16-byte loads at the last row's end)
    if (h->dist() && (grow || nPad > n)) {
      HIP_CHECK(hipMemsetAsync(h->dH.p, 0, nPad * B * sizeof(double), h->stream));
      HIP_CHECK(hipMemsetAsync(h->dMinv.p, 0, nPad * B * sizeof(float), h->stream));
      HIP_CHECK(hipMemsetAsync(h->dHd.p, 0, nPad * sizeof(double), h->stream));
    }
  }
  h->dQPart.ensure(std::max<size_t>(1, static_cast<size_t>(std::max(std::max(h->qRows, c.nItems * 2), c.cross ? h->xRows : 0)) * B));
  h->dFdot.ensure(static_cast<size_t>(c.L.F) * 4 + 64);  // (+64: the published p.q sum of k_pcg_tail, in a line of its own)
  h->dCostItem.ensure(std::max(1, c.nItems));
  h->dCostFrame.ensure(c.L.F);
  h->dFocal.ensure(static_cast<size_t>(c.L.F) * 2);
  h->dScal.ensure(S_COUNT);
  h->dFc.ensure(c.L.F);
  h->dFail.ensure(1);
  if (!h->dCounters.p) {
    h->dCounters.ensure(8);
    HIP_CHECK(hipMemsetAsync(h->dCounters.p, 0, 8 * sizeof(unsigned int), h->stream));
  }
  if (!h->hScal) HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hScal), (S_COUNT + 1) * sizeof(double)));  // (+ 1: the end-of-solve status word)
  if (!h->hPcg) {
    HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->hPcg), 16 * sizeof(double)));
    for (auto& e : h->pcgEvent) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
}
// Median of every frame's source depth (reference lib/PoseOptimizer.cpp:1363-1375: std::nth_element at size / 2), the
// reference value of the scale regulariser.  On the device: one segmented radix sort of the depth maps that are resident
// anyway, element n / 2 of every sorted frame -- the same order statistic, bit for bit.  (The host nth_element this
// replaces cost 0.25 ms per 384x224 frame inside cvd_set_depth: 70 of the 89 ms a 300-frame upload took.)
__global__ void k_pick_median(const float* __restrict__ sorted, size_t n, int first, int count, float* __restrict__ median) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) median[first + i] = sorted[static_cast<size_t>(i) * n + n / 2];
}
void refreshMedians(cvd_handle* h) {
  if (!h->medianDirty) return;
  hipStream_t s = h->stream;
  const size_t n = static_cast<size_t>(h->W) * h->H;
  const int F = h->F;
  h->dMedian.ensure(F);
  const int PB = static_cast<int>(std::max<size_t>(1, std::min<size_t>(F, (size_t(1) << 28) / (n * sizeof(float)))));  // <= 256 MiB
  if (static_cast<size_t>(PB) * n > 0xFFFFFFFFull) throw std::runtime_error("depth maps too large for the median sort");
  DevBuf<float> sorted;
  DevBuf<unsigned int> seg;
  DevBuf<unsigned char> tmp;
  sorted.ensure(static_cast<size_t>(PB) * n);
  std::vector<unsigned int> segH(PB + 1);
  for (int i = 0; i <= PB; ++i) segH[i] = static_cast<unsigned int>(static_cast<size_t>(i) * n);
  seg.upload(segH.data(), segH.size(), s);
  size_t tmpBytes = 0;
  HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, tmpBytes, h->dDepth.p, sorted.p, static_cast<unsigned int>(PB * n),
                                               static_cast<unsigned int>(PB), seg.p, seg.p + 1, 0, 32, s));
  tmp.ensure(tmpBytes);
  for (int p0 = 0; p0 < F; p0 += PB) {
    const int nb = std::min(PB, F - p0);
    size_t tb = tmpBytes;
    HIP_CHECK(rocprim::segmented_radix_sort_keys(tmp.p, tb, h->dDepth.p + static_cast<size_t>(p0) * n, sorted.p,
                                                 static_cast<unsigned int>(nb * n), static_cast<unsigned int>(nb), seg.p,
                                                 seg.p + 1, 0, 32, s));
    hipLaunchKernelGGL(k_pick_median, dim3((nb + 63) / 64), dim3(64), 0, s, sorted.p, n, p0, nb, h->dMedian.p);
    HIP_CHECK(hipGetLastError());
  }
  HIP_CHECK(hipStreamSynchronize(s));  // (the temporaries go out of scope)
  h->medianDirty = false;
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_setup() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_pick_median));
}

}  // namespace cvd
