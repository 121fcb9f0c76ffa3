// oracle/cvd_oracle.cpp
//
// *** TEST INFRASTRUCTURE ONLY -- never imported, linked or executed by the product path. ***
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
//
// CPU restatement (double precision, forward-mode dual numbers in passes of 4, Ceres-default
// Levenberg-Marquardt with an exact Cholesky solve) of the geometric-consistency optimizer of
// facebookresearch/robust_cvd:
//     lib/PoseOptimizer.cpp, lib/DepthMapTransform.cpp, lib/ValueTransform.h, lib/Processor.cpp:888-1013
// Each function cites the reference file:line it follows (paths relative to the reference root).
//
// PARITY UNPINNED: the reference cannot be built here (Ceres / Eigen / OpenCV / glog / fmt / boost are
// neither vendored nor installed) and ships no tests, golden vectors or fixtures for this
// path.  The arithmetic that lives in Ceres (version unpinned by the reference's CMake:
// `find_package(Ceres REQUIRED)`, lib/CMakeLists.txt:35) is restated from its published algorithm:
//   ceres/rotation.h        AngleAxisRotatePoint, RotationMatrixToAngleAxis, AngleAxisToRotationMatrix
//   ceres/jet.h             Jet<double,4> (oracle/jet.h)
//   ceres/loss_function.cc  CauchyLoss, ScaledLoss;  ceres/corrector.cc  Corrector
//   ceres/trust_region_minimizer.cc + levenberg_marquardt_strategy.cc  (default Solver::Options)
// The oracle is pinned instead by independent cross-checks in tests/ (central finite differences,
// closed forms, zero-noise ground-truth recovery, scipy least_squares on the same residuals) and, where the
// reference holds a second statement of the same arithmetic in Python, by running THAT: the ReproDisparity
// residuals of every constraint and their Jacobian columns for pose and focal length against
// utils/geometry.py:62-166 + loss/consistency_loss.py:27-199 at random non-converged states
// (tests/test_reference_residuals.py, cvdo_static_residuals), the output conventions against
// VideoDataset.update_poses (tests/test_reference_reprojection.py), the pair sampler against
// utils/frame_sampling.py (tests/test_synth.py).  What stays unpinned is the Ceres SOLVE.
//
// Build: oracle/Makefile -> oracle/_build/libcvd_oracle.so  (g++ -O2 -ffp-contract=off -fopenmp)

#include <algorithm>
#include <array>
#include <chrono>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/cvd_types.h"
#include "block_sparse.h"
#include "jet.h"

namespace cvdo {

static double nowSeconds() {
  using namespace std::chrono;
  return duration<double>(steady_clock::now().time_since_epoch()).count();
}

// =====================================================================================================
// ceres/rotation.h restatement
// =====================================================================================================

// ceres::AngleAxisRotatePoint (used at reference lib/PoseOptimizer.cpp:185,211).
template <typename T>
void angleAxisRotatePoint(const T aa[3], const T pt[3], T result[3]) {
  const T theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (scalar(theta2) > std::numeric_limits<double>::epsilon()) {
    const T theta = tsqrt(theta2);
    const T costheta = tcos(theta);
    const T sintheta = tsin(theta);
    const T theta_inverse = T(1.0) / theta;
    const T w[3] = {aa[0] * theta_inverse, aa[1] * theta_inverse, aa[2] * theta_inverse};
    const T w_cross_pt[3] = {w[1] * pt[2] - w[2] * pt[1], w[2] * pt[0] - w[0] * pt[2],
                             w[0] * pt[1] - w[1] * pt[0]};
    const T tmp = (w[0] * pt[0] + w[1] * pt[1] + w[2] * pt[2]) * (T(1.0) - costheta);
    result[0] = pt[0] * costheta + w_cross_pt[0] * sintheta + w[0] * tmp;
    result[1] = pt[1] * costheta + w_cross_pt[1] * sintheta + w[1] * tmp;
    result[2] = pt[2] * costheta + w_cross_pt[2] * sintheta + w[2] * tmp;
  } else {
    // Near zero: first-order Taylor R ~ I + [aa]x. Every frame starts exactly here (identity poses).
    const T w_cross_pt[3] = {aa[1] * pt[2] - aa[2] * pt[1], aa[2] * pt[0] - aa[0] * pt[2],
                             aa[0] * pt[1] - aa[1] * pt[0]};
    result[0] = pt[0] + w_cross_pt[0];
    result[1] = pt[1] + w_cross_pt[1];
    result[2] = pt[2] + w_cross_pt[2];
  }
}

// ceres::RotationMatrixToAngleAxis on a column-major 3x3 (reference lib/PoseOptimizer.cpp:779):
// RotationMatrixToQuaternion followed by QuaternionToAngleAxis.
static void rotationMatrixToAngleAxis(const double R[9] /*col-major*/, double aa[3]) {
  auto M = [&](int r, int c) { return R[r + 3 * c]; };
  double q[4];
  const double trace = M(0, 0) + M(1, 1) + M(2, 2);
  if (trace >= 0.0) {
    double t = std::sqrt(trace + 1.0);
    q[0] = 0.5 * t;
    t = 0.5 / t;
    q[1] = (M(2, 1) - M(1, 2)) * t;
    q[2] = (M(0, 2) - M(2, 0)) * t;
    q[3] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    double t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    q[i + 1] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M(k, j) - M(j, k)) * t;
    q[j + 1] = (M(j, i) + M(i, j)) * t;
    q[k + 1] = (M(k, i) + M(i, k)) * t;
  }
  const double sin_squared_theta = q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  if (sin_squared_theta > 0.0) {
    const double sin_theta = std::sqrt(sin_squared_theta);
    const double cos_theta = q[0];
    const double two_theta = 2.0 * ((cos_theta < 0.0) ? std::atan2(-sin_theta, -cos_theta)
                                                       : std::atan2(sin_theta, cos_theta));
    const double k = two_theta / sin_theta;
    aa[0] = q[1] * k;
    aa[1] = q[2] * k;
    aa[2] = q[3] * k;
  } else {
    aa[0] = q[1] * 2.0;
    aa[1] = q[2] * 2.0;
    aa[2] = q[3] * 2.0;
  }
}

// ceres::AngleAxisToRotationMatrix, column-major output (reference lib/PoseOptimizer.cpp:972).
static void angleAxisToRotationMatrix(const double aa[3], double R[9] /*col-major*/) {
  auto M = [&](int r, int c) -> double& { return R[r + 3 * c]; };
  const double theta2 = aa[0] * aa[0] + aa[1] * aa[1] + aa[2] * aa[2];
  if (theta2 > std::numeric_limits<double>::epsilon()) {
    const double theta = std::sqrt(theta2);
    const double wx = aa[0] / theta, wy = aa[1] / theta, wz = aa[2] / theta;
    const double c = std::cos(theta), s = std::sin(theta);
    M(0, 0) = c + wx * wx * (1.0 - c);
    M(1, 0) = wz * s + wx * wy * (1.0 - c);
    M(2, 0) = -wy * s + wx * wz * (1.0 - c);
    M(0, 1) = wx * wy * (1.0 - c) - wz * s;
    M(1, 1) = c + wy * wy * (1.0 - c);
    M(2, 1) = wx * s + wy * wz * (1.0 - c);
    M(0, 2) = wy * s + wx * wz * (1.0 - c);
    M(1, 2) = -wx * s + wy * wz * (1.0 - c);
    M(2, 2) = c + wz * wz * (1.0 - c);
  } else {
    M(0, 0) = 1.0;  M(1, 0) = aa[2];  M(2, 0) = -aa[1];
    M(0, 1) = -aa[2]; M(1, 1) = 1.0;  M(2, 1) = aa[0];
    M(0, 2) = aa[1]; M(1, 2) = -aa[0]; M(2, 2) = 1.0;
  }
}

// Eigen::Quaterniond(Matrix3d) (reference lib/PoseOptimizer.cpp:974); q = (x, y, z, w).
static void rotationMatrixToEigenQuaternion(const double R[9] /*col-major*/, double q[4]) {
  auto M = [&](int r, int c) { return R[r + 3 * c]; };
  double t = M(0, 0) + M(1, 1) + M(2, 2);
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (M(2, 1) - M(1, 2)) * t;
    q[1] = (M(0, 2) - M(2, 0)) * t;
    q[2] = (M(1, 0) - M(0, 1)) * t;
  } else {
    int i = 0;
    if (M(1, 1) > M(0, 0)) i = 1;
    if (M(2, 2) > M(i, i)) i = 2;
    const int j = (i + 1) % 3;
    const int k = (j + 1) % 3;
    t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (M(k, j) - M(j, k)) * t;
    q[j] = (M(j, i) + M(i, j)) * t;
    q[k] = (M(k, i) + M(i, k)) * t;
  }
}

// Eigen quaternion * vector (reference lib/PoseOptimizer.cpp:769-772): v + w*uv + qv x uv, uv = 2 qv x v.
static void quatRotate(const double q[4] /*x,y,z,w*/, const double v[3], double out[3]) {
  const double uv[3] = {2.0 * (q[1] * v[2] - q[2] * v[1]), 2.0 * (q[2] * v[0] - q[0] * v[2]),
                        2.0 * (q[0] * v[1] - q[1] * v[0])};
  out[0] = v[0] + q[3] * uv[0] + (q[1] * uv[2] - q[2] * uv[1]);
  out[1] = v[1] + q[3] * uv[1] + (q[2] * uv[0] - q[0] * uv[2]);
  out[2] = v[2] + q[3] * uv[2] + (q[0] * uv[1] - q[1] * uv[0]);
}

// =====================================================================================================
// Value transform (reference lib/ValueTransform.h:57-94)
// =====================================================================================================
static int valueNumParams(int type) {
  if (type == CVD_VALUE_SCALE) return 1;
  if (type == CVD_VALUE_SCALE_SHIFT) return 2;
  throw std::runtime_error("Invalid value transform.");
}
template <typename T>
static T valueXform(int type, const T& src, const T* p) {
  if (type == CVD_VALUE_SCALE) return src * p[0];
  return (src * p[0]) + p[1];
}

// =====================================================================================================
// Transforms and their per-sample gathers (reference lib/DepthMapTransform.cpp)
// =====================================================================================================

// cubicSpline, reference lib/DepthMapTransform.cpp:671-678
static void cubicSpline(std::array<double, 4>& w, const double t) {
  const double t2 = t * t;
  const double t3 = t2 * t;
  w[0] = -0.5 * t3 + t2 - 0.5 * t;
  w[1] = 1.5 * t3 - 2.5 * t2 + 1.0;
  w[2] = -1.5 * t3 + 2.0 * t2 + 0.5 * t;
  w[3] = 0.5 * t3 - 0.5 * t2;
}

struct Gather {
  int n = 0;
  int idx[16];  // parameter-block index inside the transform (vertex id)
  double w[16];
};

struct Xform {
  cvd_xform_desc desc{};
  std::vector<double> params;
  int blockSize = 0;
  int numBlocks = 0;

  bool isDepth() const { return desc.type == CVD_XFORM_DEPTH; }

  // createDepthXform / createSpatialXform, reference lib/DepthMapTransform.cpp:1435-1491
  static Xform create(const cvd_xform_desc& d) {
    Xform x;
    x.desc = d;
    if (d.type == CVD_XFORM_DEPTH) {
      switch (d.depth_type) {
        case CVD_DEPTH_IDENTITY:
          x.blockSize = 0;
          x.numBlocks = 0;
          break;
        case CVD_DEPTH_GLOBAL:  // :526-534
          x.blockSize = valueNumParams(d.value_xform);
          x.numBlocks = 1;
          x.params.assign(x.blockSize, 1.0);
          break;
        case CVD_DEPTH_GRID: {  // :682-737
          const int gx = d.grid_size[0], gy = d.grid_size[1], gz = d.grid_size[2];
          if (gx > 1 || gy > 1) {
            if (gx < 2 || gy < 2)
              throw std::runtime_error(
                  "Spatial grid transforms must have at least two rows and columns, respectively.");
          }
          x.blockSize = valueNumParams(d.value_xform);
          const int np = x.blockSize * gx * gy * gz;
          if (np <= 1) throw std::runtime_error("Grid transform cannot have an empty grid.");
          if (gz > 1) {
            if (d.depth_min_max[0] <= 0.0 || d.depth_min_max[1] <= 0.0)
              throw std::runtime_error("Depth values must be positive.");
            if (d.depth_min_max[1] - d.depth_min_max[0] <= 0.0)
              throw std::runtime_error("Depth range must be positive.");
          }
          x.numBlocks = gx * gy * gz;
          x.params.assign(np, 1.0);
          break;
        }
        default:
          throw std::runtime_error("Invalid depth transform type.");
      }
    } else if (d.type == CVD_XFORM_SPATIAL) {
      x.blockSize = 2;
      switch (d.spatial_type) {
        case CVD_SPATIAL_IDENTITY:
          x.blockSize = 0;
          x.numBlocks = 0;
          break;
        case CVD_SPATIAL_VERTICAL_LINEAR:  // :1098-1105
          x.numBlocks = 2;
          break;
        case CVD_SPATIAL_CORNERS_BILINEAR:  // :1172-1179
          x.numBlocks = 4;
          break;
        case CVD_SPATIAL_BILINEAR_GRID:
        case CVD_SPATIAL_BICUBIC_GRID:  // :1346-1363
          if (d.grid_size[1] < 2 || d.grid_size[0] < 2)
            throw std::logic_error("Need at least two rows and columns in depth transform grid.");
          x.numBlocks = d.grid_size[0] * d.grid_size[1];
          break;
        default:
          throw std::runtime_error("Invalid spatial transform type.");
      }
      x.params.assign(static_cast<size_t>(x.numBlocks) * x.blockSize, 0.0);
    } else {
      throw std::runtime_error("Invalid transform type.");
    }
    return x;
  }

  // Cell coordinates shared by every grid gather, reference lib/DepthMapTransform.cpp:750-764.
  static void cell(const float loc, const int g, int& i, double& r) {
    const double maxc = std::nextafter(static_cast<double>(g - 1), 0.0);
    const double s = std::min(std::max((loc + 1.0) * (g - 1) / 2.0, 0.0), maxc);
    i = static_cast<int>(s);
    r = s - i;
  }

  // GridDepthXform::linearGather, reference lib/DepthMapTransform.cpp:739-851
  void linearGather(const float srcDepth, const float lx, const float ly, Gather& g) const {
    const int gx = desc.grid_size[0], gy = desc.grid_size[1], gz = desc.grid_size[2];
    const bool spatial = gx > 1;
    const bool depthWise = gz > 1;
    if (blockSize != 1 && (spatial || depthWise)) {
      // The reference indexes `&params_[i]` (not i*N) here (:829-832): with ScaleShift the blocks alias
      // and ceres::Problem aborts on overlapping parameter blocks. Not a defined configuration.
      throw std::runtime_error(
          "Linear grid gather is only defined for 1-parameter value transforms (reference :829).");
    }
    int ix = 0, iy = 0, iz = 0;
    double rx = 0, ry = 0, rz = 0;
    if (spatial) {
      cell(lx, gx, ix, rx);
      cell(ly, gy, iy, ry);
    }
    if (depthWise) {
      const double dispMin = 1.0 / desc.depth_min_max[1];
      const double dispMax = 1.0 / desc.depth_min_max[0];
      const double interval = (dispMax - dispMin) / (gz - 1);
      const double maxz = std::nextafter(static_cast<double>(gz - 1), 0.0);
      const double srcDisparity = 1.0 / static_cast<double>(srcDepth);
      const double sz = std::min(std::max((srcDisparity - dispMin) / interval, 0.0), maxz);
      iz = static_cast<int>(sz);
      rz = sz - iz;
    }
    const int ys = gx, zs = gx * gy;
    if (spatial && depthWise) {
      g.n = 8;
      const int i0 = ix + iy * ys + iz * zs;
      const int ids[8] = {i0, i0 + 1, i0 + ys, i0 + ys + 1, i0 + zs, i0 + zs + 1, i0 + zs + ys,
                          i0 + zs + ys + 1};
      const double ws[8] = {(1.0 - rx) * (1.0 - ry) * (1.0 - rz), rx * (1.0 - ry) * (1.0 - rz),
                            (1.0 - rx) * ry * (1.0 - rz),         rx * ry * (1.0 - rz),
                            (1.0 - rx) * (1.0 - ry) * rz,         rx * (1.0 - ry) * rz,
                            (1.0 - rx) * ry * rz,                 rx * ry * rz};
      for (int k = 0; k < 8; ++k) { g.idx[k] = ids[k]; g.w[k] = ws[k]; }
    } else if (spatial) {
      g.n = 4;
      const int i0 = ix + iy * ys;
      g.idx[0] = i0;          g.w[0] = (1.0 - rx) * (1.0 - ry);
      g.idx[1] = i0 + 1;      g.w[1] = rx * (1.0 - ry);
      g.idx[2] = i0 + ys;     g.w[2] = (1.0 - rx) * ry;
      g.idx[3] = i0 + ys + 1; g.w[3] = rx * ry;
    } else if (depthWise) {
      g.n = 2;
      g.idx[0] = iz;     g.w[0] = 1.0 - rz;
      g.idx[1] = iz + 1; g.w[1] = rz;
    } else {
      g.n = 0;
    }
  }

  // Shared 2-D cubic gather with border folding: GridDepthXform::cubicGather
  // (reference lib/DepthMapTransform.cpp:853-948, gridSize.z == 1 only: quirk q3 -- wz is never applied
  // and wx/wy are uninitialised for z-only grids) and bicubicSpatialGridGather (:1288-1343).
  static void cubicGather2d(const float lx, const float ly, const int gx, const int gy, Gather& g) {
    int ix, iy;
    double rx, ry;
    cell(lx, gx, ix, rx);
    cell(ly, gy, iy, ry);
    std::array<double, 4> wx, wy;
    cubicSpline(wx, rx);
    cubicSpline(wy, ry);
    const int x0 = (ix == 0 ? 1 : 0);
    const int x1 = (ix == gx - 2 ? 3 : 4);
    const int y0 = (iy == 0 ? 1 : 0);
    const int y1 = (iy == gy - 2 ? 3 : 4);
    const int xstride = x1 - x0;
    const int ystride = y1 - y0;
    g.n = 0;
    for (int y = y0; y < y1; ++y) {
      const int py = iy - 1 + y;
      for (int x = x0; x < x1; ++x) {
        const int px = ix - 1 + x;
        g.idx[g.n] = px + py * gx;
        g.w[g.n] = 0.0;
        ++g.n;
      }
    }
    for (int y = 0; y < 4; ++y) {
      for (int x = 0; x < 4; ++x) {
        const int cx = std::min(std::max(x - x0, 0), xstride - 1);
        const int cy = std::min(std::max(y - y0, 0), ystride - 1);
        g.w[cx + cy * xstride] += wx[x] * wy[y];
      }
    }
  }

  // DepthXform::createFunctor: which blocks / weights a sample at `loc` with `srcDepth` depends on.
  // reference lib/DepthMapTransform.cpp:483-486 (identity), :536-540 (global), :1016-1027 (grid)
  void depthGather(const float srcDepth, const float lx, const float ly, Gather& g) const {
    switch (desc.depth_type) {
      case CVD_DEPTH_IDENTITY:
        g.n = 0;
        break;
      case CVD_DEPTH_GLOBAL:
        g.n = 1;
        g.idx[0] = 0;
        g.w[0] = 1.0;
        break;
      case CVD_DEPTH_GRID:
        if (desc.cubic_interpolation) {
          if (desc.grid_size[2] > 1 || desc.grid_size[0] < 2)
            throw std::runtime_error(
                "Cubic depth grids are only defined for gridSize.z == 1 (reference quirk q3).");
          cubicGather2d(lx, ly, desc.grid_size[0], desc.grid_size[1], g);
        } else {
          linearGather(srcDepth, lx, ly, g);
        }
        break;
      default:
        throw std::runtime_error("Invalid depth transform type.");
    }
  }

  // SpatialXform::createFunctor, reference lib/DepthMapTransform.cpp:1053-1056, 1107-1114, 1181-1191,
  // 1253-1286, 1288-1343.
  void spatialGather(const float lx, const float ly, Gather& g) const {
    switch (desc.spatial_type) {
      case CVD_SPATIAL_IDENTITY:
        g.n = 0;
        break;
      case CVD_SPATIAL_VERTICAL_LINEAR: {
        const double w0 = 0.5 + 0.5 * ly;
        g.n = 2;
        g.idx[0] = 0; g.w[0] = w0;
        g.idx[1] = 1; g.w[1] = 1.0 - w0;
        break;
      }
      case CVD_SPATIAL_CORNERS_BILINEAR: {
        const double wx = 0.5 + 0.5 * lx;
        const double wy = 0.5 + 0.5 * ly;
        g.n = 4;
        g.idx[0] = 0; g.w[0] = wx * wy;
        g.idx[1] = 1; g.w[1] = (1.0 - wx) * wy;
        g.idx[2] = 2; g.w[2] = wx * (1.0 - wy);
        g.idx[3] = 3; g.w[3] = (1.0 - wx) * (1.0 - wy);
        break;
      }
      case CVD_SPATIAL_BILINEAR_GRID: {
        const int gx = desc.grid_size[0], gy = desc.grid_size[1];
        int ix, iy;
        double rx, ry;
        cell(lx, gx, ix, rx);
        cell(ly, gy, iy, ry);
        g.n = 4;
        const int i0 = ix + iy * gx;
        g.idx[0] = i0;          g.w[0] = (1.0 - rx) * (1.0 - ry);
        g.idx[1] = i0 + 1;      g.w[1] = rx * (1.0 - ry);
        g.idx[2] = i0 + gx;     g.w[2] = (1.0 - rx) * ry;
        g.idx[3] = i0 + gx + 1; g.w[3] = rx * ry;
        break;
      }
      case CVD_SPATIAL_BICUBIC_GRID:
        cubicGather2d(lx, ly, desc.grid_size[0], desc.grid_size[1], g);
        break;
      default:
        throw std::runtime_error("Invalid spatial transform type.");
    }
  }

  // numDeformationCostResiduals: reference lib/DepthMapTransform.cpp:996-1002 (depth grid),
  // :1116-1118, :1193-1195, :1365-1367 (spatial); 0 for everything else (DepthMapTransform.h:165).
  int numDeformationResiduals() const {
    if (isDepth()) {
      if (desc.depth_type != CVD_DEPTH_GRID) return 0;
      const int X = desc.grid_size[0], Y = desc.grid_size[1], Z = desc.grid_size[2];
      const int edges = (X - 1) * Y * Z + X * (Y - 1) * Z + X * Y * (Z - 1);
      return edges * blockSize;
    }
    return numBlocks * blockSize;  // paramsToResiduals (:60-70); Identity has 0 blocks
  }

  // computeDeformationCost: computeGridDeformationCost (reference lib/DepthMapTransform.cpp:631-667) for
  // depth grids, paramsToResiduals (:60-70) for spatial transforms.
  template <typename T>
  void deformationCost(T const* const* p, T* residuals) const {
    if (!isDepth()) {
      int count = 0;
      for (int b = 0; b < numBlocks; ++b)
        for (int i = 0; i < blockSize; ++i) residuals[count++] = p[b][i];
      return;
    }
    T* out = residuals;
    const int X = desc.grid_size[0], Y = desc.grid_size[1], Z = desc.grid_size[2];
    const int yStride = X, zStride = X * Y;
    for (int z = 0; z < Z; ++z) {
      for (int y = 0; y < Y; ++y) {
        for (int x = 0; x < X; ++x) {
          T const* thisBlock = p[x + y * yStride + z * zStride];
          auto addResidual = [&](T const* thatBlock) {
            for (int i = 0; i < blockSize; ++i) {
              T scale = tmin(tabs(thisBlock[i]), tabs(thatBlock[i]));
              *(out++) = (thisBlock[i] - thatBlock[i]) / scale;
            }
          };
          if (x > 0) addResidual(p[(x - 1) + y * yStride + z * zStride]);
          if (y > 0) addResidual(p[x + (y - 1) * yStride + z * zStride]);
          if (z > 0) addResidual(p[x + y * yStride + (z - 1) * zStride]);
        }
      }
    }
  }
};

// =====================================================================================================
// Observation + cost functors (reference lib/PoseOptimizer.cpp:91-554)
// =====================================================================================================

// Observation, reference lib/PoseOptimizer.cpp:91-128. Block order: [pose(6), depth blocks, spatial blocks].
struct Obs {
  float ndc[2];
  float sourceDepth;
  int valueType = 0;
  int depthType = 0;
  Gather dg;  // depth functor blocks / weights
  Gather sg;  // spatial functor blocks / weights
  int numBlocks() const { return 1 + dg.n + sg.n; }
};

// GridDepthFunctor::eval / GlobalDepthFunctor / IdentityDepthFunctor,
// reference lib/DepthMapTransform.cpp:597-606, 504-510, 464-470.
template <typename T>
static T depthFunctor(const Obs& o, T const* const* dp) {
  if (o.depthType == CVD_DEPTH_IDENTITY) return T(static_cast<double>(o.sourceDepth));
  const T src(static_cast<double>(o.sourceDepth));
  if (o.depthType == CVD_DEPTH_GLOBAL) return valueXform(o.valueType, src, dp[0]);
  T res(0.0);
  for (int i = 0; i < o.dg.n; ++i) res += valueXform(o.valueType, src, dp[i]) * T(o.dg.w[i]);
  return res;
}

// GridSpatialFunctor::eval & friends, reference lib/DepthMapTransform.cpp:1225-1233, 1075-1085, 1146-1160.
template <typename T>
static void spatialFunctor(const Obs& o, T const* const* sp, T out[2]) {
  out[0] = T(0.0);
  out[1] = T(0.0);
  for (int i = 0; i < o.sg.n; ++i) {
    out[0] += sp[i][0] * T(o.sg.w[i]);
    out[1] += sp[i][1] * T(o.sg.w[i]);
  }
}

template <typename T>
struct ObsParams {
  T const* pose;
  T const* const* depthXform;
  T const* const* spatialXform;
};

// unpack, reference lib/PoseOptimizer.cpp:142-157
template <typename T>
static ObsParams<T> unpack(int& offset, T const* const* params, const Obs& obs) {
  ObsParams<T> p;
  p.pose = params[offset];
  offset += 1;
  p.depthXform = &params[offset];
  offset += obs.dg.n;
  p.spatialXform = &params[offset];
  offset += obs.sg.n;
  return p;
}

// obsToCamera, reference lib/PoseOptimizer.cpp:162-171
template <typename T>
static void obsToCamera(const Obs& obs, const ObsParams<T>& op, T out[3]) {
  T depth = depthFunctor(obs, op.depthXform);
  T warp[2];
  spatialFunctor(obs, op.spatialXform, warp);
  out[0] = T(static_cast<double>(obs.ndc[0])) + warp[0];
  out[1] = T(static_cast<double>(obs.ndc[1])) + warp[1];
  out[2] = depth;
}

// cameraToWorld, reference lib/PoseOptimizer.cpp:174-192
template <typename T>
static void cameraToWorld(const T pointCam[3], const T focal[2], T const* pose, T out[3]) {
  T dirCam[3] = {pointCam[0] * focal[0], pointCam[1] * focal[1], T(-1.0)};
  T dirWorld[3];
  angleAxisRotatePoint(pose + 3, dirCam, dirWorld);
  const T& depth = pointCam[2];
  out[0] = pose[0] + dirWorld[0] * depth;
  out[1] = pose[1] + dirWorld[1] * depth;
  out[2] = pose[2] + dirWorld[2] * depth;
}

// worldToCamera, reference lib/PoseOptimizer.cpp:195-221
template <typename T>
static void worldToCamera(const T pointWorld[3], const T focal[2], T const* pose, T out[3]) {
  T pointRel[3];
  for (int i = 0; i < 3; ++i) pointRel[i] = pointWorld[i] - pose[i];
  T rotInv[3];
  for (int i = 0; i < 3; ++i) rotInv[i] = -pose[i + 3];
  T pointCam[3];
  angleAxisRotatePoint(rotInv, pointRel, pointCam);
  const T depth = -pointCam[2];
  out[0] = pointCam[0] / depth / focal[0];
  out[1] = pointCam[1] / depth / focal[1];
  out[2] = depth;
}

struct CostFunction {
  int numResiduals = 0;
  std::vector<int> blockSizes;
  virtual ~CostFunction() = default;
  virtual void evalD(double const* const* p, double* r) const = 0;
  virtual void evalJ(Jet const* const* p, Jet* r) const = 0;
};

template <typename F>
struct AutoDiff : CostFunction {
  F f;
  explicit AutoDiff(F&& fn) : f(std::move(fn)) {}
  void evalD(double const* const* p, double* r) const override { f(p, r); }
  void evalJ(Jet const* const* p, Jet* r) const override { f(p, r); }
};

// StaticSceneCost, reference lib/PoseOptimizer.cpp:223-319
struct StaticSceneCost {
  Obs obs0, obs1;
  double fixedVFocal, aspect;
  int intrOpt, lossType;
  double spatialWeight, depthWeight;

  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    int off = 0;
    ObsParams<T> p0 = unpack(off, params, obs0);
    ObsParams<T> p1 = unpack(off, params, obs1);
    T focal0[2], focal1[2];
    if (intrOpt == CVD_INTR_SHARED) {
      focal0[1] = focal1[1] = params[off++][0];
    } else if (intrOpt == CVD_INTR_PER_FRAME) {
      focal0[1] = params[off++][0];
      focal1[1] = params[off++][0];
    } else {
      focal0[1] = focal1[1] = T(fixedVFocal);
    }
    focal0[0] = focal0[1] * aspect;
    focal1[0] = focal1[1] * aspect;

    T pointCam0[3], pointWorld0[3], pointCam1[3];
    obsToCamera(obs0, p0, pointCam0);
    cameraToWorld(pointCam0, focal0, p0.pose, pointWorld0);
    obsToCamera(obs1, p1, pointCam1);

    if (lossType == CVD_STATIC_EUCLIDEAN) {
      T pointWorld1[3];
      cameraToWorld(pointCam1, focal1, p1.pose, pointWorld1);
      for (int i = 0; i < 3; ++i) residuals[i] = pointWorld1[i] - pointWorld0[i];
    } else {
      T c01[3];
      worldToCamera(pointWorld0, focal1, p1.pose, c01);
      residuals[0] = (c01[0] - pointCam1[0]) * T(spatialWeight);
      residuals[1] = (c01[1] - pointCam1[1]) * T(spatialWeight);
      if (lossType == CVD_STATIC_REPRO_DISPARITY) {
        constexpr double epsilon = 1e-6;
        T reproDisp = 1.0 / tmax(c01[2], T(epsilon));
        T disp1 = 1.0 / tmax(pointCam1[2], T(epsilon));
        residuals[2] = (reproDisp - disp1) * T(depthWeight);
      } else {
        T maxDepth = tmax(c01[2], pointCam1[2]);
        T minDepth = tmin(c01[2], pointCam1[2]);
        if (lossType == CVD_STATIC_REPRO_DEPTH_RATIO) {
          residuals[2] = (maxDepth / minDepth - 1.0) * depthWeight;
        } else if (lossType == CVD_STATIC_REPRO_LOG_DEPTH) {
          residuals[2] = tlog(minDepth / maxDepth) * depthWeight;
        } else {
          throw std::runtime_error("Invalid loss type.");
        }
      }
    }
  }
};

// SceneFlowSmoothnessLoss, reference lib/PoseOptimizer.cpp:321-423
struct SceneFlowSmoothnessLoss {
  Obs obs0, obs1, obs2;
  double fixedVFocal, aspect;
  int intrOpt, lossType;

  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    int off = 0;
    ObsParams<T> p0 = unpack(off, params, obs0);
    T pc0[3];
    obsToCamera(obs0, p0, pc0);
    ObsParams<T> p1 = unpack(off, params, obs1);
    T pc1[3];
    obsToCamera(obs1, p1, pc1);
    ObsParams<T> p2 = unpack(off, params, obs2);
    T pc2[3];
    obsToCamera(obs2, p2, pc2);

    T f0[2], f1[2], f2[2];
    if (intrOpt == CVD_INTR_SHARED) {
      f0[1] = f1[1] = f2[1] = params[off++][0];
    } else if (intrOpt == CVD_INTR_PER_FRAME) {
      f0[1] = params[off++][0];
      f1[1] = params[off++][0];
      f2[1] = params[off++][0];
    } else {
      f0[1] = f1[1] = f2[1] = T(fixedVFocal);
    }
    f0[0] = f0[1] * aspect;
    f1[0] = f1[1] * aspect;
    f2[0] = f2[1] * aspect;

    if (lossType == CVD_SMOOTH_EUCLIDEAN_LAPLACIAN) {
      T w0[3], w1[3], w2[3];
      cameraToWorld(pc0, f0, p0.pose, w0);
      cameraToWorld(pc1, f1, p1.pose, w1);
      cameraToWorld(pc2, f2, p2.pose, w2);
      for (int i = 0; i < 3; ++i) residuals[i] = w0[i] + w2[i] - 2.0 * w1[i];
    } else {
      T w0[3], w2[3], c01[3], c21[3];
      cameraToWorld(pc0, f0, p0.pose, w0);
      cameraToWorld(pc2, f2, p2.pose, w2);
      worldToCamera(w0, f1, p1.pose, c01);
      worldToCamera(w2, f1, p1.pose, c21);
      residuals[0] = (c01[0] + c21[0] - pc1[0] * 2.0) / f1[1];
      residuals[1] = (c01[1] + c21[1] - pc1[1] * 2.0) / f1[1];
      if (lossType == CVD_SMOOTH_REPRO_DISPARITY_LAPLACIAN) {
        constexpr double epsilon = 1e-6;
        T d01 = 1.0 / tmax(c01[2], T(epsilon));
        T d1 = 1.0 / tmax(pc1[2], T(epsilon));
        T d21 = 1.0 / tmax(c21[2], T(epsilon));
        residuals[2] = d01 + d21 - d1 * 2.0;
      } else {
        T baseDepth = pc1[2];
        T otherDepth = c01[2] + c21[2] - pc1[2];
        T maxDepth = tmax(baseDepth, otherDepth);
        T minDepth = tmin(baseDepth, otherDepth);
        if (lossType == CVD_SMOOTH_REPRO_DEPTH_RATIO_CONSISTENCY) {
          residuals[2] = (maxDepth / minDepth - 1.0);
        } else if (lossType == CVD_SMOOTH_REPRO_LOG_DEPTH_CONSISTENCY) {
          residuals[2] = tlog(minDepth / maxDepth);
        } else {
          throw std::runtime_error("Invalid loss type.");
        }
      }
    }
  }
};

// DisparityDissimilarityCost, reference lib/PoseOptimizer.cpp:425-462 (obs carry only the depth functor)
struct DisparityDissimilarityCost {
  Obs obs0, obs1;
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    T const* const* x0 = &params[0];
    T const* const* x1 = &params[obs0.dg.n];
    T depth0 = depthFunctor(obs0, x0);
    T depth1 = depthFunctor(obs1, x1);
    T epsilon = T(1e-6);
    T disp0 = 1.0 / tmax(depth0, epsilon);
    T disp1 = 1.0 / tmax(depth1, epsilon);
    residuals[0] = disp0 - disp1;
  }
};

// ParameterRegularizationCost, reference lib/PoseOptimizer.cpp:464-483
struct ParameterRegularizationCost {
  int size;
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    for (int i = 0; i < size; ++i) residuals[i] = params[0][i] - T(2.0) * params[1][i] + params[2][i];
  }
};

// TargetDisparityCost, reference lib/PoseOptimizer.cpp:488-517
struct TargetDisparityCost {
  Obs obs;
  double targetDisparity;
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    T depth = depthFunctor(obs, params);
    T epsilon = T(1e-6);
    T disparity = 1.0 / tmax(depth, epsilon);
    residuals[0] = disparity - targetDisparity;
  }
};

// TargetFocalCost, reference lib/PoseOptimizer.cpp:520-533
struct TargetFocalCost {
  double targetFocal;
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    residuals[0] = params[0][0] - T(targetFocal);
  }
};

// DeformationCost, reference lib/PoseOptimizer.cpp:536-554
struct DeformationCost {
  const Xform* xform;
  double baseWeight;
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    xform->deformationCost(params, residuals);
    const int n = xform->numDeformationResiduals();
    for (int i = 0; i < n; ++i) residuals[i] *= baseWeight;
  }
};

// AdaptiveDeformationCost, reference lib/PoseOptimizer.cpp:559-656: vertex weights from the dynamic mask in the
// constructor (:580-618), residuals of computeDeformationCost modulated in its enumeration order (:622-645)
struct AdaptiveDeformationCost {
  const Xform* xform;
  double baseWeight, adaptiveWeight;
  int gw, gh, gz;
  std::vector<double> weights;  // gh x gw
  AdaptiveDeformationCost(const Xform* x, const uint8_t* dynamicMask, int dw, int dh, double base, double adaptive)
      : xform(x), baseWeight(base), adaptiveWeight(adaptive) {
    if (x->desc.type != CVD_XFORM_DEPTH || x->desc.depth_type != CVD_DEPTH_GRID)
      throw std::runtime_error("Adaptive deformation cost is only implemented for grid transforms.");
    gw = x->desc.grid_size[0];
    gh = x->desc.grid_size[1];
    gz = std::max(1, x->desc.grid_size[2]);
    std::vector<double> dynamicWeights(static_cast<size_t>(gw) * gh, 0.0), staticWeights(static_cast<size_t>(gw) * gh, 0.0);
    for (int y = 0; y < dh; ++y) {
      const double fy = double(y) * (gh - 1) / dh;
      const int iy = int(fy);
      const double ry = fy - iy;
      for (int xx = 0; xx < dw; ++xx) {
        const double fx = double(xx) * (gw - 1) / dw;
        const int ix = int(fx);
        const double rx = fx - ix;
        std::vector<double>& w = dynamicMask[static_cast<size_t>(y) * dw + xx] > 127 ? staticWeights : dynamicWeights;
        w[static_cast<size_t>(iy) * gw + ix] += (1.0 - rx) * (1.0 - ry);
        w[static_cast<size_t>(iy) * gw + ix + 1] += rx * (1.0 - ry);
        w[static_cast<size_t>(iy + 1) * gw + ix] += (1.0 - rx) * ry;
        w[static_cast<size_t>(iy + 1) * gw + ix + 1] += rx * ry;
      }
    }
    weights.resize(static_cast<size_t>(gw) * gh);
    for (size_t i = 0; i < weights.size(); ++i) weights[i] = dynamicWeights[i] / (dynamicWeights[i] + staticWeights[i]);
  }
  template <typename T>
  void operator()(T const* const* params, T* residuals) const {
    xform->deformationCost(params, residuals);
    // As written in the reference: ONE multiplication per edge, although computeGridDeformationCost emits blockSize
    // residuals per edge.  With Scale (one parameter, the default) every residual gets its own edge's weight; with
    // ScaleShift the first #edges residuals get the weights of edges 0 .. #edges-1 in enumeration order and the rest
    // stay unscaled.  Restated literally.
    int idx = 0;
    for (int z = 0; z < gz; ++z)
      for (int y = 0; y < gh; ++y)
        for (int x = 0; x < gw; ++x) {
          const double w0 = weights[static_cast<size_t>(y) * gw + x];
          if (x > 0) {
            const double w1 = weights[static_cast<size_t>(y) * gw + x - 1];
            residuals[idx++] *= baseWeight + std::max(w0, w1) * adaptiveWeight;
          }
          if (y > 0) {
            const double w1 = weights[static_cast<size_t>(y - 1) * gw + x];
            residuals[idx++] *= baseWeight + std::max(w0, w1) * adaptiveWeight;
          }
          if (z > 0) residuals[idx++] *= baseWeight + w0 * adaptiveWeight;
        }
  }
};

// =====================================================================================================
// ceres::Problem / evaluator restatement
// =====================================================================================================

enum LossKind { LOSS_NONE = 0, LOSS_CAUCHY = 1, LOSS_SCALED = 2, LOSS_HUBER = 3 };

struct ParamBlock {
  double* ptr = nullptr;
  int size = 0;
  int frame = 0;   // owner frame (for deterministic parallel accumulation + canonical layout)
  int canon = 0;   // offset in the canonical [F x B] layout
  bool constant = false;
  bool hasLower0 = false;
  double lower0 = 0.0;
  int offset = -1;  // offset in the reduced (active) parameter vector
};

struct ResidualBlock {
  std::unique_ptr<CostFunction> cost;
  int lossKind = LOSS_NONE;
  double lossParam = 0.0;
  std::vector<int> blocks;
};

struct Problem {
  std::vector<ParamBlock> blocks;
  std::map<const double*, int> lookup;
  std::vector<ResidualBlock> residuals;
  int numActive = 0;
  // reduced (active) unknowns grouped by frame: frame f owns [frameOff[f], frameOff[f + 1]); hPairs = frame pairs that
  // share a residual block = the block structure of J^T J (block_sparse.h)
  std::vector<int> frameOff;
  std::vector<std::pair<int, int>> hPairs;

  int blockId(double* ptr, int size, int frame, int canon) {
    auto it = lookup.find(ptr);
    if (it != lookup.end()) return it->second;
    ParamBlock b;
    b.ptr = ptr;
    b.size = size;
    b.frame = frame;
    b.canon = canon;
    blocks.push_back(b);
    lookup[ptr] = static_cast<int>(blocks.size()) - 1;
    return static_cast<int>(blocks.size()) - 1;
  }
  bool has(const double* ptr) const { return lookup.count(ptr) != 0; }
  void setConstant(const double* ptr) {
    auto it = lookup.find(ptr);
    if (it != lookup.end()) blocks[it->second].constant = true;
  }
  void setLowerBound0(const double* ptr, double lb) {
    auto it = lookup.find(ptr);
    if (it != lookup.end()) {
      blocks[it->second].hasLower0 = true;
      blocks[it->second].lower0 = lb;
    }
  }
  // Reduced ordering: active parameter blocks sorted by (frame, canonical offset), so that every frame's unknowns
  // are contiguous (Ceres orders its parameter blocks itself; the order does not change the solution).
  void finalize() {
    std::vector<int> order(blocks.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<int>(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (blocks[a].frame != blocks[b].frame) return blocks[a].frame < blocks[b].frame;
      return blocks[a].canon < blocks[b].canon;
    });
    int nF = 0;
    for (const auto& b : blocks) nF = std::max(nF, b.frame + 1);
    frameOff.assign(nF + 1, 0);
    int off = 0;
    for (int id : order) {
      ParamBlock& b = blocks[id];
      if (b.constant) {
        b.offset = -1;
      } else {
        b.offset = off;
        off += b.size;
        frameOff[b.frame + 1] += b.size;
      }
    }
    for (int f = 0; f < nF; ++f) frameOff[f + 1] += frameOff[f];
    numActive = off;
    std::set<std::pair<int, int>> pairs;
    for (const auto& rb : residuals) {
      int fr[8];
      int nfr = 0;
      for (int id : rb.blocks) {
        if (blocks[id].offset < 0) continue;
        const int f = blocks[id].frame;
        bool seen = false;
        for (int k = 0; k < nfr; ++k) seen |= (fr[k] == f);
        if (!seen && nfr < 8) fr[nfr++] = f;
      }
      for (int a = 0; a < nfr; ++a)
        for (int b = a + 1; b < nfr; ++b) pairs.insert({std::max(fr[a], fr[b]), std::min(fr[a], fr[b])});
    }
    hPairs.assign(pairs.begin(), pairs.end());
  }
};

// ceres::CauchyLoss / HuberLoss / ScaledLoss(nullptr, a) -> rho[0..2] (ceres/loss_function.cc).
static void evalLoss(int kind, double p, double s, double rho[3]) {
  if (kind == LOSS_CAUCHY) {
    const double b = p * p;
    const double c = 1.0 / b;
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c * (inv * inv);
  } else if (kind == LOSS_HUBER) {
    // ceres::HuberLoss(a) (ceres/loss_function.cc): outlier region s > a^2
    const double b = p * p;
    if (s > b) {
      const double r = std::sqrt(s);
      rho[0] = 2.0 * p * r - b;
      rho[1] = std::max(std::numeric_limits<double>::min(), p / r);
      rho[2] = -rho[1] / (2.0 * s);
    } else {
      rho[0] = s;
      rho[1] = 1.0;
      rho[2] = 0.0;
    }
  } else if (kind == LOSS_SCALED) {
    rho[0] = p * s;
    rho[1] = p;
    rho[2] = 0.0;
  } else {
    rho[0] = s;
    rho[1] = 1.0;
    rho[2] = 0.0;
  }
}

// Values (+ Jacobian, by dual numbers in passes of kStride exactly like ceres::DynamicAutoDiffCostFunction::Evaluate) of
// one cost function at the parameter blocks pd[b].  `jac` is laid out [numResiduals x totalParams] row-major over the
// concatenated parameter blocks.  No loss function here.
static void evaluateCostFunctionAt(const CostFunction& cf, const double* const* pd, double* res, double* jac) {
  const int nb = static_cast<int>(cf.blockSizes.size());
  const int nr = cf.numResiduals;
  int total = 0;
  for (int b = 0; b < nb; ++b) total += cf.blockSizes[b];
  if (!jac) {
    cf.evalD(pd, res);
    return;
  }
  std::vector<Jet> store(total);
  std::vector<const Jet*> pj(nb);
  {
    int k = 0;
    for (int b = 0; b < nb; ++b) {
      pj[b] = &store[k];
      for (int i = 0; i < cf.blockSizes[b]; ++i) store[k++] = Jet(pd[b][i]);
    }
  }
  std::vector<Jet> out(nr);
  for (int start = 0; start < total; start += kStride) {
    const int end = std::min(total, start + kStride);
    for (int k = start; k < end; ++k) store[k].v[k - start] = 1.0;
    cf.evalJ(pj.data(), out.data());
    for (int k = start; k < end; ++k) {
      for (int r = 0; r < nr; ++r) jac[static_cast<size_t>(r) * total + k] = out[r].v[k - start];
      store[k].v[k - start] = 0.0;
    }
    if (start == 0)
      for (int r = 0; r < nr; ++r) res[r] = out[r].a;
  }
  if (total == 0) cf.evalD(pd, res);
}

// One residual block at the current state of its parameter blocks, then the loss corrector (ceres/corrector.cc).
static double evaluateResidualBlock(const Problem& pb, const ResidualBlock& rb, double* res, double* jac) {
  const CostFunction& cf = *rb.cost;
  const int nb = static_cast<int>(rb.blocks.size());
  const int nr = cf.numResiduals;
  int total = 0;
  for (int b = 0; b < nb; ++b) total += cf.blockSizes[b];

  std::vector<const double*> pd(nb);
  for (int b = 0; b < nb; ++b) pd[b] = pb.blocks[rb.blocks[b]].ptr;
  evaluateCostFunctionAt(cf, pd.data(), res, jac);

  double sq = 0.0;
  for (int r = 0; r < nr; ++r) sq += res[r] * res[r];
  if (rb.lossKind == LOSS_NONE) return 0.5 * sq;

  double rho[3];
  evalLoss(rb.lossKind, rb.lossParam, sq, rho);
  // Corrector: rho'' <= 0 for Cauchy and Scaled => plain sqrt(rho') scaling of residuals and Jacobian.
  const double sqrtRho1 = std::sqrt(rho[1]);
  double residualScaling, alphaSqNorm;
  if (sq == 0.0 || rho[2] <= 0.0) {
    residualScaling = sqrtRho1;
    alphaSqNorm = 0.0;
  } else {
    const double D = 1.0 + 2.0 * sq * rho[2] / rho[1];
    const double alpha = 1.0 - std::sqrt(D);
    residualScaling = sqrtRho1 / (1.0 - alpha);
    alphaSqNorm = alpha / sq;
  }
  if (jac) {
    if (alphaSqNorm == 0.0) {
      for (size_t i = 0; i < static_cast<size_t>(nr) * total; ++i) jac[i] *= sqrtRho1;
    } else {
      // J = sqrt(rho') * (J - alpha/|r|^2 r r^T J)
      for (int c = 0; c < total; ++c) {
        double rtj = 0.0;
        for (int r = 0; r < nr; ++r) rtj += res[r] * jac[static_cast<size_t>(r) * total + c];
        for (int r = 0; r < nr; ++r)
          jac[static_cast<size_t>(r) * total + c] =
              sqrtRho1 * (jac[static_cast<size_t>(r) * total + c] - alphaSqNorm * res[r] * rtj);
      }
    }
  }
  for (int r = 0; r < nr; ++r) res[r] *= residualScaling;
  return 0.5 * rho[0];
}

struct Evaluation {
  double cost = 0.0;
  std::vector<double> g;  // reduced gradient J^T r
  BlockSym H;             // reduced J^T J, one dense block per coupled frame pair (block_sparse.h)
};

// Evaluator::Evaluate: cost (+ gradient + J^T J). Deterministic for any thread count: Jacobians are
// computed in parallel, accumulation rows are owned by (frame % threads).  J^T J is accumulated straight into its
// frame-pair blocks (lower block triangle; the thread that owns the larger frame of a pair writes the block).
static void evaluateProblem(const Problem& pb, int numThreads, bool wantDerivs, Evaluation& ev) {
  const int n = pb.numActive;
  const size_t R = pb.residuals.size();
  ev.cost = 0.0;
  if (wantDerivs) {
    ev.g.assign(n, 0.0);
    const int nF = static_cast<int>(pb.frameOff.size()) - 1;
    if (ev.H.nb != nF || ev.H.n() != n) {
      std::vector<int> sizes(std::max(nF, 0));
      for (int f = 0; f < nF; ++f) sizes[f] = pb.frameOff[f + 1] - pb.frameOff[f];
      ev.H.build(sizes, pb.hPairs);
    } else {
      ev.H.zero();
    }
  }
  int T = std::max(1, numThreads);
#ifdef _OPENMP
  T = std::min(T, omp_get_max_threads());
#else
  T = 1;
#endif
  const size_t kChunk = 8192;
  std::vector<double> costs(kChunk);
  std::vector<std::vector<double>> resBuf(kChunk), jacBuf(kChunk);
  for (size_t c0 = 0; c0 < R; c0 += kChunk) {
    const size_t c1 = std::min(R, c0 + kChunk);
#pragma omp parallel for schedule(dynamic, 64) num_threads(T)
    for (long long i = static_cast<long long>(c0); i < static_cast<long long>(c1); ++i) {
      const ResidualBlock& rb = pb.residuals[i];
      const int nr = rb.cost->numResiduals;
      int total = 0;
      for (int s : rb.cost->blockSizes) total += s;
      auto& rbuf = resBuf[i - c0];
      auto& jbuf = jacBuf[i - c0];
      rbuf.resize(nr);
      if (wantDerivs) jbuf.resize(static_cast<size_t>(nr) * total);
      costs[i - c0] = evaluateResidualBlock(pb, rb, rbuf.data(), wantDerivs ? jbuf.data() : nullptr);
    }
    for (size_t i = c0; i < c1; ++i) ev.cost += costs[i - c0];
    if (!wantDerivs) continue;
#pragma omp parallel num_threads(T)
    {
#ifdef _OPENMP
      const int tid = omp_get_thread_num();
      const int nt = omp_get_num_threads();
#else
      const int tid = 0, nt = 1;
#endif
      std::vector<int> colOf, frameOfBlock;  // per parameter block of the residual: column in J, frame (-1 constant)
      for (size_t i = c0; i < c1; ++i) {
        const ResidualBlock& rb = pb.residuals[i];
        const int nr = rb.cost->numResiduals;
        const int nb = static_cast<int>(rb.blocks.size());
        int total = 0;
        colOf.resize(nb);
        frameOfBlock.resize(nb);
        bool mine = false;
        for (int b = 0; b < nb; ++b) {
          colOf[b] = total;
          total += rb.cost->blockSizes[b];
          const ParamBlock& P = pb.blocks[rb.blocks[b]];
          frameOfBlock[b] = P.offset >= 0 ? P.frame : -1;
          mine |= (P.offset >= 0 && (P.frame % nt) == tid);
        }
        if (!mine) continue;
        const double* J = jacBuf[i - c0].data();
        const double* r = resBuf[i - c0].data();
        // frame-pair blocks of this residual (a handful of distinct frames): looked up once per pair
        int fr[8], nfr = 0;
        for (int b = 0; b < nb; ++b) {
          const int f = frameOfBlock[b];
          if (f < 0) continue;
          bool seen = false;
          for (int k = 0; k < nfr; ++k) seen |= (fr[k] == f);
          if (!seen && nfr < 8) fr[nfr++] = f;
        }
        double* blkPtr[8][8];
        int blkLd[8][8];
        for (int a = 0; a < nfr; ++a)
          for (int b = 0; b < nfr; ++b) {
            blkPtr[a][b] = nullptr;
            blkLd[a][b] = 0;
            if (fr[a] >= fr[b] && (fr[a] % nt) == tid) {
              const int e = ev.H.find(fr[a], fr[b]);
              blkPtr[a][b] = ev.H.val.data() + ev.H.blkOff[e];
              blkLd[a][b] = ev.H.size(fr[b]);
            }
          }
        auto slot = [&](int f) { for (int k = 0; k < nfr; ++k) if (fr[k] == f) return k; return 0; };
        for (int bi = 0; bi < nb; ++bi) {
          const int fi = frameOfBlock[bi];
          if (fi < 0 || (fi % nt) != tid) continue;
          const ParamBlock& Bi = pb.blocks[rb.blocks[bi]];
          const int si = rb.cost->blockSizes[bi];
          const int ci = colOf[bi];
          const int li = Bi.offset - pb.frameOff[fi];
          const int sa = slot(fi);
          for (int a = 0; a < si; ++a) {
            double gs = 0.0;
            for (int k = 0; k < nr; ++k) gs += J[static_cast<size_t>(k) * total + ci + a] * r[k];
            ev.g[Bi.offset + a] += gs;
          }
          for (int bj = 0; bj < nb; ++bj) {
            const int fj = frameOfBlock[bj];
            if (fj < 0 || fj > fi) continue;  // lower block triangle (the diagonal blocks get both triangles)
            const ParamBlock& Bj = pb.blocks[rb.blocks[bj]];
            const int sj = rb.cost->blockSizes[bj];
            const int cj = colOf[bj];
            const int lj = Bj.offset - pb.frameOff[fj];
            const int sb = slot(fj);
            double* Hb = blkPtr[sa][sb];
            const int ld = blkLd[sa][sb];
            for (int a = 0; a < si; ++a)
              for (int b = 0; b < sj; ++b) {
                double s2 = 0.0;
                for (int k = 0; k < nr; ++k)
                  s2 += J[static_cast<size_t>(k) * total + ci + a] * J[static_cast<size_t>(k) * total + cj + b];
                Hb[static_cast<size_t>(li + a) * ld + lj + b] += s2;
              }
          }
        }
      }
    }
  }
}

// Dense in-place lower Cholesky A = L L^T (row-major, n x n); returns false if not positive definite.
static bool choleskyFactor(std::vector<double>& A, int n, int numThreads) {
  const int NB = 64;
  (void)numThreads;
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int k1 = std::min(n, k0 + NB);
    // factor diagonal block
    for (int j = k0; j < k1; ++j) {
      double d = A[static_cast<size_t>(j) * n + j];
      for (int p = k0; p < j; ++p) d -= A[static_cast<size_t>(j) * n + p] * A[static_cast<size_t>(j) * n + p];
      if (!(d > 0.0) || !std::isfinite(d)) return false;
      d = std::sqrt(d);
      A[static_cast<size_t>(j) * n + j] = d;
      for (int i = j + 1; i < k1; ++i) {
        double s = A[static_cast<size_t>(i) * n + j];
        for (int p = k0; p < j; ++p) s -= A[static_cast<size_t>(i) * n + p] * A[static_cast<size_t>(j) * n + p];
        A[static_cast<size_t>(i) * n + j] = s / d;
      }
    }
    // panel solve: rows below
#pragma omp parallel for schedule(static) num_threads(std::max(1, numThreads)) if (n - k1 > 256)
    for (int i = k1; i < n; ++i) {
      for (int j = k0; j < k1; ++j) {
        double s = A[static_cast<size_t>(i) * n + j];
        for (int p = k0; p < j; ++p) s -= A[static_cast<size_t>(i) * n + p] * A[static_cast<size_t>(j) * n + p];
        A[static_cast<size_t>(i) * n + j] = s / A[static_cast<size_t>(j) * n + j];
      }
    }
    // trailing update (lower part only)
#pragma omp parallel for schedule(dynamic, 8) num_threads(std::max(1, numThreads)) if (n - k1 > 256)
    for (int i = k1; i < n; ++i) {
      const double* Li = &A[static_cast<size_t>(i) * n + k0];
      for (int j = k1; j <= i; ++j) {
        const double* Lj = &A[static_cast<size_t>(j) * n + k0];
        double s = 0.0;
        for (int p = 0; p < k1 - k0; ++p) s += Li[p] * Lj[p];
        A[static_cast<size_t>(i) * n + j] -= s;
      }
    }
  }
  return true;
}

static void choleskySolve(const std::vector<double>& L, int n, std::vector<double>& b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    const double* Li = &L[static_cast<size_t>(i) * n];
    for (int p = 0; p < i; ++p) s -= Li[p] * b[p];
    b[i] = s / Li[i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int p = i + 1; p < n; ++p) s -= L[static_cast<size_t>(p) * n + i] * b[p];
    b[i] = s / L[static_cast<size_t>(i) * n + i];
  }
}

// Ceres Solver::Options defaults that matter on this path (the reference only sets linear solver type,
// max_num_iterations and num_threads: lib/PoseOptimizer.cpp:955-961).
struct CeresDefaults {
  static constexpr double initial_trust_region_radius = 1e4;
  static constexpr double max_trust_region_radius = 1e16;
  static constexpr double min_trust_region_radius = 1e-32;
  static constexpr double min_relative_decrease = 1e-3;
  static constexpr double min_lm_diagonal = 1e-6;
  static constexpr double max_lm_diagonal = 1e32;
  static constexpr double function_tolerance = 1e-6;
  static constexpr double gradient_tolerance = 1e-10;
  static constexpr double parameter_tolerance = 1e-8;
  static constexpr int max_num_consecutive_invalid_steps = 5;
};

static void gatherState(const Problem& pb, std::vector<double>& x) {
  x.resize(pb.numActive);
  for (const auto& b : pb.blocks)
    if (b.offset >= 0)
      for (int i = 0; i < b.size; ++i) x[b.offset + i] = b.ptr[i];
}
static void scatterState(const Problem& pb, const std::vector<double>& x) {
  for (const auto& b : pb.blocks)
    if (b.offset >= 0)
      for (int i = 0; i < b.size; ++i) b.ptr[i] = x[b.offset + i];
}
// ParameterBlock::Plus with box projection.
static void plusProject(const Problem& pb, const std::vector<double>& x, const std::vector<double>& d,
                        std::vector<double>& out) {
  out.resize(x.size());
  for (size_t i = 0; i < x.size(); ++i) out[i] = x[i] + d[i];
  for (const auto& b : pb.blocks)
    if (b.offset >= 0 && b.hasLower0) out[b.offset] = std::max(out[b.offset], b.lower0);
}

// ceres::Solve with TRUST_REGION / LEVENBERG_MARQUARDT / exact normal-equation Cholesky, jacobi_scaling,
// monotonic steps (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc restated).
// Not restated: the projected line search Ceres runs for bound-constrained problems (only normalizeDepth
// has bounds; they stay inactive at its solution).
// linearSolver 0: exact block-sparse Cholesky on the frame graph (block_sparse.h), the stand-in for the reference's
// SPARSE_NORMAL_CHOLESKY; 1: the same system assembled densely and factorised by choleskyFactor (cross-check of the
// sparse code, small problems only).  functionTolerance: Ceres default 1e-6 unless a test asks for a tighter reference
// solution.
static void solveProblem(Problem& pb, int maxIterations, int numThreads, cvd_solve_summary* summary,
                         std::vector<cvd_iteration_record>* records, int linearSolver = 0,
                         double functionTolerance = -1.0) {
  using D = CeresDefaults;
  const double t0 = nowSeconds();
  double tEval = 0.0, tLin = 0.0;
  pb.finalize();
  const int n = pb.numActive;
  cvd_solve_summary sum{};
  sum.num_residual_blocks = static_cast<int>(pb.residuals.size());
  sum.num_parameters = n;

  std::vector<double> x, cand, delta(n), step(n), scale(n, 1.0);
  gatherState(pb, x);
  bool constrained = false;
  for (const auto& b : pb.blocks) constrained |= (b.offset >= 0 && b.hasLower0);
  if (constrained) {
    std::vector<double> zero(n, 0.0);
    plusProject(pb, x, zero, cand);
    x = cand;
    scatterState(pb, x);
  }

  Evaluation ev;
  double te = nowSeconds();
  evaluateProblem(pb, numThreads, true, ev);
  tEval += nowSeconds() - te;
  double xCost = ev.cost;
  sum.initial_cost = xCost;

  auto gradMaxNorm = [&](const std::vector<double>& g) {
    double m = 0.0;
    if (!constrained) {
      for (double v : g) m = std::max(m, std::abs(v));
    } else {
      std::vector<double> neg(n), proj;
      for (int i = 0; i < n; ++i) neg[i] = -g[i];
      plusProject(pb, x, neg, proj);
      for (int i = 0; i < n; ++i) m = std::max(m, std::abs(x[i] - proj[i]));
    }
    return m;
  };

  const double fTol = functionTolerance > 0.0 ? functionTolerance : D::function_tolerance;
  // Jacobi scaling from the first Jacobian: 1 / (1 + sqrt(squared column norm)).
  {
    std::vector<double> hd(n);
    ev.H.diagonal(hd.data());
    for (int i = 0; i < n; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(hd[i]));
  }
  BlockCholesky chol;
  if (linearSolver == 0 && n > 0) {
    const double ta = nowSeconds();
    const double fl = chol.analyze(ev.H);
    if (std::getenv("CVDO_VERBOSE"))
      std::fprintf(stderr, "[oracle] finalize+first evaluation %.2f s, block Cholesky analysis %.2f s: %zu blocks, %.2f GB, %.3f Tflop per factorisation\n",
                   ta - t0, nowSeconds() - ta, chol.numBlocks(), chol.numValues() * 8e-9, fl * 1e-12);
  }

  double radius = D::initial_trust_region_radius;
  double decreaseFactor = 2.0;
  int invalidSteps = 0;
  int iteration = 0;
  int termination = 1;
  double xNorm = 0.0;
  for (double v : x) xNorm += v * v;
  xNorm = std::sqrt(xNorm);

  cvd_iteration_record rec0{};
  rec0.iteration = 0;
  rec0.cost = xCost;
  rec0.gradient_max_norm = gradMaxNorm(ev.g);
  rec0.trust_region_radius = radius;
  rec0.step_is_successful = 1;
  if (records) records->push_back(rec0);

  if (n == 0 || rec0.gradient_max_norm <= D::gradient_tolerance) {
    termination = 0;
  } else {
    std::vector<double> A, y(n), gs(n), diag(n);
    while (true) {
      if (iteration >= maxIterations) { termination = 1; break; }
      if (radius < D::min_trust_region_radius) { termination = 0; break; }
      ++iteration;
      cvd_iteration_record rec{};
      rec.iteration = iteration;

      // ---- LevenbergMarquardtStrategy::ComputeStep on the column-scaled system
      //      (S H S + diag(clamp(diag(S H S))) / radius) y = S g,  step = -y
      double tl = nowSeconds();
      ev.H.diagonal(diag.data());
      for (int i = 0; i < n; ++i) {
        const double dii = diag[i] * scale[i] * scale[i];
        diag[i] = std::min(std::max(dii, D::min_lm_diagonal), D::max_lm_diagonal) / radius;
        gs[i] = ev.g[i] * scale[i];
      }
      bool ok;
      y = gs;
      if (linearSolver == 0) {
        ok = chol.factor(ev.H, scale.data(), diag.data(), numThreads);
        if (ok) chol.solve(y.data());
      } else {
        A.assign(static_cast<size_t>(n) * n, 0.0);
        for (int I = 0; I < ev.H.nb; ++I)
          for (int e = ev.H.rowPtr[I]; e < ev.H.rowPtr[I + 1]; ++e) {
            const int J = ev.H.rowCol[e];
            const int ni = ev.H.size(I), nj = ev.H.size(J);
            const double* Bv = ev.H.val.data() + ev.H.blkOff[e];
            for (int a = 0; a < ni; ++a)
              for (int b = 0; b < nj; ++b) {
                const int row = ev.H.off[I] + a, col = ev.H.off[J] + b;
                A[static_cast<size_t>(row) * n + col] = Bv[static_cast<size_t>(a) * nj + b] * scale[row] * scale[col];
              }
          }
        for (int i = 0; i < n; ++i) A[static_cast<size_t>(i) * n + i] += diag[i];
        ok = choleskyFactor(A, n, numThreads);
        if (ok) choleskySolve(A, n, y);
      }
      if (ok)
        for (int i = 0; i < n; ++i) {
          step[i] = -y[i];
          if (!std::isfinite(step[i])) ok = false;
        }
      tLin += nowSeconds() - tl;

      double modelCostChange = 0.0;
      if (ok) {
        // model_cost_change = -(J step)^T (r + J step / 2) = -(step^T gs + step^T Hs step / 2)
        for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];
        ev.H.multiply(delta.data(), y.data());
        double sg = 0.0, sHs = 0.0;
        for (int i = 0; i < n; ++i) {
          sg += step[i] * gs[i];
          sHs += delta[i] * y[i];
        }
        modelCostChange = -(sg + 0.5 * sHs);
        ok = modelCostChange > 0.0;
      }
      if (!ok) {
        // HandleInvalidStep + LevenbergMarquardtStrategy::StepIsInvalid
        if (++invalidSteps >= D::max_num_consecutive_invalid_steps) { termination = 2; break; }
        radius = radius / decreaseFactor;
        decreaseFactor *= 2.0;
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        if (records) records->push_back(rec);
        continue;
      }
      invalidSteps = 0;
      for (int i = 0; i < n; ++i) delta[i] = step[i] * scale[i];

      // ---- candidate point + cost
      plusProject(pb, x, delta, cand);
      scatterState(pb, cand);
      Evaluation evc;
      te = nowSeconds();
      evaluateProblem(pb, numThreads, false, evc);
      tEval += nowSeconds() - te;
      double candCost = evc.cost;
      if (!std::isfinite(candCost)) candCost = std::numeric_limits<double>::max();

      double stepNorm = 0.0;
      for (int i = 0; i < n; ++i) stepNorm += delta[i] * delta[i];
      stepNorm = std::sqrt(stepNorm);
      rec.step_norm = stepNorm;
      rec.cost_change = xCost - candCost;
      rec.relative_decrease = (xCost - candCost) / modelCostChange;

      // ParameterToleranceReached
      if (stepNorm <= D::parameter_tolerance * (xNorm + D::parameter_tolerance)) {
        scatterState(pb, x);
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        if (records) records->push_back(rec);
        termination = 0;
        break;
      }
      // FunctionToleranceReached
      if (std::abs(xCost - candCost) <= fTol * xCost) {
        // Ceres keeps the iterate it had unless the step is an improvement recorded earlier; the
        // candidate is NOT accepted here (minimizer returns before IsStepSuccessful).
        scatterState(pb, x);
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        if (records) records->push_back(rec);
        termination = 0;
        break;
      }

      if (rec.relative_decrease > D::min_relative_decrease) {
        // HandleSuccessfulStep
        x = cand;
        xNorm = 0.0;
        for (double v : x) xNorm += v * v;
        xNorm = std::sqrt(xNorm);
        xCost = candCost;
        te = nowSeconds();
        evaluateProblem(pb, numThreads, true, ev);
        tEval += nowSeconds() - te;
        ++sum.num_successful_steps;
        rec.step_is_successful = 1;
        // LevenbergMarquardtStrategy::StepAccepted
        const double q = rec.relative_decrease;
        radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * q - 1.0, 3));
        radius = std::min(D::max_trust_region_radius, radius);
        decreaseFactor = 2.0;
        rec.cost = xCost;
        rec.gradient_max_norm = gradMaxNorm(ev.g);
        rec.trust_region_radius = radius;
        if (records) records->push_back(rec);
        if (rec.gradient_max_norm <= D::gradient_tolerance) { termination = 0; break; }
      } else {
        // StepRejected
        scatterState(pb, x);
        radius = radius / decreaseFactor;
        decreaseFactor *= 2.0;
        rec.cost = xCost;
        rec.trust_region_radius = radius;
        if (records) records->push_back(rec);
      }
    }
  }
  scatterState(pb, x);
  sum.num_iterations = iteration;
  sum.termination = termination;
  sum.final_cost = xCost;
  sum.total_seconds = nowSeconds() - t0;
  sum.evaluate_seconds = tEval;
  sum.linear_solve_seconds = tLin;
  if (summary) *summary = sum;
}

#ifdef CVDO_WITH_CERES
}  // namespace cvdo
#include <ceres/ceres.h>
namespace cvdo {
// ---- the "true Ceres" column: the SAME residual blocks handed to a real ceres::Problem -------------------
// Built only with `make -C oracle CERES=1` on a machine that has Ceres (none here: SURVEY.md 8c).  Every oracle residual
// block becomes a ceres::CostFunction whose values / Jacobians come from the oracle's dual-number evaluation; the loss
// functions (ceres::CauchyLoss / HuberLoss / ScaledLoss), the corrector, the trust-region loop, its termination tests and
// SPARSE_NORMAL_CHOLESKY are Ceres' own -- exactly the parts the oracle restates from Ceres' published algorithm, so a
// matching end state pins them (tests/test_oracle_ceres.py).
class OracleCostFunction : public ceres::CostFunction {
 public:
  explicit OracleCostFunction(const CostFunction* cf) : cf_(cf) {
    set_num_residuals(cf->numResiduals);
    for (int sz : cf->blockSizes) mutable_parameter_block_sizes()->push_back(sz);
  }
  bool Evaluate(double const* const* parameters, double* residuals, double** jacobians) const override {
    const int nb = static_cast<int>(cf_->blockSizes.size());
    const int nr = cf_->numResiduals;
    int total = 0;
    for (int sz : cf_->blockSizes) total += sz;
    if (!jacobians) {
      evaluateCostFunctionAt(*cf_, parameters, residuals, nullptr);
      return true;
    }
    std::vector<double> jac(static_cast<size_t>(nr) * total);
    evaluateCostFunctionAt(*cf_, parameters, residuals, jac.data());
    int col = 0;
    for (int b = 0; b < nb; ++b) {
      const int sz = cf_->blockSizes[b];
      if (jacobians[b])
        for (int r = 0; r < nr; ++r)
          for (int k = 0; k < sz; ++k) jacobians[b][r * sz + k] = jac[static_cast<size_t>(r) * total + col + k];
      col += sz;
    }
    return true;
  }

 private:
  const CostFunction* cf_;
};

static void solveProblemCeres(Problem& pb, int maxIterations, int numThreads, cvd_solve_summary* summary,
                              std::vector<cvd_iteration_record>* records) {
  ceres::Problem problem;
  for (const auto& rb : pb.residuals) {
    ceres::LossFunction* loss = nullptr;
    if (rb.lossKind == LOSS_CAUCHY) loss = new ceres::CauchyLoss(rb.lossParam);
    else if (rb.lossKind == LOSS_HUBER) loss = new ceres::HuberLoss(rb.lossParam);
    else if (rb.lossKind == LOSS_SCALED) loss = new ceres::ScaledLoss(nullptr, rb.lossParam, ceres::TAKE_OWNERSHIP);
    std::vector<double*> ptrs;
    for (int id : rb.blocks) ptrs.push_back(pb.blocks[id].ptr);
    problem.AddResidualBlock(new OracleCostFunction(rb.cost.get()), loss, ptrs);
  }
  for (const auto& b : pb.blocks) {
    if (!problem.HasParameterBlock(b.ptr)) continue;
    if (b.constant) problem.SetParameterBlockConstant(b.ptr);
    if (b.hasLower0) problem.SetParameterLowerBound(b.ptr, 0, b.lower0);
  }
  ceres::Solver::Options options;  // what the reference sets (lib/PoseOptimizer.cpp:955-961), nothing else
  options.linear_solver_type = ceres::SPARSE_NORMAL_CHOLESKY;
  options.minimizer_progress_to_stdout = false;
  options.max_num_iterations = maxIterations;
  options.num_threads = numThreads;
  ceres::Solver::Summary cs;
  ceres::Solve(options, &problem, &cs);
  cvd_solve_summary sum{};
  sum.num_residual_blocks = static_cast<int>(pb.residuals.size());
  sum.num_parameters = cs.num_effective_parameters_reduced;
  sum.num_iterations = static_cast<int>(cs.iterations.size()) - 1;
  sum.num_successful_steps = cs.num_successful_steps;
  sum.termination = cs.termination_type == ceres::CONVERGENCE ? 0 : (cs.termination_type == ceres::NO_CONVERGENCE ? 1 : 2);
  sum.initial_cost = cs.initial_cost;
  sum.final_cost = cs.final_cost;
  sum.total_seconds = cs.total_time_in_seconds;
  sum.evaluate_seconds = cs.residual_evaluation_time_in_seconds + cs.jacobian_evaluation_time_in_seconds;
  sum.linear_solve_seconds = cs.linear_solver_time_in_seconds;
  if (summary) *summary = sum;
  if (records)
    for (const auto& it : cs.iterations) {
      cvd_iteration_record r{};
      r.iteration = it.iteration;
      r.step_is_successful = it.step_is_successful ? 1 : 0;
      r.cost = it.cost;
      r.cost_change = it.cost_change;
      r.gradient_max_norm = it.gradient_max_norm;
      r.step_norm = it.step_norm;
      r.relative_decrease = it.relative_decrease;
      r.trust_region_radius = it.trust_region_radius;
      records->push_back(r);
    }
}
#endif  // CVDO_WITH_CERES

// =====================================================================================================
// The optimizer (reference lib/PoseOptimizer.cpp:748-1549, lib/Processor.cpp:888-1013)
// =====================================================================================================

struct Oracle {
  int robustLoss = 0;  // 0 CauchyLoss (reference), 1 HuberLoss (BASELINE configs[4] stress variant): cvdo_set_robust_loss
  int linearSolver = 0;            // 0 block-sparse Cholesky (default), 1 dense Cholesky (cross-check): cvdo_set_linear_solver
  double functionTolerance = -1.0; // <= 0: Ceres' default 1e-6 (cvdo_set_function_tolerance: tighter reference solutions)
  int F = 0, W = 0, Hh = 0;
  float aspect = 1.f, invAspect = 1.f;
  std::vector<float> depth;  // F * H * W source depth (already inverted from disparity; invalid -> 0)

  std::vector<int> pairFrames;      // 2 per pair
  std::vector<int64_t> pairOffsets; // P + 1
  std::vector<float> pairLoc;       // 4 per constraint: loc0.xy, loc1.xy in [0,1]x[0,invAspect]
  std::vector<uint8_t> pairStatic;

  std::vector<float> sampledLoc, sampledTrip;  // results of cvdo_sample_pair / _triplet_constraints
  std::vector<uint8_t> dynMasks;  // F * dynH * dynW (dynamic_mask stream), empty = none
  int dynW = 0, dynH = 0;
  std::vector<int> tripletCenters;
  std::vector<int64_t> tripletOffsets;
  std::vector<float> tripletLoc;  // 6 per constraint
  std::vector<uint8_t> tripletStatic;

  std::vector<cvd_frame_pose> poses;
  std::vector<Xform> depthXforms, spatialXforms;
  cvd_xform_desc depthDesc{}, spatialDesc{};

  std::vector<std::array<double, 7>> poseParams;  // PoseOptimizer.h:149
  std::vector<cvd_iteration_record> records;
  cvd_solve_summary lastSummary{};
  std::string lastError;

  const float* depthImg(int f) const { return &depth[static_cast<size_t>(f) * W * Hh]; }

  // linearSolver 0 / 1: the oracle's own Ceres-default LM (block-sparse / dense Cholesky); 2: a real ceres::Solve
  void solveAny(Problem& pb, const cvd_opt_params& p) {
    if (linearSolver == 2) {
#ifdef CVDO_WITH_CERES
      pb.finalize();
      solveProblemCeres(pb, p.max_iterations, p.num_threads, &lastSummary, &records);
#else
      throw std::runtime_error("this oracle was built without Ceres (make -C oracle CERES=1 on a machine that has it)");
#endif
      return;
    }
    solveProblem(pb, p.max_iterations, p.num_threads, &lastSummary, &records, linearSolver, functionTolerance);
  }

  // ---- frame range helpers (FrameRange: ordered set of frame ids) -----------------------------------
  static std::vector<int> rangeOf(const cvd_opt_params& p, int F) {
    std::vector<int> r;
    if (!p.frame_range || p.num_range_frames <= 0) {
      for (int i = 0; i < F; ++i) r.push_back(i);
    } else {
      r.assign(p.frame_range, p.frame_range + p.num_range_frames);
      std::sort(r.begin(), r.end());
      r.erase(std::unique(r.begin(), r.end()), r.end());
    }
    return r;
  }

  void init(int numFrames, int w, int h, float asp, float invAsp) {
    F = numFrames; W = w; Hh = h; aspect = asp; invAspect = invAsp;
    depth.assign(static_cast<size_t>(F) * W * Hh, 0.f);
    poses.assign(F, cvd_frame_pose{{0, 0, 0}, {0, 0, 0, 1}, 0.f, 0.f});
    cvd_xform_desc dd{};
    dd.type = CVD_XFORM_DEPTH;
    dd.depth_type = CVD_DEPTH_IDENTITY;
    cvd_xform_desc sd{};
    sd.type = CVD_XFORM_SPATIAL;
    sd.spatial_type = CVD_SPATIAL_IDENTITY;
    resetDepthXforms(dd);
    resetSpatialXforms(sd);
  }

  // DepthStream::resetDepthXforms / resetSpatialXforms (reference lib/DepthStream.cpp:368-383)
  void resetDepthXforms(const cvd_xform_desc& d) {
    depthDesc = d;
    depthXforms.clear();
    for (int f = 0; f < F; ++f) depthXforms.push_back(Xform::create(d));
  }
  void resetSpatialXforms(const cvd_xform_desc& d) {
    spatialDesc = d;
    spatialXforms.clear();
    for (int f = 0; f < F; ++f) spatialXforms.push_back(Xform::create(d));
  }

  // DepthVideoProcessor::resetPoses, reference lib/Processor.cpp:987-1003
  void resetPoses(double focalLong) {
    for (int f = 0; f < F; ++f) {
      cvd_frame_pose& p = poses[f];
      p.position[0] = p.position[1] = p.position[2] = 0.f;
      p.orientation[0] = p.orientation[1] = p.orientation[2] = 0.f;
      p.orientation[3] = 1.f;
      const float focal = static_cast<float>(focalLong);
      if (aspect >= 1.f) {
        p.hfov = std::atan(focal) * 2.f;
        p.vfov = std::atan(focal / aspect) * 2.f;
      } else {
        p.hfov = std::atan(focal * aspect) * 2.f;
        p.vfov = std::atan(focal) * 2.f;
      }
    }
  }

  // DepthVideoProcessor::gridXformSplit, reference lib/Processor.cpp:888-985
  void gridXformSplit(const cvd_xform_desc& nd) {
    if (nd.depth_type != CVD_DEPTH_GRID) throw std::runtime_error("Transform type must be a grid type.");
    const cvd_xform_desc prev = depthDesc;
    if (prev.depth_type != CVD_DEPTH_GLOBAL && prev.depth_type != CVD_DEPTH_GRID)
      throw std::runtime_error("Can only split global or grid type transforms.");
    if (nd.value_xform != prev.value_xform)
      throw std::runtime_error("Old and new transforms must use same value transform.");
    if (prev.depth_type != CVD_DEPTH_GLOBAL &&
        (prev.grid_size[0] > nd.grid_size[0] || prev.grid_size[1] > nd.grid_size[1]))
      throw std::runtime_error(
          "New transform must have at least the same number of rows and columns as the old transform.");
    std::vector<Xform> old = depthXforms;
    resetDepthXforms(nd);
    const int newCols = nd.grid_size[0], newRows = nd.grid_size[1];
    for (int f = 0; f < F; ++f) {
      const Xform& px = old[f];
      Xform& nx = depthXforms[f];
      const int N = nx.blockSize;
      for (int row = 0; row < newRows; ++row) {
        for (int col = 0; col < newCols; ++col) {
          const int idx = col + row * newCols;
          double* dst = &nx.params[static_cast<size_t>(idx) * N];
          if (prev.depth_type == CVD_DEPTH_GLOBAL) {
            for (int i = 0; i < N; ++i) dst[i] = px.params[i];
          } else {
            const int prevRows = prev.grid_size[1], prevCols = prev.grid_size[0];
            const double maxx = std::nextafter(static_cast<double>(prevCols - 1), 0.0);
            const double maxy = std::nextafter(static_cast<double>(prevRows - 1), 0.0);
            const double sx = std::min(col / double(newCols - 1) * (prevCols - 1), maxx);
            const double sy = std::min(row / double(newRows - 1) * (prevRows - 1), maxy);
            const int ix = static_cast<int>(sx), iy = static_cast<int>(sy);
            const double rx = sx - ix, ry = sy - iy;
            const double* b0 = &px.params[static_cast<size_t>(ix + iy * prevCols) * N];
            const double* b1 = &px.params[static_cast<size_t>((ix + 1) + iy * prevCols) * N];
            const double* b2 = &px.params[static_cast<size_t>(ix + (iy + 1) * prevCols) * N];
            const double* b3 = &px.params[static_cast<size_t>((ix + 1) + (iy + 1) * prevCols) * N];
            // The reference mixes 1.f and double here (:965-968): (1.f - rx) is evaluated in double.
            const double w0 = (1.f - rx) * (1.f - ry);
            const double w1 = rx * (1.f - ry);
            const double w2 = (1.f - rx) * ry;
            const double w3 = rx * ry;
            for (int i = 0; i < N; ++i) dst[i] = b0[i] * w0 + b1[i] * w1 + b2[i] * w2 + b3[i] * w3;
          }
        }
      }
    }
  }

  // DepthVideoPoseOptimizer ctor, reference lib/PoseOptimizer.cpp:748-783
  void posesToParams() {
    poseParams.resize(F);
    for (int f = 0; f < F; ++f) {
      const cvd_frame_pose& p = poses[f];
      auto& pose = poseParams[f];
      pose[0] = p.position[0];
      pose[1] = p.position[1];
      pose[2] = p.position[2];
      const double q[4] = {p.orientation[0], p.orientation[1], p.orientation[2], p.orientation[3]};
      const double ex[3] = {1, 0, 0}, ey[3] = {0, 1, 0}, ezn[3] = {0, 0, -1};
      double right[3], up[3], front[3];
      quatRotate(q, ex, right);
      quatRotate(q, ey, up);
      quatRotate(q, ezn, front);
      double R[9];
      for (int i = 0; i < 3; ++i) {
        R[i + 0] = right[i];
        R[i + 3] = up[i];
        R[i + 6] = -front[i];
      }
      rotationMatrixToAngleAxis(R, &pose[3]);
      pose[6] = std::tan(p.vfov / 2.0);
    }
  }

  // pose write-back, reference lib/PoseOptimizer.cpp:964-987
  void paramsToPoses(const cvd_opt_params& params) {
    for (int f : rangeOf(params, F)) {
      const auto& pose = poseParams[f];
      cvd_frame_pose& p = poses[f];
      p.position[0] = static_cast<float>(pose[0]);
      p.position[1] = static_cast<float>(pose[1]);
      p.position[2] = static_cast<float>(pose[2]);
      double R[9], q[4];
      angleAxisToRotationMatrix(&pose[3], R);
      rotationMatrixToEigenQuaternion(R, q);
      for (int i = 0; i < 4; ++i) p.orientation[i] = static_cast<float>(q[i]);
      const double fsrc = (params.intr_opt == CVD_INTR_SHARED) ? poseParams[0][6] : pose[6];
      p.vfov = static_cast<float>(std::atan(fsrc) * 2.f);
      p.hfov = static_cast<float>(std::atan(fsrc * aspect) * 2.f);
    }
  }

  // Observation ctor, reference lib/PoseOptimizer.cpp:104-127 (float arithmetic, truncating fetch: q1)
  Obs makeObs(int frame, const float loc[2]) const {
    Obs o;
    o.ndc[0] = -1.f + 2.f * loc[0];
    o.ndc[1] = 1.f - 2.f * loc[1] / invAspect;
    int ix = static_cast<int>(loc[0] * W);
    int iy = static_cast<int>(loc[1] / invAspect * Hh);
    // cv::Mat::at is unchecked in the reference; clamp so that the oracle never reads out of bounds.
    ix = std::min(std::max(ix, 0), W - 1);
    iy = std::min(std::max(iy, 0), Hh - 1);
    o.sourceDepth = depthImg(frame)[static_cast<size_t>(iy) * W + ix];
    const Xform& dx = depthXforms[frame];
    o.valueType = dx.desc.value_xform;
    o.depthType = dx.desc.depth_type;
    dx.depthGather(o.sourceDepth, o.ndc[0], o.ndc[1], o.dg);
    spatialXforms[frame].spatialGather(o.ndc[0], o.ndc[1], o.sg);
    return o;
  }

  int B() const { return 7 + depthXforms[0].numBlocks * depthXforms[0].blockSize +
                         spatialXforms[0].numBlocks * spatialXforms[0].blockSize; }
  int canonDepth(int k) const { return 7 + k * depthXforms[0].blockSize; }
  int canonSpatial(int k) const {
    return 7 + depthXforms[0].numBlocks * depthXforms[0].blockSize + 2 * k;
  }

  int poseBlock(Problem& pb, int f) { return pb.blockId(poseParams[f].data(), 6, f, f * B()); }
  int focalBlock(Problem& pb, int f) { return pb.blockId(&poseParams[f][6], 1, f, f * B() + 6); }
  int depthBlock(Problem& pb, int f, int k) {
    Xform& x = depthXforms[f];
    return pb.blockId(&x.params[static_cast<size_t>(k) * x.blockSize], x.blockSize, f,
                      f * B() + canonDepth(k));
  }
  int spatialBlock(Problem& pb, int f, int k) {
    Xform& x = spatialXforms[f];
    return pb.blockId(&x.params[static_cast<size_t>(k) * 2], 2, f, f * B() + canonSpatial(k));
  }

  void appendObsBlocks(Problem& pb, int f, const Obs& o, std::vector<int>& blocks, std::vector<int>& sizes) {
    blocks.push_back(poseBlock(pb, f));
    sizes.push_back(6);
    for (int i = 0; i < o.dg.n; ++i) {
      blocks.push_back(depthBlock(pb, f, o.dg.idx[i]));
      sizes.push_back(depthXforms[f].blockSize);
    }
    for (int i = 0; i < o.sg.n; ++i) {
      blocks.push_back(spatialBlock(pb, f, o.sg.idx[i]));
      sizes.push_back(2);
    }
  }

  double vFocal(const cvd_opt_params& p) const {
    const double a = aspect;
    return (a >= 1.f ? p.focal_long / a : p.focal_long);
  }

  static bool validDepth(float d) { return std::isfinite(d) && d > 0; }

  // Pair order = std::map<std::pair<int,int>> order (reference lib/FlowConstraints.h:149).
  std::vector<int> sortedPairs() const {
    const int P = static_cast<int>(pairFrames.size() / 2);
    std::vector<int> order(P);
    for (int i = 0; i < P; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      if (pairFrames[2 * a] != pairFrames[2 * b]) return pairFrames[2 * a] < pairFrames[2 * b];
      return pairFrames[2 * a + 1] < pairFrames[2 * b + 1];
    });
    return order;
  }

  // addStaticSceneLoss, reference lib/PoseOptimizer.cpp:1149-1240
  void addStaticSceneLoss(Problem& pb, const cvd_opt_params& p, const std::vector<char>& inRange) {
    for (int pi : sortedPairs()) {
      const int f0 = pairFrames[2 * pi], f1 = pairFrames[2 * pi + 1];
      if (!inRange[f0] || !inRange[f1]) continue;
      for (int64_t c = pairOffsets[pi]; c < pairOffsets[pi + 1]; ++c) {
        if (!pairStatic[c]) continue;
        Obs o0 = makeObs(f0, &pairLoc[4 * c]);
        Obs o1 = makeObs(f1, &pairLoc[4 * c + 2]);
        if (!validDepth(o0.sourceDepth) || !validDepth(o1.sourceDepth)) continue;  // q5
        ResidualBlock rb;
        std::vector<int> sizes;
        appendObsBlocks(pb, f0, o0, rb.blocks, sizes);
        appendObsBlocks(pb, f1, o1, rb.blocks, sizes);
        if (p.intr_opt == CVD_INTR_SHARED) {
          rb.blocks.push_back(focalBlock(pb, 0));  // q7
          sizes.push_back(1);
        } else if (p.intr_opt == CVD_INTR_PER_FRAME) {
          rb.blocks.push_back(focalBlock(pb, f0));
          sizes.push_back(1);
          rb.blocks.push_back(focalBlock(pb, f1));
          sizes.push_back(1);
        }
        auto cf = std::make_unique<AutoDiff<StaticSceneCost>>(
            StaticSceneCost{o0, o1, vFocal(p), static_cast<double>(aspect), p.intr_opt,
                            p.static_loss_type, p.static_spatial_weight, p.static_depth_weight});
        cf->numResiduals = 3;
        cf->blockSizes = sizes;
        rb.cost = std::move(cf);
        rb.lossKind = robustLoss == 1 ? LOSS_HUBER : LOSS_CAUCHY;  // (reference: always CauchyLoss, :1220)
        rb.lossParam = p.robustness;
        pb.residuals.push_back(std::move(rb));
      }
    }
  }

  // addSceneFlowSmoothnessLoss, reference lib/PoseOptimizer.cpp:1242-1339 (loop bound quirk q8)
  void addSceneFlowSmoothnessLoss(Problem& pb, const cvd_opt_params& p, const std::vector<int>& range,
                                  const std::vector<char>& inRange) {
    if (range.empty()) return;
    std::map<int, int> tripletIndex;
    for (size_t i = 0; i < tripletCenters.size(); ++i) tripletIndex[tripletCenters[i]] = static_cast<int>(i);
    for (int frame = range.front(); frame < range.back() - 1; ++frame) {
      if (!inRange[frame] || !inRange[frame + 1] || !inRange[frame + 2]) continue;
      auto it = tripletIndex.find(frame + 1);
      if (it == tripletIndex.end()) throw std::runtime_error("Missing triplet constraints.");
      const int ti = it->second;
      for (int64_t c = tripletOffsets[ti]; c < tripletOffsets[ti + 1]; ++c) {
        Obs o0 = makeObs(frame + 0, &tripletLoc[6 * c]);
        Obs o1 = makeObs(frame + 1, &tripletLoc[6 * c + 2]);
        Obs o2 = makeObs(frame + 2, &tripletLoc[6 * c + 4]);
        if (!validDepth(o0.sourceDepth) || !validDepth(o1.sourceDepth) || !validDepth(o2.sourceDepth))
          continue;
        ResidualBlock rb;
        std::vector<int> sizes;
        appendObsBlocks(pb, frame + 0, o0, rb.blocks, sizes);
        appendObsBlocks(pb, frame + 1, o1, rb.blocks, sizes);
        appendObsBlocks(pb, frame + 2, o2, rb.blocks, sizes);
        if (p.intr_opt == CVD_INTR_SHARED) {
          rb.blocks.push_back(focalBlock(pb, 0));
          sizes.push_back(1);
        } else if (p.intr_opt == CVD_INTR_PER_FRAME) {
          for (int k = 0; k < 3; ++k) {
            rb.blocks.push_back(focalBlock(pb, frame + k));
            sizes.push_back(1);
          }
        }
        auto cf = std::make_unique<AutoDiff<SceneFlowSmoothnessLoss>>(SceneFlowSmoothnessLoss{
            o0, o1, o2, vFocal(p), static_cast<double>(aspect), p.intr_opt, p.smooth_loss_type});
        cf->numResiduals = 3;
        cf->blockSizes = sizes;
        rb.cost = std::move(cf);
        rb.lossKind = LOSS_SCALED;
        rb.lossParam = tripletStatic[c] ? p.smooth_static_weight : p.smooth_dynamic_weight;
        pb.residuals.push_back(std::move(rb));
      }
    }
  }

  // addScaleRegularization, reference lib/PoseOptimizer.cpp:1341-1415
  void addScaleRegularization(Problem& pb, const cvd_opt_params& p, const std::vector<int>& range) {
    int gridSizeX = p.scale_reg_grid_size;
    int gridSizeY = static_cast<int>(std::round(static_cast<float>(gridSizeX) * invAspect));
    if (aspect <= 1.f) std::swap(gridSizeX, gridSizeY);
    for (int f : range) {
      std::vector<float> samples(depthImg(f), depthImg(f) + static_cast<size_t>(W) * Hh);
      std::nth_element(samples.begin(), samples.begin() + samples.size() / 2, samples.end());
      const double medianDepth = samples[samples.size() / 2];
      for (int y = 0; y < gridSizeY; ++y) {
        for (int x = 0; x < gridSizeX; ++x) {
          const float lx = -1.f + 2.f * x / (gridSizeX - 1);
          const float ly = -1.f + 2.f * y / (gridSizeY - 1);
          Obs o;
          o.ndc[0] = lx;
          o.ndc[1] = ly;
          o.sourceDepth = static_cast<float>(medianDepth);
          o.valueType = depthXforms[f].desc.value_xform;
          o.depthType = depthXforms[f].desc.depth_type;
          depthXforms[f].depthGather(o.sourceDepth, lx, ly, o.dg);
          o.sg.n = 0;
          ResidualBlock rb;
          std::vector<int> sizes;
          for (int i = 0; i < o.dg.n; ++i) {
            rb.blocks.push_back(depthBlock(pb, f, o.dg.idx[i]));
            sizes.push_back(depthXforms[f].blockSize);
          }
          auto cf = std::make_unique<AutoDiff<TargetDisparityCost>>(TargetDisparityCost{o, 1.0});
          cf->numResiduals = 1;
          cf->blockSizes = sizes;
          rb.cost = std::move(cf);
          rb.lossKind = LOSS_SCALED;
          rb.lossParam = p.scale_reg;
          pb.residuals.push_back(std::move(rb));
        }
      }
    }
  }

  // addPositionRegularization, reference lib/PoseOptimizer.cpp:1417-1447
  void addPositionRegularization(Problem& pb, const cvd_opt_params& p, const std::vector<int>& range,
                                 const std::vector<char>& inRange) {
    if (range.empty()) return;
    for (int frame = range.front(); frame < range.back() - 1; ++frame) {
      if (!inRange[frame] || !inRange[frame + 1] || !inRange[frame + 2]) continue;
      ResidualBlock rb;
      for (int k = 0; k < 3; ++k) rb.blocks.push_back(poseBlock(pb, frame + k));
      auto cf = std::make_unique<AutoDiff<ParameterRegularizationCost>>(ParameterRegularizationCost{3});
      cf->numResiduals = 3;
      cf->blockSizes = {6, 6, 6};
      rb.cost = std::move(cf);
      rb.lossKind = LOSS_SCALED;
      rb.lossParam = p.position_reg;
      pb.residuals.push_back(std::move(rb));
    }
  }

  // addDepthDeformRegularization / addSpatialDeformRegularization, reference :1449-1522.
  // adaptive > 0 (depth transforms only): AdaptiveDeformationCost with the frame's dynamic mask, :1469-1484.
  void addDeformRegularization(Problem& pb, const std::vector<int>& range, bool depthKind, double weight,
                               double adaptive = 0.0) {
    for (int f : range) {
      Xform& x = depthKind ? depthXforms[f] : spatialXforms[f];
      if (x.numDeformationResiduals() <= 0) continue;
      ResidualBlock rb;
      std::vector<int> sizes;
      for (int k = 0; k < x.numBlocks; ++k) {
        rb.blocks.push_back(depthKind ? depthBlock(pb, f, k) : spatialBlock(pb, f, k));
        sizes.push_back(x.blockSize);
      }
      if (depthKind && adaptive > 0.0) {
        if (dynMasks.empty()) throw std::runtime_error("Adaptive smoothness requires a dynamic mask stream.");
        auto cf = std::make_unique<AutoDiff<AdaptiveDeformationCost>>(AdaptiveDeformationCost(
            &x, dynMasks.data() + static_cast<size_t>(f) * dynW * dynH, dynW, dynH, weight, adaptive));
        cf->numResiduals = x.numDeformationResiduals();
        cf->blockSizes = sizes;
        rb.cost = std::move(cf);
      } else {
        auto cf = std::make_unique<AutoDiff<DeformationCost>>(DeformationCost{&x, weight});
        cf->numResiduals = x.numDeformationResiduals();
        cf->blockSizes = sizes;
        rb.cost = std::move(cf);
      }
      rb.lossKind = LOSS_NONE;
      pb.residuals.push_back(std::move(rb));
    }
  }

  // addFocalRegularization, reference lib/PoseOptimizer.cpp:1524-1549
  void addFocalRegularization(Problem& pb, const cvd_opt_params& p, const std::vector<int>& range) {
    if (p.intr_opt == CVD_INTR_FIXED) return;
    for (int f : range) {
      ResidualBlock rb;
      rb.blocks.push_back(focalBlock(pb, f));
      auto cf = std::make_unique<AutoDiff<TargetFocalCost>>(TargetFocalCost{vFocal(p)});
      cf->numResiduals = 1;
      cf->blockSizes = {1};
      rb.cost = std::move(cf);
      rb.lossKind = LOSS_SCALED;
      rb.lossParam = p.focal_reg;
      pb.residuals.push_back(std::move(rb));
    }
  }

  // Problem of poseOptimizationStep, reference lib/PoseOptimizer.cpp:890-952
  void buildPoseProblem(Problem& pb, const cvd_opt_params& p, double depthDeformReg) {
    const std::vector<int> range = rangeOf(p, F);
    std::vector<char> inRange(F, 0);
    for (int f : range) inRange[f] = 1;
    addStaticSceneLoss(pb, p, inRange);
    if (p.smooth_static_weight > 0.0 || p.smooth_dynamic_weight > 0.0)
      addSceneFlowSmoothnessLoss(pb, p, range, inRange);
    if (p.position_reg > 0.0) addPositionRegularization(pb, p, range, inRange);
    if (depthDeformReg > 0.0) addDeformRegularization(pb, range, true, depthDeformReg, p.adaptive_deformation_cost);
    if (p.spatial_deform_reg > 0.0) addDeformRegularization(pb, range, false, p.spatial_deform_reg);
    if (p.fix_poses)
      for (int f : range) pb.setConstant(poseParams[f].data());
    if (p.fix_depth_xforms) {
      for (int f : range) {
        Xform& x = depthXforms[f];
        for (int k = 0; k < x.numBlocks; ++k) pb.setConstant(&x.params[static_cast<size_t>(k) * x.blockSize]);
      }
    } else if (p.scale_reg > 0.0) {
      addScaleRegularization(pb, p, range);
    }
    if (p.fix_spatial_xforms) {
      for (int f : range) {
        Xform& x = spatialXforms[f];
        for (int k = 0; k < x.numBlocks; ++k) pb.setConstant(&x.params[static_cast<size_t>(k) * 2]);
      }
    }
    if (p.focal_reg > 0.0) addFocalRegularization(pb, p, range);
  }

  // poseOptimizationStep, reference lib/PoseOptimizer.cpp:890-990
  void poseOptimizationStep(const cvd_opt_params& p, double depthDeformReg) {
    Problem pb;
    buildPoseProblem(pb, p, depthDeformReg);
    solveAny(pb, p);
    paramsToPoses(p);
  }

  // poseOptimization, reference lib/PoseOptimizer.cpp:788-888
  void poseOptimization(const cvd_opt_params& p) {
    posesToParams();
    records.clear();
    int ctfRows = p.ctf_long, ctfCols = p.ctf_short;
    int dsoRows = p.dso_long, dsoCols = p.dso_short;
    if (aspect >= 1.f) {
      std::swap(ctfCols, ctfRows);
      std::swap(dsoCols, dsoRows);
    }
    auto gridSize = [&](const cvd_xform_desc& d, int g[3]) {
      if (d.depth_type == CVD_DEPTH_GRID) {
        g[0] = d.grid_size[0]; g[1] = d.grid_size[1]; g[2] = d.grid_size[2];
      } else {
        g[0] = g[1] = g[2] = 1;
      }
    };
    int initGrid[3];
    gridSize(depthDesc, initGrid);
    if (p.deferred_spatial_opt) {
      cvd_xform_desc sd{};
      sd.type = CVD_XFORM_SPATIAL;
      sd.spatial_type = CVD_SPATIAL_IDENTITY;
      resetSpatialXforms(sd);
    }
    for (int step = 0; step < p.num_steps; ++step) {
      const double stepIter = (p.num_steps > 1 ? step / double(p.num_steps - 1) : 0.0);
      double depthDeformReg = p.depth_deform_reg_final;
      if (p.graduate_depth_deform_reg) {
        const double a = std::log(p.depth_deform_reg_initial);
        const double b = std::log(p.depth_deform_reg_final);
        depthDeformReg = std::exp(a + (b - a) * stepIter);
      }
      poseOptimizationStep(p, depthDeformReg);
      if (p.coarse_to_fine && step < p.num_steps - 1) {
        const double ctfIter = (step + 1) / double(p.num_steps - 1);
        cvd_xform_desc nd = depthDesc;
        if (nd.depth_type == CVD_DEPTH_GLOBAL) nd.depth_type = CVD_DEPTH_GRID;
        nd.grid_size[0] = static_cast<int>(initGrid[0] + (ctfCols - initGrid[0]) * ctfIter + 0.5);
        nd.grid_size[1] = static_cast<int>(initGrid[1] + (ctfRows - initGrid[1]) * ctfIter + 0.5);
        nd.grid_size[2] = initGrid[2];
        gridXformSplit(nd);
      }
    }
    if (p.deferred_spatial_opt) {
      cvd_xform_desc sd{};
      sd.type = CVD_XFORM_SPATIAL;
      sd.spatial_type = CVD_SPATIAL_BICUBIC_GRID;
      sd.grid_size[1] = dsoRows;
      sd.grid_size[0] = dsoCols;
      resetSpatialXforms(sd);
      poseOptimizationStep(p, p.depth_deform_reg_final);
    }
  }

  // normalizeDepth, reference lib/PoseOptimizer.cpp:992-1147
  void normalizeDepth(const cvd_opt_params& p) {
    posesToParams();
    records.clear();
    const std::vector<int> range = rangeOf(p, F);
    std::vector<char> inRange(F, 0);
    for (int f : range) inRange[f] = 1;
    Problem pb;
    for (int pi : sortedPairs()) {
      const int i0 = pairFrames[2 * pi], i1 = pairFrames[2 * pi + 1];
      if (!inRange[i0] || !inRange[i1]) continue;
      if (p.normalize_depth_from_first_frame) break;
      for (int64_t c = pairOffsets[pi]; c < pairOffsets[pi + 1]; ++c) {
        Obs o0 = makeObs(i0, &pairLoc[4 * c]);
        Obs o1 = makeObs(i1, &pairLoc[4 * c + 2]);
        if (!validDepth(o0.sourceDepth) || !validDepth(o1.sourceDepth)) continue;
        o0.sg.n = 0;
        o1.sg.n = 0;
        ResidualBlock rb;
        std::vector<int> sizes;
        for (int i = 0; i < o0.dg.n; ++i) {
          rb.blocks.push_back(depthBlock(pb, i0, o0.dg.idx[i]));
          sizes.push_back(depthXforms[i0].blockSize);
        }
        for (int i = 0; i < o1.dg.n; ++i) {
          rb.blocks.push_back(depthBlock(pb, i1, o1.dg.idx[i]));
          sizes.push_back(depthXforms[i1].blockSize);
        }
        auto cf = std::make_unique<AutoDiff<DisparityDissimilarityCost>>(DisparityDissimilarityCost{o0, o1});
        cf->numResiduals = 1;
        cf->blockSizes = sizes;
        rb.cost = std::move(cf);
        rb.lossKind = LOSS_CAUCHY;
        rb.lossParam = p.robustness;
        pb.residuals.push_back(std::move(rb));
      }
    }
    if (p.scale_reg > 0.0) addScaleRegularization(pb, p, range);
    if (p.depth_deform_reg_initial > 0.0)
      addDeformRegularization(pb, range, true, p.depth_deform_reg_initial, p.adaptive_deformation_cost);
    for (int f : range) {
      Xform& x = depthXforms[f];
      for (int k = 0; k < x.numBlocks; ++k)
        pb.setLowerBound0(&x.params[static_cast<size_t>(k) * x.blockSize], 0.0);
    }
    solveAny(pb, p);
    if (p.normalize_depth_from_first_frame && !range.empty()) {
      const int first = range.front();
      for (int f : range)
        if (f != first) depthXforms[f].params = depthXforms[first].params;
    }
  }

  // Parity hook: evaluate the poseOptimizationStep problem at the current double-precision state and
  // return cost / gradient / J^T J in the canonical per-frame layout [t(3) w(3) f(1) theta.. phi..].
  void evaluate(const cvd_opt_params& p, double depthDeformReg, const double* pose7, double* cost,
                int* numResidualBlocks, double* gradient /*F*B*/, double* hdiag /*F*B*B*/,
                double* hfull /*(F*B)^2*/) {
    if (pose7) {
      poseParams.resize(F);
      for (int f = 0; f < F; ++f)
        for (int i = 0; i < 7; ++i) poseParams[f][i] = pose7[f * 7 + i];
    } else {
      posesToParams();
    }
    Problem pb;
    buildPoseProblem(pb, p, depthDeformReg);
    pb.finalize();
    Evaluation ev;
    const bool derivs = gradient || hdiag || hfull;
    evaluateProblem(pb, p.num_threads, derivs, ev);
    if (cost) *cost = ev.cost;
    if (numResidualBlocks) *numResidualBlocks = static_cast<int>(pb.residuals.size());
    const int Bf = B();
    const size_t NC = static_cast<size_t>(F) * Bf;
    if (gradient) std::fill(gradient, gradient + NC, 0.0);
    if (hdiag) std::fill(hdiag, hdiag + NC * Bf, 0.0);
    if (hfull) std::fill(hfull, hfull + NC * NC, 0.0);
    if (!derivs) return;
    const int n = pb.numActive;
    std::vector<int> canon(n, -1);
    for (const auto& b : pb.blocks)
      if (b.offset >= 0)
        for (int i = 0; i < b.size; ++i) canon[b.offset + i] = b.canon + i;
    if (gradient)
      for (int i = 0; i < n; ++i) gradient[canon[i]] = ev.g[i];
    const BlockSym& Hs = ev.H;
    for (int I = 0; I < Hs.nb; ++I)
      for (int e = Hs.rowPtr[I]; e < Hs.rowPtr[I + 1]; ++e) {
        const int J = Hs.rowCol[e];
        if (!hfull && J != I) continue;
        const int ni = Hs.size(I), nj = Hs.size(J);
        const double* Bv = Hs.val.data() + Hs.blkOff[e];
        for (int a = 0; a < ni; ++a)
          for (int b = 0; b < nj; ++b) {
            const double h = Bv[static_cast<size_t>(a) * nj + b];
            if (h == 0.0) continue;
            const int ci = canon[Hs.off[I] + a], cj = canon[Hs.off[J] + b];
            if (hfull) {
              hfull[static_cast<size_t>(ci) * NC + cj] = h;
              hfull[static_cast<size_t>(cj) * NC + ci] = h;
            }
            if (hdiag && ci / Bf == cj / Bf)
              hdiag[(static_cast<size_t>(ci / Bf) * Bf + ci % Bf) * Bf + cj % Bf] = h;
          }
      }
  }

  // Parity hook for the reference-held residual pin (tests/test_reference_residuals.py): every StaticSceneCost block of the
  // poseOptimizationStep problem at the current state, WITHOUT the robust loss -- frames (a, b); the two observations' NDC
  // (float, as stored) and warped NDC (obsToCamera x / y: NDC + spatial warp) and deformed depths (obsToCamera z); the three
  // residuals; and their dual-number Jacobian with respect to [pose_a(6) | pose_b(6) | vfocal_a | vfocal_b] (14 columns; the
  // focal columns stay 0 under Fixed intrinsics, both hold the one shared column under Shared).  Returns the number of blocks;
  // with null outputs only counts.
  int staticResiduals(const cvd_opt_params& p, double depthDeformReg, const double* pose7, int maxBlocks, int32_t* frames,
                      double* obs /*10 per block*/, double* res /*3*/, double* jac /*3 x 14*/) {
    if (pose7) {
      poseParams.resize(F);
      for (int f = 0; f < F; ++f)
        for (int i = 0; i < 7; ++i) poseParams[f][i] = pose7[f * 7 + i];
    } else {
      posesToParams();
    }
    Problem pb;
    buildPoseProblem(pb, p, depthDeformReg);
    pb.finalize();
    int n = 0;
    for (const auto& rb : pb.residuals) {
      const auto* cf = dynamic_cast<const AutoDiff<StaticSceneCost>*>(rb.cost.get());
      if (!cf) continue;
      if (frames && n < maxBlocks) {
        const StaticSceneCost& sc = cf->f;
        const int nb = static_cast<int>(rb.blocks.size());
        int total = 0;
        std::vector<int> start(nb);
        for (int b = 0; b < nb; ++b) { start[b] = total; total += cf->blockSizes[b]; }
        std::vector<const double*> pd(nb);
        for (int b = 0; b < nb; ++b) pd[b] = pb.blocks[rb.blocks[b]].ptr;
        std::vector<double> J(static_cast<size_t>(3) * total);
        double r[3];
        evaluateCostFunctionAt(*cf, pd.data(), r, J.data());
        const int b0 = 0, b1 = sc.obs0.numBlocks(), bf = b1 + sc.obs1.numBlocks();
        frames[2 * n] = pb.blocks[rb.blocks[b0]].frame;
        frames[2 * n + 1] = pb.blocks[rb.blocks[b1]].frame;
        {
          int off = 0;
          ObsParams<double> p0 = unpack(off, pd.data(), sc.obs0);
          ObsParams<double> p1 = unpack(off, pd.data(), sc.obs1);
          double c0[3], c1[3];
          obsToCamera(sc.obs0, p0, c0);
          obsToCamera(sc.obs1, p1, c1);
          double* o = obs + static_cast<size_t>(10) * n;
          o[0] = sc.obs0.ndc[0]; o[1] = sc.obs0.ndc[1]; o[2] = sc.obs1.ndc[0]; o[3] = sc.obs1.ndc[1];
          o[4] = c0[0]; o[5] = c0[1]; o[6] = c0[2]; o[7] = c1[2];
          o[8] = c1[0]; o[9] = c1[1];   // (the target's WARPED NDC: differs from ndc_b under a spatial transform)
        }
        for (int k = 0; k < 3; ++k) {
          res[3 * n + k] = r[k];
          double* row = jac + (static_cast<size_t>(3) * n + k) * 14;
          for (int i = 0; i < 14; ++i) row[i] = 0.0;
          for (int i = 0; i < 6; ++i) {
            row[i] = J[static_cast<size_t>(k) * total + start[b0] + i];
            row[6 + i] = J[static_cast<size_t>(k) * total + start[b1] + i];
          }
          if (p.intr_opt == CVD_INTR_SHARED) {
            row[12] = row[13] = J[static_cast<size_t>(k) * total + start[bf]];
          } else if (p.intr_opt == CVD_INTR_PER_FRAME) {
            row[12] = J[static_cast<size_t>(k) * total + start[bf]];
            row[13] = J[static_cast<size_t>(k) * total + start[bf + 1]];
          }
        }
      }
      ++n;
    }
    return n;
  }

  // The depth-parameter columns of every StaticSceneCost block in a form code held by the reference can check
  // (tests/test_reference_residuals.py: central differences of utils/geometry.py along the two deformed depths).  The depth functor
  // is D = sum_k w_k V(d, theta_k) (reference lib/DepthMapTransform.cpp:597-606, lib/ValueTransform.h:59-81), so the dual-number
  // column of theta_k[0] is (d r / d D) w_k d and that of theta_k[1] (ScaleShift) is (d r / d D) w_k.  Per block and side
  // (0 = source, 1 = target):  jd[3] = column of the tap with the largest weight divided by that factor = d r / d D;
  // euler[3] = sum over the side's depth columns of column x parameter (Scale: = (d r / d D) D); tapdev = largest deviation of any
  // other tap's quotient from jd, relative to |jd|_max (0 when the columns are exactly rank one in (residual, tap)).
  int staticResidualsDepth(const cvd_opt_params& p, double depthDeformReg, const double* pose7, int maxBlocks, double* jd /*3 x 2*/,
                           double* euler /*3 x 2*/, double* tapdev /*2*/) {
    if (pose7) {
      poseParams.resize(F);
      for (int f = 0; f < F; ++f)
        for (int i = 0; i < 7; ++i) poseParams[f][i] = pose7[f * 7 + i];
    } else {
      posesToParams();
    }
    Problem pb;
    buildPoseProblem(pb, p, depthDeformReg);
    pb.finalize();
    int n = 0;
    for (const auto& rb : pb.residuals) {
      const auto* cf = dynamic_cast<const AutoDiff<StaticSceneCost>*>(rb.cost.get());
      if (!cf) continue;
      if (jd && n < maxBlocks) {
        const StaticSceneCost& sc = cf->f;
        const int nb = static_cast<int>(rb.blocks.size());
        int total = 0;
        std::vector<int> start(nb);
        for (int b = 0; b < nb; ++b) { start[b] = total; total += cf->blockSizes[b]; }
        std::vector<const double*> pd(nb);
        for (int b = 0; b < nb; ++b) pd[b] = pb.blocks[rb.blocks[b]].ptr;
        std::vector<double> J(static_cast<size_t>(3) * total);
        double r[3];
        evaluateCostFunctionAt(*cf, pd.data(), r, J.data());
        const Obs* obs[2] = {&sc.obs0, &sc.obs1};
        const int first[2] = {1, sc.obs0.numBlocks() + 1};   // (block 0 of a side is its pose)
        for (int side = 0; side < 2; ++side) {
          const Obs& o = *obs[side];
          double* q = jd + (static_cast<size_t>(n) * 3) * 2;
          double* e = euler + (static_cast<size_t>(n) * 3) * 2;
          for (int k = 0; k < 3; ++k) { q[k * 2 + side] = 0.0; e[k * 2 + side] = 0.0; }
          tapdev[static_cast<size_t>(n) * 2 + side] = 0.0;
          if (o.depthType == CVD_DEPTH_IDENTITY || o.dg.n == 0) continue;
          const double d = static_cast<double>(o.sourceDepth);
          int best = 0;
          for (int i = 1; i < o.dg.n; ++i) if (o.dg.w[i] > o.dg.w[best]) best = i;
          auto weightOf = [&](int i) { return o.depthType == CVD_DEPTH_GLOBAL ? 1.0 : static_cast<double>(o.dg.w[i]); };
          double scale = 0.0;
          for (int k = 0; k < 3; ++k) {
            q[k * 2 + side] = J[static_cast<size_t>(k) * total + start[first[side] + best]] / (weightOf(best) * d);
            scale = std::max(scale, std::abs(q[k * 2 + side]));
          }
          for (int i = 0; i < o.dg.n; ++i) {
            const int b = first[side] + i;
            for (int k = 0; k < 3; ++k) {
              for (int a = 0; a < cf->blockSizes[b]; ++a) {
                const double col = J[static_cast<size_t>(k) * total + start[b] + a];
                e[k * 2 + side] += col * pd[b][a];
                if (weightOf(i) > 1e-9) {
                  const double quot = col / (weightOf(i) * (a == 0 ? d : 1.0));
                  tapdev[static_cast<size_t>(n) * 2 + side] =
                      std::max(tapdev[static_cast<size_t>(n) * 2 + side], std::abs(quot - q[k * 2 + side]) / std::max(scale, 1e-300));
                }
              }
            }
          }
        }
      }
      ++n;
    }
    return n;
  }

  // Development hook (tools/pcg_lab.py: preconditioner experiments on the CPU): the block-sparse normal equations of the
  // poseOptimizationStep problem at the given state, written to a file in the canonical per-frame layout --
  // i32 F, i32 B, i64 numBlocks, f64 cost, f64 gradient[F B], then per block i32 I, i32 J (I >= J), f64 [B x B] = H_IJ.
  void dumpBlocks(const cvd_opt_params& p, double depthDeformReg, const double* pose7, const char* path) {
    if (pose7) {
      poseParams.resize(F);
      for (int f = 0; f < F; ++f)
        for (int i = 0; i < 7; ++i) poseParams[f][i] = pose7[f * 7 + i];
    } else {
      posesToParams();
    }
    Problem pb;
    buildPoseProblem(pb, p, depthDeformReg);
    pb.finalize();
    Evaluation ev;
    evaluateProblem(pb, p.num_threads, true, ev);
    const int Bf = B();
    const int n = pb.numActive;
    std::vector<int> canon(n, -1);
    for (const auto& b : pb.blocks)
      if (b.offset >= 0)
        for (int i = 0; i < b.size; ++i) canon[b.offset + i] = b.canon + i;
    FILE* fp = std::fopen(path, "wb");
    if (!fp) throw std::runtime_error("dumpBlocks: cannot open the output file");
    const BlockSym& Hs = ev.H;
    const int32_t hdr[2] = {F, Bf};
    const int64_t nBlocks = static_cast<int64_t>(Hs.rowCol.size());
    std::fwrite(hdr, sizeof(int32_t), 2, fp);
    std::fwrite(&nBlocks, sizeof(int64_t), 1, fp);
    std::fwrite(&ev.cost, sizeof(double), 1, fp);
    std::vector<double> g(static_cast<size_t>(F) * Bf, 0.0);
    for (int i = 0; i < n; ++i) g[canon[i]] = ev.g[i];
    std::fwrite(g.data(), sizeof(double), g.size(), fp);
    std::vector<double> blk(static_cast<size_t>(Bf) * Bf);
    for (int I = 0; I < Hs.nb; ++I)
      for (int e = Hs.rowPtr[I]; e < Hs.rowPtr[I + 1]; ++e) {
        const int J = Hs.rowCol[e];
        const int ni = Hs.size(I), nj = Hs.size(J);
        if (ni == 0 || nj == 0) continue;
        std::fill(blk.begin(), blk.end(), 0.0);
        const double* Bv = Hs.val.data() + Hs.blkOff[e];
        const int fI = canon[Hs.off[I]] / Bf, fJ = canon[Hs.off[J]] / Bf;
        for (int a = 0; a < ni; ++a)
          for (int b = 0; b < nj; ++b)
            blk[static_cast<size_t>(canon[Hs.off[I] + a] % Bf) * Bf + canon[Hs.off[J] + b] % Bf] = Bv[static_cast<size_t>(a) * nj + b];
        const int32_t ij[2] = {fI, fJ};
        std::fwrite(ij, sizeof(int32_t), 2, fp);
        std::fwrite(blk.data(), sizeof(double), blk.size(), fp);
      }
    std::fclose(fp);
  }
};

}  // namespace cvdo

// =====================================================================================================
// C ABI (ctypes) -- mirrors include/cvd_hip.h with the prefix cvdo_
// =====================================================================================================
using cvdo::Oracle;

#define CVDO_TRY(h, ...)                                \
  try {                                                 \
    __VA_ARGS__;                                        \
    return 0;                                           \
  } catch (const std::exception& e) {                   \
    if (h) static_cast<Oracle*>(h)->lastError = e.what(); \
    return -1;                                          \
  }

extern "C" {

void* cvdo_create() { return new Oracle(); }
void cvdo_destroy(void* h) { delete static_cast<Oracle*>(h); }
const char* cvdo_last_error(void* h) { return static_cast<Oracle*>(h)->lastError.c_str(); }

int cvdo_set_robust_loss(void* h, int kind) {
  CVDO_TRY(h, {
    if (kind != 0 && kind != 1) throw std::runtime_error("robust_loss must be 0 (Cauchy) or 1 (Huber)");
    static_cast<Oracle*>(h)->robustLoss = kind;
  });
}
// Known-answer hook for the block-sparse Cholesky (tests/test_oracle_sparse.py): the symmetric positive definite matrix
// `dense` (n x n row-major, n = sum sizes) restricted to the block structure {diagonal blocks} + pairs[2 * npairs] is
// factorised and  (S A S + diag(extra)) x = b  solved in place (scale / extra may be NULL); y (may be NULL) receives
// A b through the block-sparse product.  Returns -1 when a pivot fails, else the number of blocks of L incl. fill.
int cvdo_block_sparse_solve(int nb, const int* sizes, int npairs, const int* pairs, const double* dense,
                            const double* scale, const double* extra, double* b, double* y, int numThreads) {
  try {
    std::vector<int> sz(sizes, sizes + nb);
    std::vector<std::pair<int, int>> pr;
    for (int i = 0; i < npairs; ++i) pr.push_back({pairs[2 * i], pairs[2 * i + 1]});
    cvdo::BlockSym A;
    A.build(sz, pr);
    const int n = A.n();
    for (int I = 0; I < A.nb; ++I)
      for (int e = A.rowPtr[I]; e < A.rowPtr[I + 1]; ++e) {
        const int J = A.rowCol[e];
        for (int a = 0; a < A.size(I); ++a)
          for (int c = 0; c < A.size(J); ++c)
            A.val[A.blkOff[e] + static_cast<size_t>(a) * A.size(J) + c] = dense[static_cast<size_t>(A.off[I] + a) * n + A.off[J] + c];
      }
    if (y) A.multiply(b, y);
    cvdo::BlockCholesky ch;
    ch.analyze(A);
    if (!ch.factor(A, scale, extra, numThreads)) return -1;
    ch.solve(b);
    return static_cast<int>(ch.numBlocks());
  } catch (const std::exception&) {
    return -2;
  }
}

int cvdo_set_linear_solver(void* h, int kind) {
  CVDO_TRY(h, {
    if (kind < 0 || kind > 2)
      throw std::runtime_error("linear solver must be 0 (block-sparse Cholesky), 1 (dense Cholesky) or 2 (real ceres::Solve)");
    static_cast<Oracle*>(h)->linearSolver = kind;
  });
}
int cvdo_has_ceres() {
#ifdef CVDO_WITH_CERES
  return 1;
#else
  return 0;
#endif
}
int cvdo_set_function_tolerance(void* h, double tol) { CVDO_TRY(h, static_cast<Oracle*>(h)->functionTolerance = tol); }
int cvdo_set_video(void* h, int numFrames, int width, int height, float aspect, float invAspect) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->init(numFrames, width, height, aspect, invAspect));
}
int cvdo_set_depth(void* h, int frame, const float* depth) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    if (frame < 0 || frame >= o->F) throw std::runtime_error("frame out of range");
    std::memcpy(&o->depth[static_cast<size_t>(frame) * o->W * o->Hh], depth,
                sizeof(float) * o->W * o->Hh);
  });
}
int cvdo_set_pair_constraints(void* h, int numPairs, const int32_t* pairFrames, const int64_t* offsets,
                              const float* loc4, const uint8_t* isStatic) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    o->pairFrames.assign(pairFrames, pairFrames + 2 * numPairs);
    o->pairOffsets.assign(offsets, offsets + numPairs + 1);
    const int64_t C = offsets[numPairs];
    o->pairLoc.assign(loc4, loc4 + 4 * C);
    if (isStatic) o->pairStatic.assign(isStatic, isStatic + C);
    else o->pairStatic.assign(C, 1);
  });
}
int cvdo_set_dynamic_masks(void* h, int height, int width, const uint8_t* masks) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    o->dynMasks.clear();
    if (masks) {
      o->dynW = width;
      o->dynH = height;
      o->dynMasks.assign(masks, masks + static_cast<size_t>(o->F) * width * height);
    }
  });
}
int cvdo_set_triplet_constraints(void* h, int numTriplets, const int32_t* centers, const int64_t* offsets,
                                 const float* loc6, const uint8_t* isStatic) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    o->tripletCenters.assign(centers, centers + numTriplets);
    o->tripletOffsets.assign(offsets, offsets + numTriplets + 1);
    const int64_t C = offsets[numTriplets];
    o->tripletLoc.assign(loc6, loc6 + 6 * C);
    if (isStatic) o->tripletStatic.assign(isStatic, isStatic + C);
    else o->tripletStatic.assign(C, 1);
  });
}
int cvdo_set_poses(void* h, const cvd_frame_pose* poses) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, o->poses.assign(poses, poses + o->F));
}
int cvdo_get_poses(void* h, cvd_frame_pose* poses) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, std::memcpy(poses, o->poses.data(), sizeof(cvd_frame_pose) * o->F));
}
int cvdo_reset_poses(void* h, double focalLong) { CVDO_TRY(h, static_cast<Oracle*>(h)->resetPoses(focalLong)); }
int cvdo_reset_depth_xforms(void* h, const cvd_xform_desc* d) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->resetDepthXforms(*d));
}
int cvdo_reset_spatial_xforms(void* h, const cvd_xform_desc* d) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->resetSpatialXforms(*d));
}
int cvdo_grid_xform_split(void* h, const cvd_xform_desc* d) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->gridXformSplit(*d));
}
int cvdo_get_xform_desc(void* h, int spatial, cvd_xform_desc* d) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, *d = spatial ? o->spatialDesc : o->depthDesc);
}
int cvdo_num_xform_params(void* h, int spatial) {
  Oracle* o = static_cast<Oracle*>(h);
  const auto& v = spatial ? o->spatialXforms : o->depthXforms;
  return v.empty() ? 0 : static_cast<int>(v[0].params.size());
}
int cvdo_get_xform_params(void* h, int spatial, double* out /*F x numParams*/) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    const auto& v = spatial ? o->spatialXforms : o->depthXforms;
    for (int f = 0; f < o->F; ++f) {
      const size_t np = v[f].params.size();
      std::memcpy(out + f * np, v[f].params.data(), sizeof(double) * np);
    }
  });
}
int cvdo_set_xform_params(void* h, int spatial, const double* in) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    auto& v = spatial ? o->spatialXforms : o->depthXforms;
    for (int f = 0; f < o->F; ++f) {
      const size_t np = v[f].params.size();
      std::memcpy(v[f].params.data(), in + f * np, sizeof(double) * np);
    }
  });
}
int cvdo_normalize_depth(void* h, const cvd_opt_params* p) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->normalizeDepth(*p));
}
int cvdo_pose_optimization(void* h, const cvd_opt_params* p) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->poseOptimization(*p));
}
int cvdo_pose_optimization_step(void* h, const cvd_opt_params* p, double depthDeformReg, int convertPoses) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    if (convertPoses) o->posesToParams();
    o->records.clear();
    o->poseOptimizationStep(*p, depthDeformReg);
  });
}
int cvdo_get_pose_params(void* h, double* pose7) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    if (static_cast<int>(o->poseParams.size()) != o->F) o->posesToParams();
    for (int f = 0; f < o->F; ++f)
      for (int i = 0; i < 7; ++i) pose7[f * 7 + i] = o->poseParams[f][i];
  });
}
int cvdo_set_pose_params(void* h, const double* pose7) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    o->poseParams.resize(o->F);
    for (int f = 0; f < o->F; ++f)
      for (int i = 0; i < 7; ++i) o->poseParams[f][i] = pose7[f * 7 + i];
  });
}
int cvdo_block_size(void* h) { return static_cast<Oracle*>(h)->B(); }
int cvdo_evaluate(void* h, const cvd_opt_params* p, double depthDeformReg, const double* pose7, double* cost,
                  int* numResidualBlocks, double* gradient, double* hdiag, double* hfull) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->evaluate(*p, depthDeformReg, pose7, cost, numResidualBlocks, gradient,
                                                hdiag, hfull));
}
int cvdo_static_residuals(void* h, const cvd_opt_params* p, double depthDeformReg, const double* pose7, int maxBlocks,
                          int32_t* frames, double* obs, double* res, double* jac) {
  auto* o = static_cast<Oracle*>(h);
  try {
    return o->staticResiduals(*p, depthDeformReg, pose7, maxBlocks, frames, obs, res, jac);
  } catch (const std::exception& e) {
    o->lastError = e.what();
    return -1;
  }
}
int cvdo_static_residuals_depth(void* h, const cvd_opt_params* p, double depthDeformReg, const double* pose7, int maxBlocks,
                                double* jd, double* euler, double* tapdev) {
  auto* o = static_cast<Oracle*>(h);
  try {
    return o->staticResidualsDepth(*p, depthDeformReg, pose7, maxBlocks, jd, euler, tapdev);
  } catch (const std::exception& e) {
    o->lastError = e.what();
    return -1;
  }
}
int cvdo_dump_blocks(void* h, const cvd_opt_params* p, double depthDeformReg, const double* pose7, const char* path) {
  CVDO_TRY(h, static_cast<Oracle*>(h)->dumpBlocks(*p, depthDeformReg, pose7, path));
}
int cvdo_get_summary(void* h, cvd_solve_summary* s) {
  CVDO_TRY(h, *s = static_cast<Oracle*>(h)->lastSummary);
}
int cvdo_num_records(void* h) { return static_cast<int>(static_cast<Oracle*>(h)->records.size()); }
int cvdo_get_records(void* h, cvd_iteration_record* out) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, std::memcpy(out, o->records.data(), sizeof(cvd_iteration_record) * o->records.size()));
}

// ---- constraint sampling: same signatures as include/cvd_hip.h -------------------------------------
// FlowConstraintsCollection::compute(PairKey), reference lib/FlowConstraints.cpp:400-465, and sampleConstraints,
// :352-397 (Pixel ordering :304-315, buildDiskMask :317-332, scaleConstraint :334-339).  Deviation: ties in the corner
// response keep pixel order (std::stable_sort); the reference's std::sort leaves their order unspecified.
int cvdo_sample_pair_constraints(void* h, int numPairs, const int32_t* pairFrames, const float* corner,
                                 const float* flow, const uint8_t* mask, const float* dynDist, int dynW, int dynH,
                                 int matchSeparation, float minDynamicDistance, int64_t* offsets) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    const int w = o->W, hh = o->Hh;
    const size_t npx = static_cast<size_t>(w) * hh;
    o->sampledLoc.clear();
    offsets[0] = 0;
    const int dw = dynDist ? dynW : w, dh = dynDist ? dynH : hh;
    const float scaleX = dw / float(w), scaleY = dh / float(hh);  // dynamicMaskScale, :411-413
    struct Pixel { float cornerStrength; int ix0, iy0; float fx1, fy1; };
    for (int p = 0; p < numPairs; ++p) {
      const int fa = pairFrames[2 * p], fb = pairFrames[2 * p + 1];
      const float* cornerPtr = corner + fa * npx;
      const float* fl = flow + static_cast<size_t>(p) * npx * 2;
      const uint8_t* mk = mask + static_cast<size_t>(p) * npx;
      std::vector<Pixel> pixels;
      for (int iy0 = 0; iy0 < hh; ++iy0) {
        // (cv::Mat access is unchecked in the reference; a half-resolution mask can be indexed one past its end: clamp)
        const int iy0s = std::min(static_cast<int>(iy0 * scaleY + 0.5f), dh - 1);
        for (int ix0 = 0; ix0 < w; ++ix0) {
          const int ix0s = std::min(static_cast<int>(ix0 * scaleX + 0.5f), dw - 1);
          const float d0 = dynDist ? dynDist[static_cast<size_t>(fa) * dw * dh + static_cast<size_t>(iy0s) * dw + ix0s] : FLT_MAX;
          if (mk[iy0 * w + ix0] && d0 > minDynamicDistance) {
            const float fx1 = ix0 + fl[(static_cast<size_t>(iy0) * w + ix0) * 2];
            const float fy1 = iy0 + fl[(static_cast<size_t>(iy0) * w + ix0) * 2 + 1];
            const int ix1 = fx1 + 0.5f;
            const int iy1 = fy1 + 0.5f;
            if (ix1 >= 0 && ix1 < w && iy1 >= 0 && iy1 < hh) {
              const int ix1s = std::min(std::max(static_cast<int>(fx1 * scaleX + 0.5f), 0), dw - 1);
              const int iy1s = std::min(std::max(static_cast<int>(fy1 * scaleY + 0.5f), 0), dh - 1);
              const float d1 = dynDist ? dynDist[static_cast<size_t>(fb) * dw * dh + static_cast<size_t>(iy1s) * dw + ix1s] : FLT_MAX;
              if (d1 > minDynamicDistance) pixels.push_back({cornerPtr[iy0 * w + ix0], ix0, iy0, fx1, fy1});
            }
          }
        }
      }
      std::stable_sort(pixels.begin(), pixels.end(),
                       [](const Pixel& a, const Pixel& b) { return a.cornerStrength > b.cornerStrength; });
      std::vector<uint8_t> invalid(npx, 0);
      const int r = matchSeparation;
      const float sx = 1.f / w, sy = o->invAspect / hh;
      for (const Pixel& px : pixels) {
        if (invalid[static_cast<size_t>(px.iy0) * w + px.ix0]) continue;
        const float l[4] = {px.ix0 * sx, px.iy0 * sy, px.fx1 * sx, px.fy1 * sy};
        o->sampledLoc.insert(o->sampledLoc.end(), l, l + 4);
        for (int my = std::max(0, px.iy0 - r); my <= std::min(hh - 1, px.iy0 + r); ++my)
          for (int mx = std::max(0, px.ix0 - r); mx <= std::min(w - 1, px.ix0 + r); ++mx) {
            const int rx = mx - px.ix0, ry = my - px.iy0;
            if (rx * rx + ry * ry <= r * r) invalid[static_cast<size_t>(my) * w + mx] = 255;
          }
      }
      offsets[p + 1] = static_cast<int64_t>(o->sampledLoc.size() / 4);
    }
  });
}
// FlowConstraintsCollection::compute(TripletKey), reference lib/FlowConstraints.cpp:467-550, including its two index
// slips: corner response read at cornerPtr[ix0] (:533) and the third dynamic-distance test on
// dynamicDistance1 (:527).  Unchecked Mat accesses are clamped like in the pair version.
int cvdo_sample_triplet_constraints(void* h, int numTriplets, const int32_t* centers, const float* corner,
                                    const float* flow10, const uint8_t* mask10, const float* flow12,
                                    const uint8_t* mask12, const float* dynDist, int dynW, int dynH, int matchSeparation,
                                    float minDynamicDistance, int64_t* offsets) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    const int w = o->W, hh = o->Hh;
    const size_t npx = static_cast<size_t>(w) * hh;
    o->sampledTrip.clear();
    offsets[0] = 0;
    const int dw = dynDist ? dynW : w, dh = dynDist ? dynH : hh;
    const float scaleX = dw / float(w), scaleY = dh / float(hh);
    auto sc = [](float v, float s, int hi) { return std::min(std::max(static_cast<int>(v * s + 0.5f), 0), hi); };
    auto dist = [&](int f, int y, int x) {
      return dynDist ? dynDist[static_cast<size_t>(f) * dw * dh + static_cast<size_t>(y) * dw + x] : FLT_MAX;
    };
    struct Pixel { float cornerStrength; int ix1, iy1; float fx0, fy0, fx2, fy2; };
    for (int g = 0; g < numTriplets; ++g) {
      const int fc = centers[g];
      const float* cornerImg = corner + fc * npx;
      const float* f10 = flow10 + static_cast<size_t>(g) * npx * 2;
      const float* f12 = flow12 + static_cast<size_t>(g) * npx * 2;
      const uint8_t* m10 = mask10 + static_cast<size_t>(g) * npx;
      const uint8_t* m12 = mask12 + static_cast<size_t>(g) * npx;
      std::vector<Pixel> pixels;
      for (int iy1 = 0; iy1 < hh; ++iy1) {
        const float* cornerPtr = cornerImg + static_cast<size_t>(iy1) * w;
        const int iy1s = sc(iy1, scaleY, dh - 1);
        for (int ix1 = 0; ix1 < w; ++ix1) {
          const int ix1s = sc(ix1, scaleX, dw - 1);
          const size_t pi = static_cast<size_t>(iy1) * w + ix1;
          if (m10[pi] && m12[pi] && dist(fc, iy1s, ix1s) > minDynamicDistance) {
            const float fx0 = ix1 + f10[pi * 2], fy0 = iy1 + f10[pi * 2 + 1];
            const int ix0 = fx0 + 0.5f, iy0 = fy0 + 0.5f;
            const float fx2 = ix1 + f12[pi * 2], fy2 = iy1 + f12[pi * 2 + 1];
            const int ix2 = fx2 + 0.5f, iy2 = fy2 + 0.5f;
            if (ix0 >= 0 && ix0 < w && iy0 >= 0 && iy0 < hh && ix2 >= 0 && ix2 < w && iy2 >= 0 && iy2 < hh) {
              const int ix0s = sc(fx0, scaleX, dw - 1), iy0s = sc(fy0, scaleY, dh - 1);
              const int ix2s = sc(fx2, scaleX, dw - 1), iy2s = sc(fy2, scaleY, dh - 1);
              if (dist(fc - 1, iy0s, ix0s) > minDynamicDistance && dist(fc, iy2s, ix2s) > minDynamicDistance)
                pixels.push_back({cornerPtr[ix0], ix1, iy1, fx0, fy0, fx2, fy2});
            }
          }
        }
      }
      std::stable_sort(pixels.begin(), pixels.end(),
                       [](const Pixel& a, const Pixel& b) { return a.cornerStrength > b.cornerStrength; });
      std::vector<uint8_t> invalid(npx, 0);
      const int r = matchSeparation;
      const float sx = 1.f / w, sy = o->invAspect / hh;
      for (const Pixel& px : pixels) {
        if (invalid[static_cast<size_t>(px.iy1) * w + px.ix1]) continue;  // referencePixel = c[1] (:345-347)
        const float l[6] = {px.fx0 * sx, px.fy0 * sy, px.ix1 * sx, px.iy1 * sy, px.fx2 * sx, px.fy2 * sy};
        o->sampledTrip.insert(o->sampledTrip.end(), l, l + 6);
        for (int my = std::max(0, px.iy1 - r); my <= std::min(hh - 1, px.iy1 + r); ++my)
          for (int mx = std::max(0, px.ix1 - r); mx <= std::min(w - 1, px.ix1 + r); ++mx) {
            const int rx = mx - px.ix1, ry = my - px.iy1;
            if (rx * rx + ry * ry <= r * r) invalid[static_cast<size_t>(my) * w + mx] = 255;
          }
      }
      offsets[g + 1] = static_cast<int64_t>(o->sampledTrip.size() / 6);
    }
  });
}
int cvdo_get_sampled_triplet_constraints(void* h, float* loc6) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, std::memcpy(loc6, o->sampledTrip.data(), sizeof(float) * o->sampledTrip.size()));
}
int cvdo_get_sampled_constraints(void* h, float* loc4) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, std::memcpy(loc4, o->sampledLoc.data(), sizeof(float) * o->sampledLoc.size()));
}

// ---- dense consumers of the result: same signatures as include/cvd_hip.h -----------------------
// Pixel-centre convention of the reference: loc = (-1 + x * 2/(w-1), 1 - y * 2/(h-1)) in f32.
static void pixelLoc(int x, int y, int w, int h, float& lx, float& ly) {
  const float xScale = 2.f / (w - 1.f), yScale = 2.f / (h - 1.f);
  lx = -1.f + x * xScale;
  ly = 1.f - y * yScale;
}
// DepthXform::apply, reference lib/DepthMapTransform.cpp:394-415
int cvdo_apply_depth_xforms(void* h, int firstFrame, int numFrames, float* out, double* /*kernelMs*/) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    for (int k = 0; k < numFrames; ++k) {
      const int f = firstFrame + k;
      const cvdo::Xform& X = o->depthXforms[f];
      const int N = X.blockSize;
      for (int y = 0; y < o->Hh; ++y)
        for (int x = 0; x < o->W; ++x) {
          const double d = o->depthImg(f)[static_cast<size_t>(y) * o->W + x];
          float lx, ly;
          pixelLoc(x, y, o->W, o->Hh, lx, ly);
          double D = d;
          if (X.desc.depth_type != CVD_DEPTH_IDENTITY && N > 0) {
            cvdo::Gather g;
            X.depthGather(static_cast<float>(d), lx, ly, g);
            D = 0.0;
            for (int i = 0; i < g.n; ++i)
              D += ((N == 2) ? (d * X.params[g.idx[i] * 2] + X.params[g.idx[i] * 2 + 1]) : d * X.params[g.idx[i]]) * g.w[i];
          }
          out[(static_cast<size_t>(k) * o->Hh + y) * o->W + x] = static_cast<float>(D);
        }
    }
  });
}
// GridDepthXform::paramMap, reference lib/DepthMapTransform.cpp:950-994 (:422-425 for the other transform types)
int cvdo_depth_param_maps(void* h, int firstFrame, int numFrames, double* out, double* /*kernelMs*/) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    for (int k = 0; k < numFrames; ++k) {
      const int f = firstFrame + k;
      const cvdo::Xform& X = o->depthXforms[f];
      if (X.desc.depth_type != CVD_DEPTH_GRID) throw std::runtime_error("Parameter map not implemented for this transform type.");
      const int N = X.blockSize;
      for (int y = 0; y < o->Hh; ++y)
        for (int x = 0; x < o->W; ++x) {
          float lx, ly;
          pixelLoc(x, y, o->W, o->Hh, lx, ly);
          cvdo::Gather g;
          X.depthGather(o->depthImg(f)[static_cast<size_t>(y) * o->W + x], lx, ly, g);
          double* dst = out + ((static_cast<size_t>(k) * o->Hh + y) * o->W + x) * N;
          for (int d = 0; d < N; ++d) dst[d] = 0.0;
          for (int i = 0; i < g.n; ++i)
            for (int d = 0; d < N; ++d) dst[d] += X.params[g.idx[i] * N + d] * g.w[i];
        }
    }
  });
}
// SpatialXform::warp(h, w), reference lib/DepthMapTransform.cpp:428-449
int cvdo_spatial_warp_maps(void* h, int firstFrame, int numFrames, int height, int width, float* out, double* /*kernelMs*/) {
  Oracle* o = static_cast<Oracle*>(h);
  CVDO_TRY(h, {
    for (int k = 0; k < numFrames; ++k) {
      const cvdo::Xform& X = o->spatialXforms[firstFrame + k];
      for (int y = 0; y < height; ++y)
        for (int x = 0; x < width; ++x) {
          float lx, ly;
          pixelLoc(x, y, width, height, lx, ly);
          cvdo::Gather g;
          X.spatialGather(lx, ly, g);
          double wx = 0.0, wy = 0.0;
          for (int i = 0; i < g.n; ++i) { wx += X.params[g.idx[i] * 2] * g.w[i]; wy += X.params[g.idx[i] * 2 + 1] * g.w[i]; }
          float* dst = out + ((static_cast<size_t>(k) * height + y) * width + x) * 2;
          dst[0] = static_cast<float>(wx);
          dst[1] = static_cast<float>(wy);
        }
    }
  });
}

// ---- stand-alone known-answer hooks ----------------------------------------------------------------
// Depth / spatial gather of one sample: returns the number of blocks, fills idx / w (<= 16).
int cvdo_gather(const cvd_xform_desc* d, float srcDepth, float lx, float ly, int32_t* idx, double* w) {
  try {
    cvdo::Xform x = cvdo::Xform::create(*d);
    cvdo::Gather g;
    if (d->type == CVD_XFORM_DEPTH) x.depthGather(srcDepth, lx, ly, g);
    else x.spatialGather(lx, ly, g);
    for (int i = 0; i < g.n; ++i) { idx[i] = g.idx[i]; w[i] = g.w[i]; }
    return g.n;
  } catch (const std::exception&) {
    return -1;
  }
}
void cvdo_angle_axis_rotate_point(const double* aa, const double* pt, double* out) {
  cvdo::angleAxisRotatePoint(aa, pt, out);
}
void cvdo_rotation_matrix_to_angle_axis(const double* Rcm, double* aa) { cvdo::rotationMatrixToAngleAxis(Rcm, aa); }
void cvdo_angle_axis_to_rotation_matrix(const double* aa, double* Rcm) { cvdo::angleAxisToRotationMatrix(aa, Rcm); }
void cvdo_rotation_matrix_to_quaternion(const double* Rcm, double* q) { cvdo::rotationMatrixToEigenQuaternion(Rcm, q); }
// Deformation cost of one transform (double): returns number of residuals.
int cvdo_deformation_cost(const cvd_xform_desc* d, const double* params, double* residuals) {
  try {
    cvdo::Xform x = cvdo::Xform::create(*d);
    std::vector<const double*> blocks(x.numBlocks);
    for (int k = 0; k < x.numBlocks; ++k) blocks[k] = params + static_cast<size_t>(k) * x.blockSize;
    x.deformationCost(blocks.data(), residuals);
    return x.numDeformationResiduals();
  } catch (const std::exception&) {
    return -1;
  }
}

// ---- image operators in front of the sampler -------------------------------------------------
// The reference calls OpenCV here (lib/FlowConstraints.cpp:417-423 cvtColor + cornerMinEigenVal, :257-286
// distanceTransform).  OpenCV is not part of /root/reference and not installed: the functions below restate its
// published algorithms (imgproc corner.cpp: Sobel with the 1/(2^(aperture-1) blockSize) scale on the smoothing taps,
// cov = (dx^2, dx dy, dy^2), unnormalised box filter, calcMinEigenVal; distransform.cpp: two-pass 5x5 chamfer in 16-bit
// fixed point with weights 1, 1.4, 2.1969), BORDER_REFLECT_101.  Parity with an OpenCV build is unpinned.
static int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}
int cvdo_corner_min_eigenval(void* h, int numImages, int height, int width, const float* bgr, float* out,
                             double* /*kernelMs*/) {
  CVDO_TRY(h, {
    const int w = width, hh = height;
    const size_t px = static_cast<size_t>(w) * hh;
    const float k0 = static_cast<float>(1.0 / 12.0), k1 = static_cast<float>(2.0 / 12.0);
    std::vector<float> gray(px), rowDiff(px), rowSmooth(px), cov(px * 3), rowSum(px * 3);
    for (int n = 0; n < numImages; ++n) {
      const float* im = bgr + static_cast<size_t>(n) * px * 3;
      for (size_t i = 0; i < px; ++i) gray[i] = (im[i * 3] * 0.114f + im[i * 3 + 1] * 0.587f) + im[i * 3 + 2] * 0.299f;
      // separable Sobel: row pass (difference / scaled smoothing), then column pass
      for (int y = 0; y < hh; ++y)
        for (int x = 0; x < w; ++x) {
          const float l = gray[static_cast<size_t>(y) * w + reflect101(x - 1, w)];
          const float r = gray[static_cast<size_t>(y) * w + reflect101(x + 1, w)];
          const float c = gray[static_cast<size_t>(y) * w + x];
          rowDiff[static_cast<size_t>(y) * w + x] = r - l;
          rowSmooth[static_cast<size_t>(y) * w + x] = c * k1 + (l + r) * k0;
        }
      for (int y = 0; y < hh; ++y)
        for (int x = 0; x < w; ++x) {
          const size_t up = static_cast<size_t>(reflect101(y - 1, hh)) * w + x;
          const size_t dn = static_cast<size_t>(reflect101(y + 1, hh)) * w + x;
          const size_t at = static_cast<size_t>(y) * w + x;
          const float dx = rowDiff[at] * k1 + (rowDiff[up] + rowDiff[dn]) * k0;
          const float dy = rowSmooth[dn] - rowSmooth[up];
          cov[at * 3] = dx * dx;
          cov[at * 3 + 1] = dx * dy;
          cov[at * 3 + 2] = dy * dy;
        }
      // boxFilter 3x3, normalize = false: row sums, then column sums
      for (int y = 0; y < hh; ++y)
        for (int x = 0; x < w; ++x)
          for (int k = 0; k < 3; ++k) {
            const float* row = cov.data() + static_cast<size_t>(y) * w * 3;
            rowSum[(static_cast<size_t>(y) * w + x) * 3 + k] =
                (row[reflect101(x - 1, w) * 3 + k] + row[x * 3 + k]) + row[reflect101(x + 1, w) * 3 + k];
          }
      for (int y = 0; y < hh; ++y)
        for (int x = 0; x < w; ++x) {
          float sum[3];
          for (int k = 0; k < 3; ++k)
            sum[k] = (rowSum[(static_cast<size_t>(reflect101(y - 1, hh)) * w + x) * 3 + k] +
                      rowSum[(static_cast<size_t>(y) * w + x) * 3 + k]) +
                     rowSum[(static_cast<size_t>(reflect101(y + 1, hh)) * w + x) * 3 + k];
          const float a = sum[0] * 0.5f, b = sum[1], c = sum[2] * 0.5f;
          const float d = a - c;
          out[static_cast<size_t>(n) * px + static_cast<size_t>(y) * w + x] = (a + c) - std::sqrt(d * d + b * b);
        }
    }
  });
}
int cvdo_dynamic_distance(void* h, int numImages, int height, int width, const uint8_t* mask, float* out,
                          double* /*kernelMs*/) {
  CVDO_TRY(h, {
    const int w = width, hh = height;
    const unsigned int DIST_MAX = 0x7fffffffu >> 2;
    const unsigned int HV = static_cast<unsigned int>(std::lround(1.0f * 65536.0));
    const unsigned int DIAG = static_cast<unsigned int>(std::lround(static_cast<double>(1.4f) * 65536.0));
    const unsigned int LONG = static_cast<unsigned int>(std::lround(static_cast<double>(2.1969f) * 65536.0));
    const float scale = 1.f / 65536.f;
    const int step = w + 4;
    std::vector<unsigned int> temp(static_cast<size_t>(step) * (hh + 4));
    for (int n = 0; n < numImages; ++n) {
      const uint8_t* src = mask + static_cast<size_t>(n) * w * hh;
      float* dst = out + static_cast<size_t>(n) * w * hh;
      std::fill(temp.begin(), temp.end(), DIST_MAX);
      for (int i = 0; i < hh; ++i) {  // forward pass
        unsigned int* tmp = temp.data() + static_cast<size_t>(i + 2) * step + 2;
        for (int j = 0; j < w; ++j) {
          if (src[static_cast<size_t>(i) * w + j] < 127) {  // reference binarisation (:271): < 127 -> 0
            tmp[j] = 0;
          } else {
            unsigned int t0 = tmp[j - step * 2 - 1] + LONG;
            unsigned int t = tmp[j - step * 2 + 1] + LONG; if (t0 > t) t0 = t;
            t = tmp[j - step - 2] + LONG; if (t0 > t) t0 = t;
            t = tmp[j - step - 1] + DIAG; if (t0 > t) t0 = t;
            t = tmp[j - step] + HV; if (t0 > t) t0 = t;
            t = tmp[j - step + 1] + DIAG; if (t0 > t) t0 = t;
            t = tmp[j - step + 2] + LONG; if (t0 > t) t0 = t;
            t = tmp[j - 1] + HV; if (t0 > t) t0 = t;
            tmp[j] = t0;
          }
        }
      }
      for (int i = hh - 1; i >= 0; --i) {  // backward pass
        unsigned int* tmp = temp.data() + static_cast<size_t>(i + 2) * step + 2;
        for (int j = w - 1; j >= 0; --j) {
          unsigned int t0 = tmp[j];
          if (t0 > HV) {
            unsigned int t = tmp[j + step * 2 + 1] + LONG; if (t0 > t) t0 = t;
            t = tmp[j + step * 2 - 1] + LONG; if (t0 > t) t0 = t;
            t = tmp[j + step + 2] + LONG; if (t0 > t) t0 = t;
            t = tmp[j + step + 1] + DIAG; if (t0 > t) t0 = t;
            t = tmp[j + step] + HV; if (t0 > t) t0 = t;
            t = tmp[j + step - 1] + DIAG; if (t0 > t) t0 = t;
            t = tmp[j + step - 2] + LONG; if (t0 > t) t0 = t;
            t = tmp[j + 1] + HV; if (t0 > t) t0 = t;
            tmp[j] = t0;
          }
          dst[static_cast<size_t>(i) * w + j] = static_cast<float>(t0) * scale;
        }
      }
    }
  });
}

// ---- DepthVideoProcessor::flowGuidedFilter, reference lib/Processor.cpp:315-590 (+ DepthVideo::project,
// lib/DepthVideo.cpp:637-681).  Batch layout as in include/cvd_hip.h (cvd_flow_guided_filter).  float arithmetic in the
// reference's order; samples are collected in a vector and std::sort-ed for the median exactly like the reference.
static void quatRotate(const float* q, const float* v, float* out) {  // Eigen: uv = 2 q.vec x v; v + w uv + q.vec x uv
  float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; ++i) uv[i] += uv[i];
  const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; ++i) out[i] = v[i] + q[3] * uv[i] + c[i];
}
int cvdo_flow_guided_filter(void* h, int numFrames, int firstOutput, int numOutputs, int height, int width, int depthHeight,
                            int depthWidth, float invAspect, const float* depth, const float* cameras,
                            const float* flowFwd, const uint8_t* maskFwd, const float* flowBwd, const uint8_t* maskBwd,
                            int frameRadius, int spatialRadius, int median, float* out, double* /*kernelMs*/) {
  CVDO_TRY(h, {
    const int w = width, hh = height, n = numFrames;
    const size_t px = static_cast<size_t>(w) * hh;
    struct Cam { float pos[3], right[3], up[3], front[3], tanH, tanV; };
    std::vector<Cam> cams(n);
    for (int k = 0; k < n; ++k) {
      const float* c = cameras + static_cast<size_t>(k) * 9;
      const float ex[3] = {1.f, 0.f, 0.f}, ey[3] = {0.f, 1.f, 0.f}, ez[3] = {0.f, 0.f, -1.f};
      for (int i = 0; i < 3; ++i) cams[k].pos[i] = c[i];
      quatRotate(c + 3, ex, cams[k].right);
      quatRotate(c + 3, ey, cams[k].up);
      quatRotate(c + 3, ez, cams[k].front);
      cams[k].tanH = std::tan(c[7] / 2.f);   // project(): tan(intr.hFov / 2.f)
      cams[k].tanV = std::tan(c[8] / 2.f);
    }
    struct SampleInfo { float depth, weight; };
    std::vector<SampleInfo> samples;
    for (int o = 0; o < numOutputs; ++o) {
      const int frame = firstOutput + o;
      const Cam& ref = cams[frame];
      const int f0 = std::max(0, frame - frameRadius);
      const int f1 = std::min(n - 1, frame + frameRadius);
      auto addSample = [&](float lx, float ly, int fi) {  // reference :437-446 + project :656-681, :637-654
        const float nx = lx / w, ny = ly / hh * invAspect;
        int x = std::min(depthWidth - 1, int(nx * depthWidth + 0.5f));
        int y = std::min(depthHeight - 1, int(ny / invAspect * depthHeight + 0.5f));
        x = std::max(x, 0);
        y = std::max(y, 0);
        const float d = depth[(static_cast<size_t>(fi) * depthHeight + y) * depthWidth + x];
        const Cam& c = cams[fi];
        const float rx = -1.f + 2.f * nx;
        const float ry = 1.f - 2.f * ny / invAspect;
        const float a = rx * c.tanH, b = ry * c.tanV;
        float pos[3];
        for (int i = 0; i < 3; ++i) {
          const float ray = (c.front[i] + c.right[i] * a) + c.up[i] * b;
          pos[i] = c.pos[i] + ray * d;
        }
        const float dd = ((pos[0] - ref.pos[0]) * ref.front[0] + (pos[1] - ref.pos[1]) * ref.front[1]) +
                         (pos[2] - ref.pos[2]) * ref.front[2];
        samples.push_back({dd, 0.f});
      };
      for (int y = 0; y < hh; ++y) {
        const int y0 = std::max(0, y - spatialRadius), y1 = std::min(hh - 1, y + spatialRadius);
        for (int x = 0; x < w; ++x) {
          const int x0 = std::max(0, x - spatialRadius), x1 = std::min(w - 1, x + spatialRadius);
          samples.clear();
          float referenceDepth = FLT_MAX;
          for (int wy = y0; wy <= y1; ++wy)
            for (int wx = x0; wx <= x1; ++wx) {
              addSample(static_cast<float>(wx), static_cast<float>(wy), frame);
              if (wx == x && wy == y) referenceDepth = samples.back().depth;
              for (int dir = 0; dir < 2; ++dir) {
                float lx = static_cast<float>(wx), ly = static_cast<float>(wy);
                for (int fi = frame + (dir ? -1 : 1); dir ? fi >= f0 : fi <= f1; fi += dir ? -1 : 1) {
                  // forward: flow (fi-1 -> fi); backward: flow (fi+1 -> fi)
                  const size_t e = static_cast<size_t>(dir ? fi : fi - 1) * px;
                  const float* flow = (dir ? flowBwd : flowFwd) + e * 2;
                  const uint8_t* mask = (dir ? maskBwd : maskFwd) + e;
                  int ix = std::min(int(lx + 0.5f), w - 1);
                  int iy = std::min(int(ly + 0.5f), hh - 1);
                  if (!mask[static_cast<size_t>(iy) * w + ix]) break;
                  lx += flow[(static_cast<size_t>(iy) * w + ix) * 2];
                  ly += flow[(static_cast<size_t>(iy) * w + ix) * 2 + 1];
                  ix = static_cast<int>(lx + 0.5f);
                  iy = static_cast<int>(ly + 0.5f);
                  if (ix < 0 || ix >= w || iy < 0 || iy >= hh) break;
                  addSample(lx, ly, fi);
                }
              }
            }
          float depthSum = 0.f, weightSum = 0.f;
          for (SampleInfo& sm : samples) {
            const float value = std::max(sm.depth, referenceDepth) / std::min(sm.depth, referenceDepth);
            sm.weight = expf(-value * 3.f);
            depthSum += sm.depth * sm.weight;
            weightSum += sm.weight;
          }
          float result = 0.f;
          if (median) {
            const float halfWeight = weightSum / 2.f;
            std::sort(samples.begin(), samples.end(),
                      [](const SampleInfo& l, const SampleInfo& r) { return l.depth < r.depth; });
            float cum = 0.f;
            for (const SampleInfo& sm : samples) {
              cum += sm.weight;
              if (cum >= halfWeight) { result = sm.depth; break; }
            }
          } else {
            result = weightSum > 0.f ? depthSum / weightSum : 0.f;
          }
          out[(static_cast<size_t>(o) * hh + y) * w + x] = result;
        }
      }
    }
  });
}

}  // extern "C"
