// cvd_device.h -- device-side math of the geometric-consistency optimizer (gfx950 / CDNA4 only).
//
// What the reference evaluates per residual block with Ceres dual numbers
// (reference lib/PoseOptimizer.cpp:142-308, lib/DepthMapTransform.cpp:585-948,1214-1343) is evaluated
// here in closed form: residual + ANALYTIC Jacobian, so that one constraint costs
// a few hundred f64 FMAs instead of ceil(P/4) Jet passes.  Rotation data that is constant per frame
// (R(w) and dR/dw_i of ceres::AngleAxisRotatePoint's two-branch formula) is hoisted into FrameConst.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvd {

// ---------------------------------------------------------------------------------------------------
// Problem layout, passed by value to every kernel (all fields uniform -> SGPRs).
// Per-frame unknown block: [t(3) w(3) fy(1) | theta (numDepthBlocks*N) | phi (numSpatialBlocks*2)].
// ---------------------------------------------------------------------------------------------------
struct Layout {
  int F, B;
  // depth transform
  int depthType;  // cvd_depth_xform_type
  int N;          // value-transform params per vertex (0 identity, 1 Scale, 2 ScaleShift)
  int cubic;
  int gx, gy;     // grid cols / rows (1 for Global)
  double maxcx, maxcy;  // nextafter(g-1, 0)
  // depth-wise axis of the grid (gridSize.z > 1, reference lib/DepthMapTransform.cpp:709-729, 771-779): the third grid
  // coordinate is the SOURCE DISPARITY of the sample, (1 / d_src - dispMin) / dispInterval clamped to [0, maxcz]
  int gz;
  double maxcz, dispMin, dispInterval;
  int nD;         // depth params per frame
  // spatial transform
  int spatialType;  // cvd_spatial_xform_type
  int sgx, sgy;
  double smaxcx, smaxcy;
  int nS;  // spatial params per frame
  // camera / loss
  double aspect, vFocal;
  int intrOpt, lossType;
  double ws, wd;      // staticSpatialWeight, staticDepthWeight
  double cauchyB;     // robustness^2
  double cauchyC;     // 1 / robustness^2
  double robustA;     // robustness (Huber: rho = 2 a sqrt(s) - a^2 beyond s = a^2)
  int robustKind;     // kRobustCauchy (the reference's CauchyLoss, lib/PoseOptimizer.cpp:1220) | kRobustHuber
  // regularisers (already decided by the host: 0 => skipped)
  double scaleRegSqrt;   // sqrt(scaleReg) (ScaledLoss => sqrt weight on the residual)
  int sregX, sregY;      // scale-regulariser sample grid
  double focalRegSqrt;   // sqrt(focalReg)
  double depthDeformW;   // plain multiplier (no loss function)
  double spatialDeformW; // plain multiplier
  int includeStatic;     // 0 for normalizeDepth's default problem
  // position regulariser t_k - 2 t_{k+1} + t_{k+2} (reference lib/PoseOptimizer.cpp:1417-1447)
  double positionRegSqrt;  // sqrt(positionReg), 0 => off
  int firstFrame, lastFrame;
  int rank, world;         // residual k of a per-frame / per-triple regulariser belongs to rank k % world
  // AdaptiveDeformationCost (reference lib/PoseOptimizer.cpp:559-656): per-frame grid-vertex weights [F][gx * gy]
  // (dynamic fraction of the mask pixels splatted onto each vertex), nullptr => plain DeformationCost
  const double* adaptW;
  double adaptive;         // params.adaptiveDeformationCost
};

// cvd enums duplicated as plain ints to keep this header free of host headers.
enum : int { kDepthIdentity = 1, kDepthGlobal = 2, kDepthGrid = 3 };
enum : int { kSpIdentity = 1, kSpVertical = 2, kSpCorners = 3, kSpBilinear = 4, kSpBicubic = 5 };
enum : int { kLossEuclid = 0, kLossDisparity = 1, kLossRatio = 2, kLossLog = 3,
             // not a StaticLossType: DisparityDissimilarityCost of normalizeDepth's pair loop (reference
             // lib/PoseOptimizer.cpp:425-462, 1014-1105): ONE residual 1/max(D_a, eps) - 1/max(D_b, eps), depth blocks only
             kLossNormalizeDisparity = 4 };
enum : int { kIntrFixed = 0, kIntrShared = 1, kIntrPerFrame = 2 };
enum : int { kRobustCauchy = 0, kRobustHuber = 1 };

// rho(s), rho'(s) of the robust loss on a static constraint with squared norm s.  Cauchy: ceres::CauchyLoss(a), what the
// reference hard-wires (lib/PoseOptimizer.cpp:1220); Huber: ceres::HuberLoss(a), the stress variant BASELINE.json's
// configs[4] names (cvd_solver_options::robust_loss).  Both have rho'' <= 0, so Ceres' corrector reduces to the plain
// sqrt(rho') scaling of residual and Jacobian in either case.  The branch is uniform over the launch.
__device__ __forceinline__ void robustRho(const Layout& L, double sq, double& rho0, double& rho1) {
  if (L.robustKind == kRobustHuber) {
    if (sq > L.cauchyB) {
      const double r = sqrt(sq);
      rho0 = 2.0 * L.robustA * r - L.cauchyB;
      rho1 = fmax(2.2250738585072014e-308, L.robustA / r);
    } else {
      rho0 = sq;
      rho1 = 1.0;
    }
  } else {
    const double sum = 1.0 + sq * L.cauchyC;
    rho0 = L.cauchyB * log(sum);
    rho1 = 1.0 / sum;
  }
}
__device__ __forceinline__ double robustRho1(const Layout& L, double sq) {
  if (L.robustKind == kRobustHuber) return sq > L.cauchyB ? fmax(2.2250738585072014e-308, L.robustA / sqrt(sq)) : 1.0;
  return 1.0 / (1.0 + sq * L.cauchyC);
}

// Per-frame constants of one evaluation point (49 doubles).
struct FrameConst {
  double R[9];      // row-major R(w): ceres::AngleAxisRotatePoint as a matrix (I + [w]x below eps)
  double dR[3][9];  // dR/dw_i, differentiated through the SAME branch
  double t[3];
  double fy;        // vertical focal actually used (vFocal when intrinsics are fixed)
  double Jl[9];     // row i = a_i, the axial vector of dR_i R^T: dR/dw_i = [a_i]x R (columns of the left Jacobian of SO(3)); the
                    // pair-major product uses the rotation derivatives in this cross-product form (k_matvec_pairs_fast)
};

__device__ __forceinline__ void frameConstFromParams(const double* __restrict__ x, int intrOpt, double vFocal,
                                                     const double* __restrict__ x0, FrameConst& fc) {
  const double wx = x[3], wy = x[4], wz = x[5];
  fc.t[0] = x[0]; fc.t[1] = x[1]; fc.t[2] = x[2];
  fc.fy = (intrOpt == kIntrFixed) ? vFocal : (intrOpt == kIntrShared ? x0[6] : x[6]);
  const double th2 = wx * wx + wy * wy + wz * wz;
  const double w[3] = {wx, wy, wz};
  if (th2 > 2.220446049250313e-16) {
    const double th = sqrt(th2);
    const double s = sin(th), c = cos(th);
    const double ti = 1.0 / th;
    const double k[3] = {wx * ti, wy * ti, wz * ti};
    const double oc = 1.0 - c;
    // R = c I + s K + (1-c) k k^T
    fc.R[0] = c + oc * k[0] * k[0];      fc.R[1] = -s * k[2] + oc * k[0] * k[1]; fc.R[2] = s * k[1] + oc * k[0] * k[2];
    fc.R[3] = s * k[2] + oc * k[1] * k[0]; fc.R[4] = c + oc * k[1] * k[1];      fc.R[5] = -s * k[0] + oc * k[1] * k[2];
    fc.R[6] = -s * k[1] + oc * k[2] * k[0]; fc.R[7] = s * k[0] + oc * k[2] * k[1]; fc.R[8] = c + oc * k[2] * k[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      // d theta / d w_i = k_i ; d k / d w_i = (e_i - k k_i) / theta
      double dk[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) dk[j] = ((i == j ? 1.0 : 0.0) - k[j] * k[i]) * ti;
      const double ki = k[i];
      double* D = fc.dR[i];
      // -s k_i I + c k_i K + s dK + s k_i k k^T + (1-c)(dk k^T + k dk^T)
      const double a = -s * ki, b = c * ki, e = s * ki;
      const double K[9] = {0.0, -k[2], k[1], k[2], 0.0, -k[0], -k[1], k[0], 0.0};
      const double dK[9] = {0.0, -dk[2], dk[1], dk[2], 0.0, -dk[0], -dk[1], dk[0], 0.0};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          D[r * 3 + q] = (r == q ? a : 0.0) + b * K[r * 3 + q] + s * dK[r * 3 + q] + e * k[r] * k[q] +
                         oc * (dk[r] * k[q] + k[r] * dk[q]);
    }
  } else {
    fc.R[0] = 1.0;   fc.R[1] = -w[2]; fc.R[2] = w[1];
    fc.R[3] = w[2];  fc.R[4] = 1.0;   fc.R[5] = -w[0];
    fc.R[6] = -w[1]; fc.R[7] = w[0];  fc.R[8] = 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int q = 0; q < 9; ++q) fc.dR[i][q] = 0.0;
    // d/dw_0 [w]x = [[0,0,0],[0,0,-1],[0,1,0]] etc.
    fc.dR[0][5] = -1.0; fc.dR[0][7] = 1.0;
    fc.dR[1][2] = 1.0;  fc.dR[1][6] = -1.0;
    fc.dR[2][1] = -1.0; fc.dR[2][3] = 1.0;
  }
  // a_i = axial vector of M = dR_i R^T (antisymmetric up to rounding: the antisymmetric part is taken)
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double* D = fc.dR[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = (k + 2) % 3, cc = (k + 1) % 3;
      fc.Jl[i * 3 + k] = 0.5 * ((D[r * 3] * fc.R[cc * 3] + D[r * 3 + 1] * fc.R[cc * 3 + 1] + D[r * 3 + 2] * fc.R[cc * 3 + 2]) -
                                (D[cc * 3] * fc.R[r * 3] + D[cc * 3 + 1] * fc.R[r * 3 + 1] + D[cc * 3 + 2] * fc.R[r * 3 + 2]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Gathers (which vertices a sample depends on, with which weights)
// reference lib/DepthMapTransform.cpp:739-851 (linear), :853-948 (cubic, border folding),
// :1107-1114, :1181-1191, :1253-1343 (spatial)
// ---------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void gridCell(float loc, int g, double maxc, int& i, double& r) {
  double s = (static_cast<double>(loc) + 1.0) * static_cast<double>(g - 1) / 2.0;
  s = fmin(fmax(s, 0.0), maxc);  // std::clamp(s, 0, maxc) for the finite s that reach here: v_max_f64 + v_min_f64 instead of
                                 // two compares and four selects (this sits four times in the hot product's loop)
  i = static_cast<int>(s);
  r = s - static_cast<double>(i);
}

__host__ __device__ __forceinline__ void cubicTaps(double t, double w[4]) {
  const double t2 = t * t;
  const double t3 = t2 * t;
  w[0] = -0.5 * t3 + t2 - 0.5 * t;
  w[1] = 1.5 * t3 - 2.5 * t2 + 1.0;
  w[2] = -1.5 * t3 + 2.0 * t2 + 0.5 * t;
  w[3] = 0.5 * t3 - 0.5 * t2;
}

template <int K>
struct Taps {
  int n;
  int idx[K];
  double w[K];
};

__host__ __device__ __forceinline__ void bilinearTaps(float lx, float ly, int gx, int gy, double mx, double my, int* idx,
                                             double* w) {
  int ix, iy;
  double rx, ry;
  gridCell(lx, gx, mx, ix, rx);
  gridCell(ly, gy, my, iy, ry);
  const int i0 = ix + iy * gx;
  idx[0] = i0;          w[0] = (1.0 - rx) * (1.0 - ry);
  idx[1] = i0 + 1;      w[1] = rx * (1.0 - ry);
  idx[2] = i0 + gx;     w[2] = (1.0 - rx) * ry;
  idx[3] = i0 + gx + 1; w[3] = rx * ry;
}

// 2-D Catmull-Rom gather with out-of-range taps folded onto the clamped neighbour, in separable form: the
// tap (x, y), x < xs, y < ys, reads control point base + x + y * gx with weight fx[x] * fy[y].
struct CubicSep {
  int base, xs, ys;
  double fx[4], fy[4];
};

__host__ __device__ __forceinline__ void bicubicSeparable(float lx, float ly, int gx, int gy, double mx, double my,
                                                          CubicSep& s) {
  int ix, iy;
  double rx, ry;
  gridCell(lx, gx, mx, ix, rx);
  gridCell(ly, gy, my, iy, ry);
  double wx[4], wy[4];
  cubicTaps(rx, wx);
  cubicTaps(ry, wy);
  const int x0 = (ix == 0 ? 1 : 0);
  const int x1 = (ix == gx - 2 ? 3 : 4);
  const int y0 = (iy == 0 ? 1 : 0);
  const int y1 = (iy == gy - 2 ? 3 : 4);
  s.xs = x1 - x0;
  s.ys = y1 - y0;
  s.base = (ix - 1 + x0) + (iy - 1 + y0) * gx;
  // fold the 1-D weights first (the 2-D weights are separable products of the folded 1-D weights)
#pragma unroll
  for (int q = 0; q < 4; ++q) { s.fx[q] = 0.0; s.fy[q] = 0.0; }
#pragma unroll
  for (int x = 0; x < 4; ++x) {
    int cx = x - x0; cx = cx < 0 ? 0 : (cx > s.xs - 1 ? s.xs - 1 : cx);
    int cy = x - y0; cy = cy < 0 ? 0 : (cy > s.ys - 1 ? s.ys - 1 : cy);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q == cx) s.fx[q] += wx[x];
      if (q == cy) s.fy[q] += wy[x];
    }
  }
}

__host__ __device__ __forceinline__ int bicubicTaps(float lx, float ly, int gx, int gy, double mx, double my, int* idx,
                                           double* w) {
  CubicSep s;
  bicubicSeparable(lx, ly, gx, gy, mx, my, s);
  int n = 0;
#pragma unroll
  for (int y = 0; y < 4; ++y) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      if (y < s.ys && x < s.xs) {
        idx[n] = s.base + x + y * gx;
        w[n] = s.fx[x] * s.fy[y];
        ++n;
      }
    }
  }
  return n;
}

// Number of deformation edges of the depth grid (x-, y- and z-neighbours; reference lib/DepthMapTransform.cpp:996-1002)
__host__ __device__ __forceinline__ int gridNumEdges(int gx, int gy, int gz) {
  return (gx - 1) * gy * gz + gx * (gy - 1) * gz + gx * gy * (gz - 1);
}

// GridDepthXform::linearGather with a depth-wise axis (reference lib/DepthMapTransform.cpp:739-851): 8 taps (spatial and
// depth-wise), or 2 (depth-wise only: gridSize.x == gridSize.y == 1).  srcDepth is the f32 source depth of the sample.
template <int K>
__device__ __forceinline__ void depthwiseTaps(const Layout& L, float lx, float ly, float srcDepth, Taps<K>& t) {
  const double srcDisparity = 1.0 / static_cast<double>(srcDepth);
  double sz = (srcDisparity - L.dispMin) / L.dispInterval;
  sz = sz < 0.0 ? 0.0 : (sz > L.maxcz ? L.maxcz : sz);  // std::clamp
  const int iz = static_cast<int>(sz);
  const double rz = sz - static_cast<double>(iz);
  if (L.gx > 1) {
    if constexpr (K >= 8) {
      int idx4[4];
      double w4[4];
      bilinearTaps(lx, ly, L.gx, L.gy, L.maxcx, L.maxcy, idx4, w4);
      const int zs = L.gx * L.gy;
      t.n = 8;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        t.idx[k] = idx4[k] + iz * zs;
        t.w[k] = w4[k] * (1.0 - rz);
        t.idx[4 + k] = idx4[k] + (iz + 1) * zs;
        t.w[4 + k] = w4[k] * rz;
      }
    } else {
      t.n = 0;
    }
  } else {
    if constexpr (K >= 2) {
      t.n = 2;
      t.idx[0] = iz;     t.w[0] = 1.0 - rz;
      t.idx[1] = iz + 1; t.w[1] = rz;
    } else {
      t.n = 0;
    }
  }
}

// srcDepth: the sample's source depth -- only the depth-wise grids look at it.
template <int KD>
__device__ __forceinline__ void depthGather(const Layout& L, float lx, float ly, float srcDepth, Taps<KD>& t) {
  if (L.depthType == kDepthGlobal) {
    t.n = 1;
    t.idx[0] = 0;
    t.w[0] = 1.0;
  } else if (L.depthType == kDepthGrid && L.gz > 1) {
    depthwiseTaps<KD>(L, lx, ly, srcDepth, t);
  } else if (L.depthType == kDepthGrid) {
    if constexpr (KD >= 16) {
      if (L.cubic) {
        t.n = bicubicTaps(lx, ly, L.gx, L.gy, L.maxcx, L.maxcy, t.idx, t.w);
        return;
      }
    }
    if constexpr (KD >= 4) {
      t.n = 4;
      bilinearTaps(lx, ly, L.gx, L.gy, L.maxcx, L.maxcy, t.idx, t.w);
    } else {
      t.n = 0;
    }
  } else {
    t.n = 0;
  }
}

template <int KS>
__device__ __forceinline__ void spatialGather(const Layout& L, float lx, float ly, Taps<KS>& t) {
  if constexpr (KS == 0) {
    t.n = 0;
    return;
  } else {
    switch (L.spatialType) {
      case kSpVertical: {
        const double w0 = 0.5 + 0.5 * static_cast<double>(ly);
        t.n = 2;
        t.idx[0] = 0; t.w[0] = w0;
        t.idx[1] = 1; t.w[1] = 1.0 - w0;
        break;
      }
      case kSpCorners: {
        const double wx = 0.5 + 0.5 * static_cast<double>(lx);
        const double wy = 0.5 + 0.5 * static_cast<double>(ly);
        t.n = 4;
        t.idx[0] = 0; t.w[0] = wx * wy;
        t.idx[1] = 1; t.w[1] = (1.0 - wx) * wy;
        t.idx[2] = 2; t.w[2] = wx * (1.0 - wy);
        t.idx[3] = 3; t.w[3] = (1.0 - wx) * (1.0 - wy);
        break;
      }
      case kSpBilinear:
        t.n = 4;
        bilinearTaps(lx, ly, L.sgx, L.sgy, L.smaxcx, L.smaxcy, t.idx, t.w);
        break;
      case kSpBicubic:
        if constexpr (KS >= 16) {
          t.n = bicubicTaps(lx, ly, L.sgx, L.sgy, L.smaxcx, L.smaxcy, t.idx, t.w);
        } else {
          t.n = 0;
        }
        break;
      default:
        t.n = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// One static constraint: residual, robust weight, and the compact Jacobian of both sides.
// Columns of one side: 7 pose-like (t, w, fy) | depth taps (value params) | spatial taps (2 each).
//   d r / d theta_k[0] = JD * w_k * d_src ;  d r / d theta_k[1] = JD * w_k ;  d r / d phi_k[c] = JP[.][c] * u_k
// ---------------------------------------------------------------------------------------------------
template <int KD, int KS>
struct Side {
  double Jp[3][7];
  double JD[3];
  double JP[3][2];
  Taps<KD> dt;
  Taps<KS> st;
  double d;  // source depth (double of the float fetched with truncation)
};

template <int KD, int KS>
struct Sample {
  double r[3];
  double rho0, rho1;  // rho(s), rho'(s) of CauchyLoss
  Side<KD, KS> a, b;
};

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

template <int KD, int KS>
__device__ __forceinline__ double sideDepth(const Layout& L, const Side<KD, KS>& s, const double* __restrict__ xf) {
  if (L.depthType == kDepthIdentity) return s.d;
  double D = 0.0;
  const double* th = xf + 7;
  for (int k = 0; k < s.dt.n; ++k) {
    const double v = (L.N == 2) ? (s.d * th[s.dt.idx[k] * 2] + th[s.dt.idx[k] * 2 + 1]) : (s.d * th[s.dt.idx[k]]);
    D += v * s.dt.w[k];
  }
  return D;
}

// xa / xb: the two frames' parameter blocks (LDS or global). Returns false when the sample is skipped.
template <int KD, int KS, bool WANT_JAC>
__device__ __forceinline__ void evalSample(const Layout& L, const FrameConst& fa, const FrameConst& fb,
                                           const double* __restrict__ xa, const double* __restrict__ xb,
                                           const float4 ndc, const float2 dsrc, Sample<KD, KS>& s) {
  constexpr double eps = 1e-6;
  s.a.d = static_cast<double>(dsrc.x);
  s.b.d = static_cast<double>(dsrc.y);
  depthGather<KD>(L, ndc.x, ndc.y, dsrc.x, s.a.dt);
  depthGather<KD>(L, ndc.z, ndc.w, dsrc.y, s.b.dt);
  spatialGather<KS>(L, ndc.x, ndc.y, s.a.st);
  spatialGather<KS>(L, ndc.z, ndc.w, s.b.st);

  const double Da = sideDepth(L, s.a, xa);
  const double Db = sideDepth(L, s.b, xb);
  double pa[2] = {static_cast<double>(ndc.x), static_cast<double>(ndc.y)};
  double pb[2] = {static_cast<double>(ndc.z), static_cast<double>(ndc.w)};
  if constexpr (KS > 0) {
    const double* pha = xa + 7 + L.nD;
    const double* phb = xb + 7 + L.nD;
    for (int k = 0; k < s.a.st.n; ++k) {
      pa[0] += pha[s.a.st.idx[k] * 2] * s.a.st.w[k];
      pa[1] += pha[s.a.st.idx[k] * 2 + 1] * s.a.st.w[k];
    }
    for (int k = 0; k < s.b.st.n; ++k) {
      pb[0] += phb[s.b.st.idx[k] * 2] * s.b.st.w[k];
      pb[1] += phb[s.b.st.idx[k] * 2 + 1] * s.b.st.w[k];
    }
  }
  if (L.lossType == kLossNormalizeDisparity) {
    // DisparityDissimilarityCost: the poses and the spatial transforms take no part (Jet max(f, g): ties keep f)
    const bool ao = !(Da < eps), bo = !(Db < eps);
    const double aa = ao ? Da : eps, bb = bo ? Db : eps;
    s.r[0] = 1.0 / aa - 1.0 / bb;
    s.r[1] = 0.0;
    s.r[2] = 0.0;
    robustRho(L, s.r[0] * s.r[0], s.rho0, s.rho1);
    if constexpr (WANT_JAC) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 7; ++c) { s.a.Jp[r][c] = 0.0; s.b.Jp[r][c] = 0.0; }
        s.a.JD[r] = 0.0; s.b.JD[r] = 0.0;
        s.a.JP[r][0] = 0.0; s.a.JP[r][1] = 0.0; s.b.JP[r][0] = 0.0; s.b.JP[r][1] = 0.0;
      }
      s.a.JD[0] = ao ? -1.0 / (aa * aa) : 0.0;
      s.b.JD[0] = bo ? 1.0 / (bb * bb) : 0.0;
    }
    return;
  }
  const double A = L.aspect;
  const double fya = fa.fy, fxa = fa.fy * A;
  const double fyb = fb.fy, fxb = fb.fy * A;

  // X = t_a + D_a R_a c_a
  const double ca[3] = {pa[0] * fxa, pa[1] * fya, -1.0};
  const double Rca[3] = {dot3(fa.R, ca), dot3(fa.R + 3, ca), dot3(fa.R + 6, ca)};
  const double X[3] = {fa.t[0] + Rca[0] * Da, fa.t[1] + Rca[1] * Da, fa.t[2] + Rca[2] * Da};

  double G[3][3];   // d r / d X
  double M[3][3];   // d r / d q (reprojection variants)
  double v[3] = {0, 0, 0};
  double u = 0, vv = 0, z = 1;
  double dr2dDb = 0.0;
  double Rcb[3] = {0, 0, 0}, cb[3] = {0, 0, 0};

  if (L.lossType == kLossEuclid) {
    cb[0] = pb[0] * fxb; cb[1] = pb[1] * fyb; cb[2] = -1.0;
    Rcb[0] = dot3(fb.R, cb); Rcb[1] = dot3(fb.R + 3, cb); Rcb[2] = dot3(fb.R + 6, cb);
#pragma unroll
    for (int i = 0; i < 3; ++i) s.r[i] = (fb.t[i] + Rcb[i] * Db) - X[i];
  } else {
    v[0] = X[0] - fb.t[0]; v[1] = X[1] - fb.t[1]; v[2] = X[2] - fb.t[2];
    // q = R_b^T v  (AngleAxisRotatePoint with the negated angle-axis)
    const double q[3] = {fb.R[0] * v[0] + fb.R[3] * v[1] + fb.R[6] * v[2],
                         fb.R[1] * v[0] + fb.R[4] * v[1] + fb.R[7] * v[2],
                         fb.R[2] * v[0] + fb.R[5] * v[1] + fb.R[8] * v[2]};
    z = -q[2];
    u = q[0] / z / fxb;
    vv = q[1] / z / fyb;
    s.r[0] = (u - pb[0]) * L.ws;
    s.r[1] = (vv - pb[1]) * L.ws;
    double dr2dA = 0.0;  // d r2 / d z
    if (L.lossType == kLossDisparity) {
      const bool zo = !(z < eps);  // max(z, eps): ties keep z
      const bool bo = !(Db < eps);
      const double zz = zo ? z : eps;
      const double bb = bo ? Db : eps;
      s.r[2] = (1.0 / zz - 1.0 / bb) * L.wd;
      dr2dA = zo ? (-L.wd / (zz * zz)) : 0.0;
      dr2dDb = bo ? (L.wd / (bb * bb)) : 0.0;
    } else {
      // maxDepth = max(z, Db) (ties -> z), minDepth = min(z, Db) (ties -> z)
      const bool zIsMax = !(z < Db);
      const bool zIsMin = !(Db < z);
      const double mx = zIsMax ? z : Db;
      const double mn = zIsMin ? z : Db;
      if (L.lossType == kLossRatio) {
        s.r[2] = (mx / mn - 1.0) * L.wd;
        const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
        dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
        dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
      } else {
        s.r[2] = log(mn / mx) * L.wd;
        const double dmn = 1.0 / mn, dmx = -1.0 / mx;
        dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
        dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
      }
    }
    if constexpr (WANT_JAC) {
      const double iz = 1.0 / z;
      M[0][0] = L.ws * iz / fxb; M[0][1] = 0.0;             M[0][2] = L.ws * u * iz;
      M[1][0] = 0.0;             M[1][1] = L.ws * iz / fyb; M[1][2] = L.ws * vv * iz;
      M[2][0] = 0.0;             M[2][1] = 0.0;             M[2][2] = -dr2dA;  // z = -q.z
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          G[r][i] = M[r][0] * fb.R[i * 3 + 0] + M[r][1] * fb.R[i * 3 + 1] + M[r][2] * fb.R[i * 3 + 2];
    }
  }

  const double sq = s.r[0] * s.r[0] + s.r[1] * s.r[1] + s.r[2] * s.r[2];
  robustRho(L, sq, s.rho0, s.rho1);

  if constexpr (WANT_JAC) {
    // world-point derivatives of side a
    double dXdw[3][3];  // [i][row]
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      dXdw[i][0] = Da * dot3(fa.dR[i], ca);
      dXdw[i][1] = Da * dot3(fa.dR[i] + 3, ca);
      dXdw[i][2] = Da * dot3(fa.dR[i] + 6, ca);
    }
    const double cf[3] = {pa[0] * A, pa[1], 0.0};
    const double dXdf[3] = {Da * dot3(fa.R, cf), Da * dot3(fa.R + 3, cf), Da * dot3(fa.R + 6, cf)};
    const double dXdpx[3] = {Da * fxa * fa.R[0], Da * fxa * fa.R[3], Da * fxa * fa.R[6]};
    const double dXdpy[3] = {Da * fya * fa.R[1], Da * fya * fa.R[4], Da * fya * fa.R[7]};

    if (L.lossType == kLossEuclid) {
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s.a.Jp[r][c] = (r == c) ? -1.0 : 0.0;
          s.b.Jp[r][c] = (r == c) ? 1.0 : 0.0;
          s.a.Jp[r][3 + c] = -dXdw[c][r];
        }
        s.a.Jp[r][6] = -dXdf[r];
        s.a.JD[r] = -Rca[r];
        s.a.JP[r][0] = -dXdpx[r];
        s.a.JP[r][1] = -dXdpy[r];
        const double* Rr = fb.R + 3 * r;
#pragma unroll
        for (int c = 0; c < 3; ++c) s.b.Jp[r][3 + c] = Db * dot3(fb.dR[c] + 3 * r, cb);
        const double cfb[3] = {pb[0] * A, pb[1], 0.0};
        s.b.Jp[r][6] = Db * dot3(Rr, cfb);
        s.b.JD[r] = Rcb[r];
        s.b.JP[r][0] = Db * fxb * Rr[0];
        s.b.JP[r][1] = Db * fyb * Rr[1];
      }
    } else {
      // d q / d w_b,i = dR_b,i^T v
      double dqdw[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* D = fb.dR[i];
        dqdw[i][0] = D[0] * v[0] + D[3] * v[1] + D[6] * v[2];
        dqdw[i][1] = D[1] * v[0] + D[4] * v[1] + D[7] * v[2];
        dqdw[i][2] = D[2] * v[0] + D[5] * v[1] + D[8] * v[2];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s.a.Jp[r][c] = G[r][c];
          s.b.Jp[r][c] = -G[r][c];
          s.a.Jp[r][3 + c] = dot3(G[r], dXdw[c]);
          s.b.Jp[r][3 + c] = dot3(M[r], dqdw[c]);
        }
        s.a.Jp[r][6] = dot3(G[r], dXdf);
        s.a.JD[r] = dot3(G[r], Rca);
        s.a.JP[r][0] = dot3(G[r], dXdpx);
        s.a.JP[r][1] = dot3(G[r], dXdpy);
      }
      s.b.Jp[0][6] = -L.ws * u / fyb;
      s.b.Jp[1][6] = -L.ws * vv / fyb;
      s.b.Jp[2][6] = 0.0;
      s.b.JD[0] = 0.0; s.b.JD[1] = 0.0; s.b.JD[2] = dr2dDb;
      s.b.JP[0][0] = -L.ws; s.b.JP[0][1] = 0.0;
      s.b.JP[1][0] = 0.0;   s.b.JP[1][1] = -L.ws;
      s.b.JP[2][0] = 0.0;   s.b.JP[2][1] = 0.0;
    }
  }
}

// Number of tap columns of one side and accessors: column id inside the frame block + d r/d column.
template <int KD, int KS>
__device__ __forceinline__ int sideNumTapCols(const Layout& L, const Side<KD, KS>& s) {
  return s.dt.n * L.N + s.st.n * 2;
}
template <int KD, int KS>
__device__ __forceinline__ void sideTapCol(const Layout& L, const Side<KD, KS>& s, int t, int& col, double J[3]) {
  const int nd = s.dt.n * L.N;
  if (t < nd) {
    const int k = (L.N == 2) ? (t >> 1) : t;
    const int n = (L.N == 2) ? (t & 1) : 0;
    col = 7 + s.dt.idx[k] * L.N + n;
    const double m = s.dt.w[k] * (n == 0 ? s.d : 1.0);
    J[0] = s.JD[0] * m; J[1] = s.JD[1] * m; J[2] = s.JD[2] * m;
  } else {
    const int tt = t - nd;
    const int k = tt >> 1, c = tt & 1;
    col = 7 + L.nD + s.st.idx[k] * 2 + c;
    const double m = s.st.w[k];
    J[0] = s.JP[0][c] * m; J[1] = s.JP[1][c] * m; J[2] = s.JP[2][c] * m;
  }
}

// (J_side p)(r) for r = 0..2 with p the frame's (masked) direction block.
template <int KD, int KS>
__device__ __forceinline__ void sideJp(const Layout& L, const Side<KD, KS>& s, const double* __restrict__ p, double out[3]) {
  double sD = 0.0, sP0 = 0.0, sP1 = 0.0;
  const double* th = p + 7;
  for (int k = 0; k < s.dt.n; ++k) {
    const double v = (L.N == 2) ? (s.d * th[s.dt.idx[k] * 2] + th[s.dt.idx[k] * 2 + 1])
                                : (L.N == 1 ? s.d * th[s.dt.idx[k]] : 0.0);
    sD += v * s.dt.w[k];
  }
  if constexpr (KS > 0) {
    const double* ph = p + 7 + L.nD;
    for (int k = 0; k < s.st.n; ++k) {
      sP0 += ph[s.st.idx[k] * 2] * s.st.w[k];
      sP1 += ph[s.st.idx[k] * 2 + 1] * s.st.w[k];
    }
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    double a = s.JD[r] * sD + s.JP[r][0] * sP0 + s.JP[r][1] * sP1;
#pragma unroll
    for (int c = 0; c < 7; ++c) a += s.Jp[r][c] * p[c];
    out[r] += a;
  }
}

// ---------------------------------------------------------------------------------------------------
// Per-frame regularisers (reference lib/PoseOptimizer.cpp:1341-1415 scale, :1524-1549 focal,
// :1449-1522 deformation with lib/DepthMapTransform.cpp:631-667 / :60-70).
// visit(r, n, cols[], jac[]) is called for every regulariser residual of the frame, residual index
// `i` strided over the calling threads.  Weights are folded in (sqrt for ScaledLoss, plain for deform).
// ---------------------------------------------------------------------------------------------------
template <int KD>
__device__ __forceinline__ int numRegResiduals(const Layout& L) {
  int n = 0;
  if (L.scaleRegSqrt > 0.0) n += L.sregX * L.sregY;
  if (L.focalRegSqrt > 0.0) n += 1;
  if (L.depthDeformW > 0.0 && L.depthType == kDepthGrid) n += gridNumEdges(L.gx, L.gy, L.gz) * L.N;
  if (L.spatialDeformW > 0.0) n += L.nS;
  return n;
}

// Evaluates regulariser residual `i` of a frame. cols/jac have room for 2*KD (>= 2) entries.
template <int KD>
__device__ __forceinline__ void regResidual(const Layout& L, int f, int i, const double* __restrict__ xf, float median,
                                            double& r, int& n, int* cols, double* jac) {
  constexpr double eps = 1e-6;
  n = 0;
  r = 0.0;
  if (L.scaleRegSqrt > 0.0) {
    const int ns = L.sregX * L.sregY;
    if (i < ns) {
      const int y = i / L.sregX, x = i - y * L.sregX;
      // float arithmetic of reference lib/PoseOptimizer.cpp:1384-1385
      const float lx = __fadd_rn(-1.f, __fdiv_rn(__fmul_rn(2.f, static_cast<float>(x)), static_cast<float>(L.sregX - 1)));
      const float ly = __fadd_rn(-1.f, __fdiv_rn(__fmul_rn(2.f, static_cast<float>(y)), static_cast<float>(L.sregY - 1)));
      Taps<KD> t;
      depthGather<KD>(L, lx, ly, median, t);
      const double d = static_cast<double>(median);
      double D = (L.depthType == kDepthIdentity) ? d : 0.0;
      const double* th = xf + 7;
      for (int k = 0; k < t.n; ++k) {
        const double v = (L.N == 2) ? (d * th[t.idx[k] * 2] + th[t.idx[k] * 2 + 1]) : (d * th[t.idx[k]]);
        D += v * t.w[k];
      }
      const bool o = !(D < eps);
      const double dd = o ? D : eps;
      r = L.scaleRegSqrt * (1.0 / dd - 1.0);
      const double drdD = o ? (-L.scaleRegSqrt / (dd * dd)) : 0.0;
      for (int k = 0; k < t.n; ++k) {
        if (L.N == 2) {
          cols[n] = 7 + t.idx[k] * 2;     jac[n++] = drdD * t.w[k] * d;
          cols[n] = 7 + t.idx[k] * 2 + 1; jac[n++] = drdD * t.w[k];
        } else {
          cols[n] = 7 + t.idx[k]; jac[n++] = drdD * t.w[k] * d;
        }
      }
      return;
    }
    i -= ns;
  }
  if (L.focalRegSqrt > 0.0) {
    if (i == 0) {
      r = L.focalRegSqrt * (xf[6] - L.vFocal);
      n = 1;
      cols[0] = 6;
      jac[0] = L.focalRegSqrt;
      return;
    }
    i -= 1;
  }
  if (L.depthDeformW > 0.0 && L.depthType == kDepthGrid) {
    const int nEdges = gridNumEdges(L.gx, L.gy, L.gz) * L.N;
    if (i < nEdges) {
      // enumerate edges in any order (a sum): the x-edges, then the y-edges, then the z-edges of every vertex layer
      const int dim = i % L.N;
      int e = i / L.N;
      int va, vb;
      const int zs = L.gx * L.gy;
      const int nxE = (L.gx - 1) * L.gy * L.gz, nyE = L.gx * (L.gy - 1) * L.gz;
      if (e < nxE) {
        const int perLayer = (L.gx - 1) * L.gy;
        const int z = e / perLayer;
        e -= z * perLayer;
        const int y = e / (L.gx - 1), x = e - y * (L.gx - 1) + 1;
        va = x + y * L.gx + z * zs;
        vb = va - 1;
      } else if (e < nxE + nyE) {
        e -= nxE;
        const int perLayer = L.gx * (L.gy - 1);
        const int z = e / perLayer;
        e -= z * perLayer;
        const int y = e / L.gx + 1, x = e - (y - 1) * L.gx;
        va = x + y * L.gx + z * zs;
        vb = va - L.gx;
      } else {
        e -= nxE + nyE;  // vertex (x, y, z), z >= 1, with its neighbour one layer below
        va = e + zs;
        vb = e;
      }
      const double a = xf[7 + va * L.N + dim];
      const double b = xf[7 + vb * L.N + dim];
      const double aa = fabs(a), ab = fabs(b);
      const bool pickB = ab < aa;  // min(|a|, |b|): ties keep |a|
      const double sc = pickB ? ab : aa;
      const double diff = a - b;
      double wgt = L.depthDeformW;
      if (L.adaptW != nullptr) {
        // AdaptiveDeformationCost::operator(), reference lib/PoseOptimizer.cpp:622-645, literally: residual number R
        // of computeGridDeformationCost's enumeration (vertices row-major; per vertex the x-edge, then the y-edge;
        // N residuals per edge) is multiplied by base + max(w) * adaptive of EDGE number R while R < #edges, and is
        // left unscaled beyond (the reference advances its index once per edge, not once per residual).  With one
        // value parameter (Scale) R is the residual's own edge.
        const bool xEdge = (vb == va - 1);
        const int vy = va / L.gx, vx = va - vy * L.gx;
        const int rowStart = vy == 0 ? 0 : (L.gx - 1) + (vy - 1) * (2 * L.gx - 1);
        const int ord = vy == 0 ? vx - 1 : rowStart + (xEdge ? 2 * vx - 1 : (vx == 0 ? 0 : 2 * vx));
        const int R = ord * L.N + dim;
        const int numEdges = (L.gx - 1) * L.gy + L.gx * (L.gy - 1);
        if (R < numEdges) {
          int ey, ex;
          bool ex_is_x;
          if (R < L.gx - 1) {
            ey = 0; ex = R + 1; ex_is_x = true;
          } else {
            const int q0 = R - (L.gx - 1);
            ey = 1 + q0 / (2 * L.gx - 1);
            const int q = q0 - (ey - 1) * (2 * L.gx - 1);
            if (q == 0) { ex = 0; ex_is_x = false; }
            else { ex = (q + 1) >> 1; ex_is_x = (q & 1) != 0; }
          }
          const double* vw = L.adaptW + static_cast<size_t>(f) * L.gx * L.gy;
          const double w0 = vw[ey * L.gx + ex];
          const double w1 = ex_is_x ? vw[ey * L.gx + ex - 1] : vw[(ey - 1) * L.gx + ex];
          wgt = L.depthDeformW + (w0 < w1 ? w1 : w0) * L.adaptive;  // std::max(w0, w1)
        } else {
          wgt = 1.0;
        }
      }
      r = wgt * diff / sc;
      const double dsda = pickB ? 0.0 : (a < 0.0 ? -1.0 : 1.0);
      const double dsdb = pickB ? (b < 0.0 ? -1.0 : 1.0) : 0.0;
      n = 2;
      cols[0] = 7 + va * L.N + dim;
      jac[0] = wgt * (1.0 / sc - diff / (sc * sc) * dsda);
      cols[1] = 7 + vb * L.N + dim;
      jac[1] = wgt * (-1.0 / sc - diff / (sc * sc) * dsdb);
      return;
    }
    i -= nEdges;
  }
  if (L.spatialDeformW > 0.0 && i < L.nS) {
    r = L.spatialDeformW * xf[7 + L.nD + i];
    n = 1;
    cols[0] = 7 + L.nD + i;
    jac[0] = L.spatialDeformW;
  }
}

// Position regulariser (reference lib/PoseOptimizer.cpp:464-483, 1417-1447): residual k couples the translations
// of frames k, k+1, k+2 with coefficients (1, -2, 1) * sqrt(positionReg); it exists for
// firstFrame <= k < lastFrame - 1 when the three frames are in range (loop bound of :1420-1426).
__device__ __forceinline__ bool posRegValid(const Layout& L, const unsigned char* __restrict__ inRange, int k) {
  return L.positionRegSqrt > 0.0 && k >= L.firstFrame && k < L.lastFrame - 1 && k >= 0 && k + 2 < L.F &&
         inRange[k] && inRange[k + 1] && inRange[k + 2] && (k % L.world) == L.rank;
}
// Contribution of every position residual that touches frame g to (cost of residual g, gradient / diagonal /
// product rows of frame g's translation).  v = per-frame vectors with stride B (x for residuals, p for products).
// out3 += w * c_o * (v_k - 2 v_{k+1} + v_{k+2}),  diag += w * c_o^2
__device__ __forceinline__ void posRegFrame(const Layout& L, const unsigned char* __restrict__ inRange, int g,
                                            const double* __restrict__ v, const double* __restrict__ vmask,
                                            double out3[3], double& diag, double& costOwn) {
  const double w = L.positionRegSqrt * L.positionRegSqrt;
  const double cf[3] = {1.0, -2.0, 1.0};
  for (int o = 0; o < 3; ++o) {
    const int k = g - o;
    if (k < 0 || !posRegValid(L, inRange, k)) continue;
    double r[3];
    for (int i = 0; i < 3; ++i) {
      double a = 0.0;
      for (int j = 0; j < 3; ++j) {
        const size_t idx = static_cast<size_t>(k + j) * L.B + i;
        a += cf[j] * v[idx] * (vmask ? vmask[idx] : 1.0);
      }
      r[i] = a;
      out3[i] += w * cf[o] * a;
    }
    diag += w * cf[o] * cf[o];
    if (o == 0) costOwn += w * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  }
}

// Coarse level of the two-level preconditioner (cvd_coarse.h): kCB modes per frame, the 7 pose-like unknowns and
// one "every depth-scale vertex moves together" mode.  coarseAt = (Z c)_i for unknown i of frame f.
constexpr int kCB = 8;
__device__ __forceinline__ double coarseAt(const double* __restrict__ cF, const Layout& L, int f, int i) {
  if (cF == nullptr) return 0.0;
  if (i < 7) return cF[f * kCB + i];
  if (L.N >= 1 && L.depthType != kDepthIdentity && i < 7 + L.nD && (L.N == 1 || ((i - 7) % L.N) == 0)) return cF[f * kCB + 7];
  return 0.0;
}
// Coarse-level pieces fused into the per-frame PCG kernels (Wb == nullptr: off).  With y = W Z^T r maintained by
// recursion, y <- y - alpha W (Z^T q), k_cg_update updates the frame's row of y itself (W rows gather from the
// restricted product qc = Z^T q that k_matvec_finish left behind), so the steady-state iteration needs no separate
// "y = W Z^T r" launch.
struct CoarseStep {
  const int* wtPtr;     // W blocks of ROW i: [wtPtr[i], wtPtr[i+1])
  const int* wtBlk;     //   W block id
  const int* wtFrame;   //   frame of the block's column
  const double* Wb;
  const double* qc;     // Z^T q, [F][kCB]
  double* y;            // [F][kCB] by elimination position
  double* fdotY;        // [F] |y_i|^2
  const int* fail;
  const double* wq;     // [nW][kCB] products W_block (Z^T q)_column in ROW order (slot = position in the row lists):
                        // written column by column by coarseColumnProducts in the kernel that completes q
};
// Dense coarse level fused into k_cg_update (Ainv == nullptr: off).  c = A_c^-1 Z^T r is linear in r, so with
// r <- r - alpha q it obeys c <- c - alpha A_c^-1 (Z^T q): the product needs only qc = Z^T q, which the kernel that
// completed q left behind, i.e. nothing k_cg_update's own frame workgroups produce -- F extra workgroups of the same
// launch do it (8 rows of the inverse each) and the separate k_coarse_dense_apply launch disappears from the
// iteration.  Z^T r is carried the same way (rc <- rc - alpha qc); both start from directly computed values at the
// first residual of every PCG solve.
struct DenseStep {
  const double* Ainv;   // [8F][8F] f64 (f32 storage loses positive definiteness once cond(A_c) passes ~1e7)
  const double* qc;     // Z^T q, [F][kCB]
  double* rc;           // Z^T r, [F][kCB]
  double* c;            // A_c^-1 Z^T r, [F][kCB]
  double* dotPart;      // [F] this frame's share of r^T Z A_c^-1 Z^T r
  const unsigned char* modeActive;
  const int* fail;
  // Rows [0, rowSplit) of a frame's 8 belong to the dense-level workgroups (two frames each), rows [rowSplit, 8) to the frame's
  // OWN workgroup, which walks them after its update: the dense-level workgroups alone were the kernel's long pole (12 us for
  // 46 MB through 150 workgroups while the 300 frame workgroups had finished after 6).  rowSplit = 8: no split.
  int rowSplit;
  int ldsPsum;          // LDS offset (doubles) of the frame workgroups' partial sums for their rows
  double* dotPart2;     // [F] the frame workgroup's rows' share
};
// Column half of y <- y - alpha W (Z^T q): the workgroup of frame f multiplies the blocks of ITS column of W (contiguous:
// the elimination-tree path of the frame, <= tree depth blocks) with its restricted product and stores each 8-vector at
// the block's slot in the row lists.  The row half (k_cg_update) then sums contiguous 8-vectors instead of gathering a
// 512 B block and a restricted vector per entry -- the rows near the root of the tree hold one entry per frame and were
// that kernel's critical path.  Fixed summation order on both sides: deterministic, no atomics.
struct CoarseColumns {
  const int* pos;       // frame -> elimination position (column of W)
  const int* wPtr;      // column j: blocks [wPtr[j], wPtr[j+1])
  const int* wSlot;     // block -> slot in the row lists
  const double* Wb;
  double* wq;           // [nW][kCB]
};
// masked restriction Z_f^T v_f of one frame's vector (LDS or global) by the calling workgroup: threads 0..6 the
// pose-like entries, wave 1 the sum over the depth-scale vertices
// PUBLISH: the values are read by OTHER workgroups of the same launch (k_pcg_tail): write-through agent-scope stores.
template <bool PUBLISH = false>
__device__ __forceinline__ void coarseRestrict(const Layout& L, const double* __restrict__ vf, int f, int tid,
                                               const unsigned char* __restrict__ modeActive, double* __restrict__ out);

// What the consumers of the preconditioned residual need from the coarse level: y = W Z^T r (k_coarse_apply_w) and
// the W blocks of the frame's elimination-tree path; c_f = sum_t W_tf^T y_t is formed where it is used, which
// saves a launch per PCG iteration -- or, Wb == nullptr and cF != nullptr, c of all frames was written by
// k_coarse_apply_wt (the default: the path walk in every consumer prologue costs more than the launch).
// Both null: level off.
struct CoarseView {
  const int* pos;    // frame -> elimination position
  const int* wPtr;   // W blocks of column j: [wPtr[j], wPtr[j+1]), rows wRow[.]
  const int* wRow;
  const double* Wb;
  const double* y;
  const unsigned char* modeActive;
  const int* fail;
  const double* cF;
  // third level (cvd_temporal.h: temporally coarse depth-grid level), tl == nullptr: off.  The level's workgroups leave the
  // frame's S coefficients (its temporal hats already interpolated) in tl[f][S]; a consumer adds the spatial prolongation
  // (P t)_i = sum_k tlW[v][k] tl[f][tlIdx[v] byte k] for depth vertex v = i - 7.
  const double* tl;
  const float4* tlW;         // [vertices] weights of the <= 4 coarse hats that are non-zero at the vertex
  const unsigned int* tlIdx; // [vertices] their indices, one byte each
  int tlS;
};
constexpr int kTlMaxS = 64;          // coarse hats per temporal node (LDS regions of the consumers: 2 x kTlMaxS doubles)
constexpr int kTlMaxWidth = 32;      // vertices in the support of one coarse hat (transposed table, ELL)
constexpr int kTlSpan = 10;          // node intervals per workgroup of a temporal level (TlStep::span): 11 rows, one per wave
__host__ __device__ inline int tlParts(int nn) { return nn > 1 ? (nn - 1 + kTlSpan - 1) / kTlSpan : 1; }
__host__ __device__ inline int tlRowsLds(int NT) { return NT + 2 * (kTlSpan + 1) + 2; }  // doubles of LDS of such a workgroup
struct TlTaps {
  float4 w;
  unsigned int idx;
};
// (issued with the other loads of a consumer's prologue; i = unknown of the frame block, nV depth vertices)
__device__ __forceinline__ void tlLoadTaps(const CoarseView& V, int i, int nV, TlTaps& t) {
  const bool in = i >= 7 && i < 7 + nV;
  const int v = in ? i - 7 : 0;
  t.w = V.tlW[v];
  t.idx = V.tlIdx[v];
  if (!in) t.w = make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ double tlAt(const double* __restrict__ tls, const TlTaps& t) {
  return static_cast<double>(t.w.x) * tls[t.idx & 255u] + static_cast<double>(t.w.y) * tls[(t.idx >> 8) & 255u] +
         static_cast<double>(t.w.z) * tls[(t.idx >> 16) & 255u] + static_cast<double>(t.w.w) * tls[t.idx >> 24];
}
// Per-iteration state of the third level inside the PCG kernels (Ainv == nullptr: off).  t = A_T^-1 P^T r is linear in r, so it
// obeys t <- t - alpha A_T^-1 (P^T q) exactly as the dense pose-graph level's c does (DenseStep): the finish half of a frame
// leaves the SPATIAL restriction of its product (sq[f][S]), S extra workgroups of the update launch -- one per coarse hat, the
// hat's nn temporal nodes are its rows -- reduce over the frames of each node, walk their rows of the inverse and leave tl.
// Unknown e = s * nn + a: hat s of temporal node a (node a sits at frame a * step, weight max(0, 1 - |f - a step| / step)).
struct TlStep {
  const double* Ainv;        // [NT][ld] f64
  double* sq;                // [F][S]
  double* rT;                // [NT] P^T r
  double* t;                 // [2][NT] A_T^-1 P^T r, double-buffered by the parity of the iteration count
  double* tl;                // [F][S] out (CoarseView::tl of the next product)
  double* dotPart;           // [S * parts] shares of r^T P t
  const int* fail;
  const float* elW;          // transposed vertex table of the restriction, ELL: entry k of hat s at [k * S + s]
  const unsigned char* elV;  //   its vertex
  int S, nn, step, NT, ld, width;
  int parts, span;           // the hat's nodes are walked by `parts` workgroups, `span` node intervals each (rows per workgroup = span + 1)
  double* rec;               // [NT] 16-byte records of the node sums' exchange inside k_pcg_tail (tlLevelRows), nullptr: every
                             // workgroup sums all of sq itself
  double weight;             // the level enters the additive preconditioner as weight * P A^-1 P^T (cvd_solver_options::temporal_weight)
};
// One wave: out[0..7] = c_f (zero for inactive modes / a failed factorisation).  lane = (r, c) of the 8x8 block.
__device__ __forceinline__ void coarseFrameCorrection(const CoarseView& V, int f, int lane, double* __restrict__ out) {
  const int r = lane >> 3;
  const int j = V.pos[f];
  const int w0 = V.wPtr[j], len = V.wPtr[j + 1] - w0;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  int t = 0;
  for (; t + 3 < len; t += 4) {
    a0 += V.Wb[static_cast<size_t>(w0 + t) * 64 + lane] * V.y[V.wRow[w0 + t] * kCB + r];
    a1 += V.Wb[static_cast<size_t>(w0 + t + 1) * 64 + lane] * V.y[V.wRow[w0 + t + 1] * kCB + r];
    a2 += V.Wb[static_cast<size_t>(w0 + t + 2) * 64 + lane] * V.y[V.wRow[w0 + t + 2] * kCB + r];
    a3 += V.Wb[static_cast<size_t>(w0 + t + 3) * 64 + lane] * V.y[V.wRow[w0 + t + 3] * kCB + r];
  }
  for (; t < len; ++t) a0 += V.Wb[static_cast<size_t>(w0 + t) * 64 + lane] * V.y[V.wRow[w0 + t] * kCB + r];
  double acc = (a0 + a1) + (a2 + a3);
  acc += __shfl_xor(acc, 8, 64);   // sum over r: lanes with equal c are 8 apart
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (lane < kCB) out[lane] = (*V.fail == 0 && V.modeActive[f * kCB + lane]) ? acc : 0.0;
}
// (Z c_f)_i from the 8 values of coarseFrameCorrection
__device__ __forceinline__ double coarseAtLds(const double* __restrict__ cl, const Layout& L, int i) {
  if (i < 7) return cl[i];
  if (L.N >= 1 && L.depthType != kDepthIdentity && i < 7 + L.nD && (L.N == 1 || ((i - 7) % L.N) == 0)) return cl[7];
  return 0.0;
}

// Wave-level sum of a double (64 lanes), result in every lane.  Four DPP butterfly steps inside each 16-lane row
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: no LDS traffic), then the four row sums are combined
// through v_readlane.
template <int CTRL>
__device__ __forceinline__ double dppMove(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void coarseColumnProducts(const CoarseColumns& cc, const double* __restrict__ qcF, int f, int tid,
                                                     int nThreads) {
  const int wv = tid >> 6, lane = tid & 63, c8 = lane & 7, nWv = nThreads >> 6;
  if (cc.Wb == nullptr) return;
  const int j = cc.pos[f];
  if (j < 0) return;
  const double qv = qcF[c8];
  // four blocks per wave and pass, loads first (clamped index, predicated store): the path is <= tree depth blocks long and
  // this sits at the very end of the kernel that completes q, so it is pure latency
  constexpr int U = 4;
  const int t1 = cc.wPtr[j + 1];
  for (int t0 = cc.wPtr[j] + wv; t0 < t1; t0 += U * nWv) {
    double v[U];
    int slot[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int tc = min(t0 + u * nWv, t1 - 1);
      v[u] = cc.Wb[static_cast<size_t>(tc) * 64 + lane];
      slot[u] = cc.wSlot[tc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      double a = v[u] * qv;
      a += dppMove<0xB1>(a);
      a += dppMove<0x4E>(a);
      a += dppMove<0x141>(a);
      if (c8 == 0 && t0 + u * nWv < t1) cc.wq[static_cast<size_t>(slot[u]) * kCB + (lane >> 3)] = a;
    }
  }
}
__device__ __forceinline__ double readLane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double waveSum(double v) {
  v += dppMove<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dppMove<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dppMove<0x141>(v);  // row_half_mirror
  v += dppMove<0x140>(v);  // row_mirror
  return (readLane(v, 0) + readLane(v, 16)) + (readLane(v, 32) + readLane(v, 48));
}

__device__ __forceinline__ void storeMaybePublished(double* p, double v, bool publish) {
  if (publish)
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    *p = v;
}
template <bool PUBLISH>
__device__ __forceinline__ void coarseRestrict(const Layout& L, const double* __restrict__ vf, int f, int tid,
                                               const unsigned char* __restrict__ modeActive, double* __restrict__ out) {
  // (inactive modes are identity rows of the coarse matrix: they must not feed the coarse solve)
  if (tid < 7) storeMaybePublished(out + f * kCB + tid, modeActive[f * kCB + tid] ? vf[tid] : 0.0, PUBLISH);
  if (tid >= 64 && tid < 128) {
    const int nV = (L.N >= 1 && L.depthType != kDepthIdentity) ? L.nD / L.N : 0;
    double a = 0.0;
    for (int v = tid - 64; v < nV; v += 64) a += vf[7 + v * L.N];
    a = waveSum(a);
    if (tid == 64) storeMaybePublished(out + f * kCB + 7, modeActive[f * kCB + 7] ? a : 0.0, PUBLISH);
  }
}

}  // namespace cvd
