// Scene-flow smoothness loss over frame triplets (f, f+1, f+2): reference lib/PoseOptimizer.cpp:321-423
// (SceneFlowSmoothnessLoss) and :1242-1339 (addSceneFlowSmoothnessLoss).  Off by default in the reference
// (smoothStaticWeight = smoothDynamicWeight = 0, :899).
//
// One constraint = three observations (one per frame) of what should be one scene point moving smoothly:
//   ReproDisparityLaplacian (default): the world points of frames 0 and 2 are reprojected into frame 1,
//       r = [ (u01 + u21 - 2 p1.x) / fy1,  (v01 + v21 - 2 p1.y) / fy1,  1/max(z01,eps) + 1/max(z21,eps) - 2/max(D1,eps) ]
//   EuclideanLaplacian: r = X0 + X2 - 2 X1 (world points).
// ScaledLoss(nullptr, w): cost 0.5 w |r|^2 with w = smoothStaticWeight / smoothDynamicWeight by the constraint's flag.
// IntrinsicsOptimization::Shared (reference lib/PoseOptimizer.cpp:1306-1330): one focal block, frame 0's, for all three
// observations -- its column is the sum of the three sides' focal columns and lives in frame 0's slot, exactly as for
// the pair constraints (k_assemble / k_shared_focal_fixup, k_matvec_finish).
//
// Layout: 36 B per constraint in HBM (3 x float2 ndc, 3 x float source depth, invalid = depth 0 in slot 0), groups
// keyed by the centre frame.  Kernels mirror the pair path: cost per group, frame-major assembly of g / H_ff
// (added to what the pair assembly wrote), group-major product with three partial rows per group that
// k_matvec_finish gathers like the pair rows.  The residual couples frames (f, f+2) as well: those off-diagonal
// blocks exist only inside the matrix-free product (neither preconditioner level needs them).
#pragma once

#include "cvd_kernels.h"

namespace cvd {

struct TripletTable {
  const float2* ndc;       // 3 per constraint
  const float* dsrc;       // 3 per constraint, dsrc[3c] <= 0: constraint inactive
  const unsigned char* isStatic;
  const long long* off;    // per group: [2g] begin, [2g+1] end of its constraints
  const int* center;       // per group: centre frame (frames centre-1, centre, centre+1)
  const int* slot;         // 3 per group: rows of the partial-product buffer
  int nGroups;
  int smoothType;          // cvd_smooth_loss_type
  double wStaticSqrt, wDynamicSqrt;
};

enum : int { kSmoothEuclidLaplacian = 0, kSmoothDisparityLaplacian = 1, kSmoothDepthRatio = 2, kSmoothLogDepth = 3 };

template <int KD, int KS>
struct TripletSample {
  double r[3];
  Side<KD, KS> s[3];
};

inline __global__ void k_build_triplet_table(int W, int H, float invAspect, long long C, const float* __restrict__ loc6,
                                      const int* __restrict__ cgroup, const int* __restrict__ center, int F,
                                      const unsigned char* __restrict__ inRange, const float* __restrict__ depth,
                                      float2* __restrict__ ndc, float* __restrict__ dsrc,
                                      unsigned long long* __restrict__ nValid) {
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool ok = false;
  if (c < C) {
    const int f1 = center[cgroup[c]];
    ok = f1 >= 1 && f1 + 1 < F && inRange[f1 - 1] && inRange[f1] && inRange[f1 + 1];
    float d[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < 3; ++k) {
      const float lx = loc6[c * 6 + 2 * k], ly = loc6[c * 6 + 2 * k + 1];
      float2 n;
      n.x = __fadd_rn(-1.f, __fmul_rn(2.f, lx));
      n.y = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly), invAspect));
      ndc[c * 3 + k] = n;
      if (ok) {
        int ix = static_cast<int>(__fmul_rn(lx, static_cast<float>(W)));
        int iy = static_cast<int>(__fmul_rn(__fdiv_rn(ly, invAspect), static_cast<float>(H)));
        ix = min(max(ix, 0), W - 1);
        iy = min(max(iy, 0), H - 1);
        d[k] = depth[static_cast<size_t>(f1 - 1 + k) * W * H + static_cast<size_t>(iy) * W + ix];
      }
    }
    ok = ok && isfinite(d[0]) && d[0] > 0.f && isfinite(d[1]) && d[1] > 0.f && isfinite(d[2]) && d[2] > 0.f;
    dsrc[c * 3 + 0] = ok ? d[0] : 0.f;
    dsrc[c * 3 + 1] = ok ? d[1] : 0.f;
    dsrc[c * 3 + 2] = ok ? d[2] : 0.f;
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(nValid, static_cast<unsigned long long>(__popcll(b)));
}

// Residual and compact Jacobian (Side layout of cvd_device.h) of one triplet constraint, unweighted.
template <int KD, int KS>
__device__ __forceinline__ void evalTriplet(const Layout& L, int smoothType, const FrameConst& F0, const FrameConst& F1,
                                            const FrameConst& F2, const double* __restrict__ x0,
                                            const double* __restrict__ x1, const double* __restrict__ x2,
                                            const float2* __restrict__ nd, const float* __restrict__ ds,
                                            TripletSample<KD, KS>& T) {
  constexpr double eps = 1e-6;
  const double A = L.aspect;
  const FrameConst* F[3] = {&F0, &F1, &F2};
  const double* X[3] = {x0, x1, x2};
  double D[3], p[3][2], cam[3][3], Rc[3][3], Xw[3][3], fy[3], fx[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    Side<KD, KS>& s = T.s[k];
    s.d = static_cast<double>(ds[k]);
    depthGather<KD>(L, nd[k].x, nd[k].y, ds[k], s.dt);
    spatialGather<KS>(L, nd[k].x, nd[k].y, s.st);
    D[k] = sideDepth(L, s, X[k]);
    p[k][0] = static_cast<double>(nd[k].x);
    p[k][1] = static_cast<double>(nd[k].y);
    if constexpr (KS > 0) {
      const double* ph = X[k] + 7 + L.nD;
      for (int t = 0; t < s.st.n; ++t) {
        p[k][0] += ph[s.st.idx[t] * 2] * s.st.w[t];
        p[k][1] += ph[s.st.idx[t] * 2 + 1] * s.st.w[t];
      }
    }
    fy[k] = F[k]->fy;
    fx[k] = fy[k] * A;
    cam[k][0] = p[k][0] * fx[k];
    cam[k][1] = p[k][1] * fy[k];
    cam[k][2] = -1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      Rc[k][i] = dot3(F[k]->R + 3 * i, cam[k]);
      Xw[k][i] = F[k]->t[i] + D[k] * Rc[k][i];
    }
  }
  // world-point derivatives of frame k: dX/dt = I, dX/dw_i, dX/df, dX/dD = Rc, dX/dP_c
  auto dXdw = [&](int k, int i, double out[3]) {
    out[0] = D[k] * dot3(F[k]->dR[i], cam[k]);
    out[1] = D[k] * dot3(F[k]->dR[i] + 3, cam[k]);
    out[2] = D[k] * dot3(F[k]->dR[i] + 6, cam[k]);
  };
  auto dXdf = [&](int k, double out[3]) {
    const double cf[3] = {p[k][0] * A, p[k][1], 0.0};
    out[0] = D[k] * dot3(F[k]->R, cf);
    out[1] = D[k] * dot3(F[k]->R + 3, cf);
    out[2] = D[k] * dot3(F[k]->R + 6, cf);
  };
  if (smoothType == kSmoothEuclidLaplacian) {
    const double coef[3] = {1.0, -2.0, 1.0};
#pragma unroll
    for (int r = 0; r < 3; ++r) T.r[r] = Xw[0][r] + Xw[2][r] - 2.0 * Xw[1][r];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Side<KD, KS>& s = T.s[k];
      double dw[3][3], df[3];
      for (int i = 0; i < 3; ++i) dXdw(k, i, dw[i]);
      dXdf(k, df);
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          s.Jp[r][c] = (r == c) ? coef[k] : 0.0;
          s.Jp[r][3 + c] = coef[k] * dw[c][r];
        }
        s.Jp[r][6] = coef[k] * df[r];
        s.JD[r] = coef[k] * Rc[k][r];
        s.JP[r][0] = coef[k] * D[k] * fx[k] * F[k]->R[3 * r + 0];
        s.JP[r][1] = coef[k] * D[k] * fy[k] * F[k]->R[3 * r + 1];
      }
    }
    return;
  }
  // Reprojection variants: frames 0 and 2 reprojected into frame 1.  Rows 0 / 1 are the same for all three; row 2 is
  // the disparity Laplacian, or the consistency of D_1 with z_0->1 + z_2->1 - D_1 (depth ratio / log depth,
  // reference lib/PoseOptimizer.cpp:393-408).
  Side<KD, KS>& s1 = T.s[1];
  // consistency variants: r_2 = h(base, other), base = D_1, other = z_0 + z_2 - D_1; ga = dh/dbase, gb = dh/dother
  // (Jet max / min: max(f, g) = f < g ? g : f, min(f, g) = g < f ? g : f)
  double consGa = 0.0, consGb = 0.0, consR = 0.0;
  if (smoothType != kSmoothDisparityLaplacian) {
    double other = -D[1];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = q ? 2 : 0;
      const double v[3] = {Xw[k][0] - F1.t[0], Xw[k][1] - F1.t[1], Xw[k][2] - F1.t[2]};
      other += -(F1.R[2] * v[0] + F1.R[5] * v[1] + F1.R[8] * v[2]);
    }
    const double base = D[1];
    const bool baseIsMax = !(base < other), baseIsMin = !(other < base);
    const double mx = baseIsMax ? base : other, mn = baseIsMin ? base : other;
    double dmx, dmn;
    if (smoothType == kSmoothDepthRatio) {
      consR = mx / mn - 1.0;
      dmx = 1.0 / mn;
      dmn = -mx / (mn * mn);
    } else {
      consR = log(mn / mx);
      dmn = 1.0 / mn;
      dmx = -1.0 / mx;
    }
    consGa = (baseIsMax ? dmx : 0.0) + (baseIsMin ? dmn : 0.0);
    consGb = (baseIsMax ? 0.0 : dmx) + (baseIsMin ? 0.0 : dmn);
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 7; ++c) s1.Jp[r][c] = 0.0;
    s1.JD[r] = 0.0;
    s1.JP[r][0] = 0.0;
    s1.JP[r][1] = 0.0;
  }
  const double ify = 1.0 / fy[1], ifx = 1.0 / fx[1];
  double usum = 0.0, vsum = 0.0, dsum = 0.0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = q ? 2 : 0;
    Side<KD, KS>& s = T.s[k];
    const double v[3] = {Xw[k][0] - F1.t[0], Xw[k][1] - F1.t[1], Xw[k][2] - F1.t[2]};
    const double q0 = F1.R[0] * v[0] + F1.R[3] * v[1] + F1.R[6] * v[2];
    const double q1 = F1.R[1] * v[0] + F1.R[4] * v[1] + F1.R[7] * v[2];
    const double q2 = F1.R[2] * v[0] + F1.R[5] * v[1] + F1.R[8] * v[2];
    const double z = -q2, iz = 1.0 / z;
    const double u = q0 * iz * ifx, w = q1 * iz * ify;
    const bool zo = !(z < eps);
    usum += u;
    vsum += w;
    dsum += zo ? iz : 1.0 / eps;
    // d r / d q: rows of M
    // (row 2: d(1/z)/dq_2 = 1/z^2 for the disparity Laplacian; z = -q_2 enters `other` with weight 1 otherwise)
    const double m22 = smoothType == kSmoothDisparityLaplacian ? (zo ? iz * iz : 0.0) : -consGb;
    const double M[3][3] = {{iz * ifx * ify, 0.0, u * iz * ify}, {0.0, iz * ify * ify, w * iz * ify}, {0.0, 0.0, m22}};
    double G[3][3];  // M R1^T
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int i = 0; i < 3; ++i) G[r][i] = M[r][0] * F1.R[i * 3 + 0] + M[r][1] * F1.R[i * 3 + 1] + M[r][2] * F1.R[i * 3 + 2];
    double dw[3][3], df[3];
    for (int i = 0; i < 3; ++i) dXdw(k, i, dw[i]);
    dXdf(k, df);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        s.Jp[r][c] = G[r][c];
        s.Jp[r][3 + c] = dot3(G[r], dw[c]);
        s1.Jp[r][c] -= G[r][c];
      }
      s.Jp[r][6] = dot3(G[r], df);
      s.JD[r] = dot3(G[r], Rc[k]);
      const double dpx[3] = {D[k] * fx[k] * F[k]->R[0], D[k] * fx[k] * F[k]->R[3], D[k] * fx[k] * F[k]->R[6]};
      const double dpy[3] = {D[k] * fy[k] * F[k]->R[1], D[k] * fy[k] * F[k]->R[4], D[k] * fy[k] * F[k]->R[7]};
      s.JP[r][0] = dot3(G[r], dpx);
      s.JP[r][1] = dot3(G[r], dpy);
    }
    // rotation of the target frame: d q / d w_i = dR1_i^T v
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double* Dm = F1.dR[i];
      const double dq[3] = {Dm[0] * v[0] + Dm[3] * v[1] + Dm[6] * v[2], Dm[1] * v[0] + Dm[4] * v[1] + Dm[7] * v[2],
                            Dm[2] * v[0] + Dm[5] * v[1] + Dm[8] * v[2]};
#pragma unroll
      for (int r = 0; r < 3; ++r) s1.Jp[r][3 + i] += dot3(M[r], dq);
    }
  }
  const bool bo = !(D[1] < eps);
  T.r[0] = (usum - 2.0 * p[1][0]) * ify;
  T.r[1] = (vsum - 2.0 * p[1][1]) * ify;
  T.r[2] = smoothType == kSmoothDisparityLaplacian ? dsum - 2.0 * (bo ? 1.0 / D[1] : 1.0 / eps) : consR;
  // frame 1's own dependence: fy1 (inside u, v through fx1 = A fy1 and fy1, and the outer division), D1, p1
  s1.Jp[0][6] = -(usum * ify + T.r[0]) * ify;
  s1.Jp[1][6] = -(vsum * ify + T.r[1]) * ify;
  s1.Jp[2][6] = 0.0;
  s1.JD[2] = smoothType == kSmoothDisparityLaplacian ? (bo ? 2.0 / (D[1] * D[1]) : 0.0) : consGa - consGb;
  s1.JP[0][0] = -2.0 * ify;
  s1.JP[1][1] = -2.0 * ify;
}

// ---- cost: one workgroup per triplet group, added to costFrame[centre] -----------------------------------------
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_cost_triplets(Layout L, TripletTable T, const double* __restrict__ x,
                                                       const FrameConst* __restrict__ fc, double* __restrict__ costFrame) {
  __shared__ double red[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int f1 = T.center[g];
  const int B = L.B;
  double acc = 0.0;
  for (long long c = T.off[2 * g] + tid; c < T.off[2 * g + 1]; c += 256) {
    if (!(T.dsrc[c * 3] > 0.f)) continue;
    const double ws = T.isStatic[c] ? T.wStaticSqrt : T.wDynamicSqrt;
    if (ws == 0.0) continue;
    TripletSample<KD, KS> s;
    evalTriplet<KD, KS>(L, T.smoothType, fc[f1 - 1], fc[f1], fc[f1 + 1], x + static_cast<size_t>(f1 - 1) * B,
                        x + static_cast<size_t>(f1) * B, x + static_cast<size_t>(f1 + 1) * B, T.ndc + c * 3,
                        T.dsrc + c * 3, s);
    acc += ws * ws * (s.r[0] * s.r[0] + s.r[1] * s.r[1] + s.r[2] * s.r[2]);
  }
  acc = waveSum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  // added to the centre frame's cost entry (written earlier in the stream by the per-frame kernels; one group per
  // centre frame, so there is no race) -- the multi-GPU reduction of the per-frame costs then covers it
  if (tid == 0) costFrame[f1] += 0.5 * ((red[0] + red[1]) + (red[2] + red[3]));
}

// ---- assembly: one workgroup per frame; adds the triplet part of g_f and H_ff to the pair assembly's output ---
// frameTripOff / frameTripList: per frame the (group << 2 | role) entries, role = position of the frame in the triplet.
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_assemble_triplets(Layout L, TripletTable T, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc,
                                                           const double* __restrict__ mask,
                                                           const int* __restrict__ ftOff, const int* __restrict__ ftList,
                                                           double* __restrict__ gOut, double* __restrict__ hOut,
                                                           double* __restrict__ focalG, double* __restrict__ focalH,
                                                           AsmPanels panels, int panelCap) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* Hs = sm;             // one row panel of the packed lower triangle (AsmPanels, cvd_kernels.h)
  double* gs = Hs + panelCap;
  const int f = blockIdx.x, tid = threadIdx.x;
  if (ftOff[f] == ftOff[f + 1]) return;
  for (int i = tid; i < B; i += 256) gs[i] = 0.0;
  const bool shared = L.intrOpt == kIntrShared;
  double shG = 0.0, shH = 0.0;  // shared focal: gradient / squared column norm, taken once per constraint (centre visit)
  const double* mf = mask + static_cast<size_t>(f) * B;
  double* hf = hOut + static_cast<size_t>(f) * B * B;
  for (int pass = 0; pass < panels.n; ++pass) {
  const int r0 = panels.row[pass], r1 = panels.row[pass + 1];
  const int base = r0 * (r0 + 1) / 2, npk = r1 * (r1 + 1) / 2 - base;
  const bool first = pass == 0;
  __syncthreads();
  for (int i = tid; i < npk; i += 256) Hs[i] = 0.0;
  __syncthreads();
  for (int e = ftOff[f]; e < ftOff[f + 1]; ++e) {
    const int code = ftList[e];
    const int g = code >> 2, role = code & 3;
    const int f1 = T.center[g];
    for (long long c = T.off[2 * g] + tid; c < T.off[2 * g + 1]; c += 256) {
      if (!(T.dsrc[c * 3] > 0.f)) continue;
      const double ws = T.isStatic[c] ? T.wStaticSqrt : T.wDynamicSqrt;
      if (ws == 0.0) continue;
      TripletSample<KD, KS> s;
      evalTriplet<KD, KS>(L, T.smoothType, fc[f1 - 1], fc[f1], fc[f1 + 1], x + static_cast<size_t>(f1 - 1) * B,
                          x + static_cast<size_t>(f1) * B, x + static_cast<size_t>(f1 + 1) * B, T.ndc + c * 3,
                          T.dsrc + c * 3, s);
      const double w = ws * ws;
      if (shared) {
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          const double tot = s.s[0].Jp[rr][6] + s.s[1].Jp[rr][6] + s.s[2].Jp[rr][6];
          s.s[0].Jp[rr][6] = tot;
          s.s[1].Jp[rr][6] = tot;
          s.s[2].Jp[rr][6] = tot;
        }
        if (role == 1 && first) {
          const Side<KD, KS>& c1 = s.s[1];
          shG += w * (c1.Jp[0][6] * s.r[0] + c1.Jp[1][6] * s.r[1] + c1.Jp[2][6] * s.r[2]);
          shH += w * (c1.Jp[0][6] * c1.Jp[0][6] + c1.Jp[1][6] * c1.Jp[1][6] + c1.Jp[2][6] * c1.Jp[2][6]);
        }
      }
      const Side<KD, KS>& me = s.s[role];
      // all columns through LDS atomics (this loss is off by default: no register-blocked fast path)
      if (first) {
        for (int i = 0; i < 7; ++i) {
          atomicAdd(&gs[i], w * (me.Jp[0][i] * s.r[0] + me.Jp[1][i] * s.r[1] + me.Jp[2][i] * s.r[2]));
          for (int j = 0; j <= i; ++j)
            atomicAdd(&Hs[packedIdx(i, j)], w * (me.Jp[0][i] * me.Jp[0][j] + me.Jp[1][i] * me.Jp[1][j] + me.Jp[2][i] * me.Jp[2][j]));
        }
      }
      const int nt = sideNumTapCols(L, me);
      for (int t = 0; t < nt; ++t) {
        int ct;
        double Jt[3];
        sideTapCol(L, me, t, ct, Jt);
        const double wj0 = w * Jt[0], wj1 = w * Jt[1], wj2 = w * Jt[2];
        if (first) atomicAdd(&gs[ct], wj0 * s.r[0] + wj1 * s.r[1] + wj2 * s.r[2]);
        if (ct >= r0 && ct < r1) {
          const int rowBase = ct * (ct + 1) / 2 - base;
          for (int i = 0; i < 7; ++i) atomicAdd(&Hs[rowBase + i], wj0 * me.Jp[0][i] + wj1 * me.Jp[1][i] + wj2 * me.Jp[2][i]);
        }
        for (int t2 = 0; t2 <= t; ++t2) {
          int c2;
          double J2[3];
          sideTapCol(L, me, t2, c2, J2);
          const int hi = ct > c2 ? ct : c2, lo = ct > c2 ? c2 : ct;
          if (hi >= r0 && hi < r1) atomicAdd(&Hs[packedIdx(hi, lo) - base], wj0 * J2[0] + wj1 * J2[1] + wj2 * J2[2]);
        }
      }
    }
  }
  __syncthreads();
  // panel write-out: ADDED to what the pair assembly wrote
  for (int idx = tid; idx < npk; idx += 256) {
    int i = static_cast<int>((sqrt(8.0 * static_cast<double>(idx + base) + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > idx + base) --i;
    while ((i + 1) * (i + 2) / 2 <= idx + base) ++i;
    const int j = idx + base - i * (i + 1) / 2;
    // shared focal (as in k_assemble): gradient / diagonal go to frame 0's slot through k_shared_focal_fixup (which runs
    // after this kernel); the column's couplings with other frames' unknowns are not held by the block preconditioner
    if (shared && ((i == 6 && j == 6) || (f != 0 && (i == 6 || j == 6)))) continue;
    const double v = Hs[idx] * mf[i] * mf[j];
    hf[static_cast<size_t>(i) * B + j] += v;
    if (i != j) hf[static_cast<size_t>(j) * B + i] += v;
  }
  }  // pass
  __syncthreads();
  if (shared) {
    __shared__ double redS[8];
    shG = waveSum(shG);
    shH = waveSum(shH);
    if ((tid & 63) == 0) { redS[tid >> 6] = shG; redS[4 + (tid >> 6)] = shH; }
    __syncthreads();
    if (tid == 0) {
      focalG[f] += redS[0] + redS[1] + redS[2] + redS[3];
      focalH[f] += redS[4] + redS[5] + redS[6] + redS[7];
    }
  }
  for (int i = tid; i < B; i += 256)
    if (!(shared && i == 6)) gOut[static_cast<size_t>(f) * B + i] += gs[i] * mf[i];
}

// ---- product: one workgroup per triplet group, three partial rows (one per frame) -------------------------------
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_matvec_triplets(Layout L, TripletTable T, const double* __restrict__ x,
                                                         const FrameConst* __restrict__ fc,
                                                         const double* __restrict__ mask, const double* __restrict__ z,
                                                         const double* __restrict__ pOld,
                                                         const double* __restrict__ scal, int useBeta,
                                                         double* __restrict__ qPart, CoarseView V) {
  if (scal[S_DONE] != 0.0) return;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* xs = sm;            // 3 x B
  double* ps = xs + 3 * B;    // 3 x B (masked direction)
  double* qs = ps + 3 * B;    // 3 x B
  double* cl = qs + 3 * B;    // 3 x kCB coarse corrections
  const int g = blockIdx.x, tid = threadIdx.x;
  const int f1 = T.center[g];
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  if (tid < 3 * kCB) {
    const int k = tid / kCB, m = tid % kCB;
    cl[tid] = (V.cF != nullptr) ? V.cF[(f1 - 1 + k) * kCB + m] : 0.0;
  }
  __syncthreads();
  for (int i = tid; i < 3 * B; i += 256) {
    const int k = i / B, j = i - k * B;
    const size_t gi = static_cast<size_t>(f1 - 1 + k) * B + j;
    xs[i] = x[gi];
    ps[i] = (z[gi] + coarseAtLds(cl + k * kCB, L, j) + (useBeta ? beta * pOld[gi] : 0.0)) * mask[gi];
    qs[i] = 0.0;
  }
  __syncthreads();
  if (L.intrOpt == kIntrShared && tid < 3) {
    // every observation's focal column is frame 0's slot (reference lib/PoseOptimizer.cpp:1306-1330)
    ps[tid * B + 6] = (z[6] + (useBeta ? beta * pOld[6] : 0.0)) * mask[6];
  }
  if (L.intrOpt == kIntrShared) __syncthreads();
  for (long long c = T.off[2 * g] + tid; c < T.off[2 * g + 1]; c += 256) {
    if (!(T.dsrc[c * 3] > 0.f)) continue;
    const double ws = T.isStatic[c] ? T.wStaticSqrt : T.wDynamicSqrt;
    if (ws == 0.0) continue;
    TripletSample<KD, KS> s;
    evalTriplet<KD, KS>(L, T.smoothType, fc[f1 - 1], fc[f1], fc[f1 + 1], xs, xs + B, xs + 2 * B, T.ndc + c * 3,
                        T.dsrc + c * 3, s);
    double t[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 3; ++k) sideJp(L, s.s[k], ps + k * B, t);
    const double w = ws * ws;
    t[0] *= w; t[1] *= w; t[2] *= w;
    for (int k = 0; k < 3; ++k) {
      const Side<KD, KS>& sd = s.s[k];
      double* q = qs + k * B;
      for (int i = 0; i < 7; ++i) atomicAdd(&q[i], sd.Jp[0][i] * t[0] + sd.Jp[1][i] * t[1] + sd.Jp[2][i] * t[2]);
      const int nt = sideNumTapCols(L, sd);
      for (int a = 0; a < nt; ++a) {
        int col;
        double J[3];
        sideTapCol(L, sd, a, col, J);
        atomicAdd(&q[col], J[0] * t[0] + J[1] * t[1] + J[2] * t[2]);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < 3 * B; i += 256) {
    const int k = i / B, j = i - k * B;
    qPart[static_cast<size_t>(T.slot[g * 3 + k]) * B + j] = qs[i];
  }
}

}  // namespace cvd
