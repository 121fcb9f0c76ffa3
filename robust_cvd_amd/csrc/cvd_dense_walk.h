// robust_cvd_amd/csrc/cvd_dense_walk.h
//
// DENSE MODE, the Jacobian evaluation in ONE walk over the pixels (round 6).  Reference: with matchSeparation = 0
// every masked in-bounds pixel of a frame pair is a constraint (lib/FlowConstraints.cpp:315-329, 381-395), each added as one
// residual block touching the two frames of its pair (lib/PoseOptimizer.cpp:1185-1232).
//
// Rounds 2-5 evaluated a constraint's Jacobian chain four times per Jacobian evaluation: once per frame side in the frame-major
// k_assemble_fast<.., DENSE> (own-side 7 x 7 block in registers), once in k_cross_assemble<.., false> (pose rows of the cross
// block: 49 more register accumulators) and once per column panel of the grid x grid part.  Three kernels at 253 VGPRs.  Here
//
//   k_dense_walk       one workgroup per DIRECTED pair: flow / mask / depths are read once, the chain is formed once, and ALL of the
//                      constraint's contributions leave the lane:
//                        pose x pose of both diagonal blocks and of the cross block + the pose gradient (14 + 1 columns): the
//                          rows sqrt(rho') [Jp_s | Jp_t | r] go through a per-wave LDS tile into v_mfma_f64_16x16x4 -- the Gram matrix
//                          of 15 columns is ONE 16 x 16 accumulator tile (8 VGPRs instead of 119 accumulators), on the matrix pipe,
//                          which is idle otherwise;
//                        everything with a grid vertex in it (rank one in (residual, tap)): LDS f64 atomics into column-major
//                          [column][vertex] side blocks (consecutive lanes = different cells = different banks);
//                        grid x grid of the CROSS block: the one scalar rho' JD_s . JD_t d_s d_t per pixel goes to HBM (8 B) for
//                          k_dense_gg -- G^2 doubles (231 KB at 17 x 10) do not fit beside the rest.
//                      The workgroup's sums are written as one compact RECORD per directed pair (256 + 40 G doubles).
//   k_assemble_fast<.., FOLD>   per frame: sums its records into H_ff / g / cost (a gather: no atomics), then regularisers etc. as before
//   k_dense_fold_cross per undirected pair: pose rows / columns of X_ab from the two directions' records
//   k_dense_gg         per (undirected pair, column panel): grid x grid of X_ab from the per-pixel scalars and the flow (taps only:
//                      no Jacobian chain, no depth reads)
//
// Record of a directed pair s -> t (doubles; G = vertices of the depth grid, one value parameter per vertex):
//   [0, 256)              tile  D[m][n], m, n in [pose_s (7) | pose_t (7) | r | 0]: D = sum rho' J^T J  (row 14 = J^T rho' r)
//   256 + c G + v         side S, c < 15: theta_s[v] x {pose_s (c = 0..6), pose_t (7..13), gradient (14)}
//   256 + 15 G + c G + v  side T, c < 15: theta_t[v] x {pose_s, pose_t, gradient}
//   256 + 30 G + d G + v  band of theta_s x theta_s: (v, v + off[d]), off = {0, 1, gx - 1, gx, gx + 1} (vertex pairs of one cell)
//   256 + 35 G + d G + v  band of theta_t x theta_t
//   256 + 40 G            sum of rho (the pair's cost)
#pragma once

#include "cvd_cross.h"

namespace cvd {

constexpr int kDwThreads = CVD_DETERMINISTIC ? 64 : 512;
constexpr int kDwRun = 16;           // consecutive pixels per lane (kDenseRun: lanes of a wave touch different cells)
constexpr int kDwLd = 17;            // row stride (doubles) of the per-wave MFMA staging tile [64 constraints][16 columns]
constexpr int kDwCols = 15;          // side-block columns
// (dwRecordDoubles, DenseRecords: cvd_kernels.h, beside the frame-major kernel that folds the records)
__host__ __device__ inline size_t dwLdsBytes(int G, int B, int threads) {
  return (static_cast<size_t>(dwRecordDoubles(G)) + 2 * B + 2 * (sizeof(FrameConst) / 8) + static_cast<size_t>(threads / 64) * 64 * kDwLd) * 8;
}

// Work list: one record per entry.
struct DenseWalkList {
  const int* pair;          // directed pair of the record
  const long long* range;   // 2 per record: pixel slots [begin, end) (within the pair)
  int count;
};

// ndc of both end points of a dense-mode constraint from its flow vector, WITHOUT the depth fetch (k_dense_gg: taps only).  The same
// float arithmetic as denseConstraintFromFlow; false: target out of bounds.
__device__ __forceinline__ bool denseNdcFromFlow(const Table& T, int pix, float2 f, float4& n) {
  const int iy = pix / T.W, ix = pix - iy * T.W;
  const float fx1 = __fadd_rn(static_cast<float>(ix), f.x), fy1 = __fadd_rn(static_cast<float>(iy), f.y);
  if (!(isfinite(fx1) && isfinite(fy1))) return false;
  const int ix1 = static_cast<int>(__fadd_rn(fx1, 0.5f)), iy1 = static_cast<int>(__fadd_rn(fy1, 0.5f));
  if (ix1 < 0 || ix1 >= T.W || iy1 < 0 || iy1 >= T.H) return false;
  const float lx0 = __fmul_rn(static_cast<float>(ix), T.sx), ly0 = __fmul_rn(static_cast<float>(iy), T.sy);
  const float lx1 = __fmul_rn(fx1, T.sx), ly1 = __fmul_rn(fy1, T.sy);
  n.x = __fadd_rn(-1.f, __fmul_rn(2.f, lx0));
  n.y = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly0), T.invAspect));
  n.z = __fadd_rn(-1.f, __fmul_rn(2.f, lx1));
  n.w = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly1), T.invAspect));
  return true;
}

// One constraint s -> t.  What stays in registers between its phases is the minimum the Jacobian can be re-expanded from:
//   d r / d q = M = [m00 0 m02; 0 m11 m12; 0 0 m22], the depth z' = -q_z, the residual, rho', d r_2 / d D_t, D_s and R_s c
// (16 doubles).  Rotation columns in cross-product form (dR/dw_i = [a_i]x R, FrameConst::Jl): d X / d w_s,i = a_i x y with
// y = D_s R_s c, d q / d w_t,i = R_t^T (v x a_i), v = X - t_t.
struct DwState {
  double m00, m02, m11, m12, m22;
  double zz;
  double r[3];
  double w, rho0, JDT2, Da;
  double Rca[3];
};

// bilinear taps of one end point in compact form: base vertex and the two fractions (10 VGPRs for both sides instead of 24)
struct DwTaps {
  int i0;
  double rx, ry;
  __device__ __forceinline__ double Wt(int k) const { return ((k & 1) ? rx : 1.0 - rx) * ((k & 2) ? ry : 1.0 - ry); }
};
__device__ __forceinline__ void dwGather(const Layout& L, float lx, float ly, DwTaps& t) {
  int ix, iy;
  gridCellFast(lx, 0.5 * static_cast<double>(L.gx - 1), L.maxcx, ix, t.rx);
  gridCellFast(ly, 0.5 * static_cast<double>(L.gy - 1), L.maxcy, iy, t.ry);
  t.i0 = ix + iy * L.gx;
}

__device__ __forceinline__ void dwChain(const Layout& L, const FrameConst& Fs, const FrameConst& Ft, const double* __restrict__ xs,
                                        const double* __restrict__ xt, const float4& nd, double da, double db, const DwTaps& ts,
                                        const DwTaps& tt, DwState& o) {
  constexpr double eps = 1e-6;
  const double A = L.aspect;
  const int gx = L.gx;
  // (the same tap order and summation order as fastGather / the other fast kernels)
  double Da = 0.0, Db = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Da += da * xs[7 + ts.i0 + (k & 1) + (k >> 1) * gx] * ts.Wt(k);
    Db += db * xt[7 + tt.i0 + (k & 1) + (k >> 1) * gx] * tt.Wt(k);
  }
  o.Da = Da;
  const double fys = Fs.fy, fxs = Fs.fy * A;
  const double fyt = Ft.fy;
  const double ifyt = 1.0 / fyt, ifxt = 1.0 / (fyt * A);
  const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
  const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
  const double ca[3] = {pax * fxs, pay * fys, -1.0};
  double v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.Rca[i] = dot3(Fs.R + 3 * i, ca);
    v[i] = Fs.t[i] + o.Rca[i] * Da - Ft.t[i];
  }
  const double q0 = Ft.R[0] * v[0] + Ft.R[3] * v[1] + Ft.R[6] * v[2];
  const double q1 = Ft.R[1] * v[0] + Ft.R[4] * v[1] + Ft.R[7] * v[2];
  const double q2 = Ft.R[2] * v[0] + Ft.R[5] * v[1] + Ft.R[8] * v[2];
  const double zz = -q2;
  const double iz = 1.0 / zz;
  const double u = q0 * iz * ifxt;
  const double vv = q1 * iz * ifyt;
  o.zz = zz;
  o.r[0] = (u - pbx) * L.ws;
  o.r[1] = (vv - pby) * L.ws;
  double dr2dA, dr2dDb;
  if (L.lossType == kLossDisparity) {
    const bool zo = !(zz < eps), bo = !(Db < eps);
    const double izc = zo ? iz : 1.0 / eps, ibc = 1.0 / (bo ? Db : eps);
    o.r[2] = (izc - ibc) * L.wd;
    dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
    dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
  } else {
    const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
    const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
    if (L.lossType == kLossRatio) {
      o.r[2] = (mx / mn - 1.0) * L.wd;
      const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
      dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
      dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
    } else {
      o.r[2] = log(mn / mx) * L.wd;
      const double dmn = 1.0 / mn, dmx = -1.0 / mx;
      dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
      dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
    }
  }
  robustRho(L, o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2], o.rho0, o.w);
  const double wiz = L.ws * iz;
  o.m00 = wiz * ifxt;
  o.m11 = wiz * ifyt;
  o.m02 = wiz * u;
  o.m12 = wiz * vv;
  o.m22 = -dr2dA;
  o.JDT2 = dr2dDb;
}

// The 15 columns [Jp_s (7) | Jp_t (7) | r] contracted with a vector mu over the components of q (g = R_t mu over those of X): with
// mu = row r of M this is row r of the Jacobian (c13, c14 = its focal-of-target entry and its residual); with
// mu = sum_r rho' (d r_r / d D_s) M_r it is the pose x theta_s column block, with mu = rho' (d r_2 / d D_t) M_2 the pose x theta_t
// one -- the columns are LINEAR in mu, so the side blocks need no accumulators beside the staging of the rows for the matrix pipe.
// jds = g . R_s c, the same contraction of d r / d D_s.
__device__ __forceinline__ void dwColumns(const DwState& c, const FrameConst& Fs, const FrameConst& Ft, double mu0, double mu1,
                                          double mu2, double c13, double c14, double (&J)[15], double& jds) {
  double g[3], y[3], v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g[i] = Ft.R[3 * i] * mu0 + Ft.R[3 * i + 1] * mu1 + Ft.R[3 * i + 2] * mu2;
    y[i] = c.Rca[i] * c.Da;
    v[i] = (Fs.t[i] + y[i]) - Ft.t[i];
  }
  const double n0 = y[1] * g[2] - y[2] * g[1], n1 = y[2] * g[0] - y[0] * g[2], n2 = y[0] * g[1] - y[1] * g[0];   // y x g
  const double m0 = g[1] * v[2] - g[2] * v[1], m1 = g[2] * v[0] - g[0] * v[2], m2 = g[0] * v[1] - g[1] * v[0];   // g x v
  J[0] = g[0]; J[1] = g[1]; J[2] = g[2];
  J[7] = -g[0]; J[8] = -g[1]; J[9] = -g[2];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    J[3 + i] = Fs.Jl[3 * i] * n0 + Fs.Jl[3 * i + 1] * n1 + Fs.Jl[3 * i + 2] * n2;
    J[10 + i] = Ft.Jl[3 * i] * m0 + Ft.Jl[3 * i + 1] * m1 + Ft.Jl[3 * i + 2] * m2;
  }
  jds = dot3(g, c.Rca);
  // d X / d fy_s = D_s R_s (p_x A, p_y, 0) = D_s (R_s c + R_s e_z) / fy_s
  J[6] = (c.Da / Fs.fy) * (jds + g[0] * Fs.R[2] + g[1] * Fs.R[5] + g[2] * Fs.R[8]);
  J[13] = c13;
  J[14] = c14;
}

#ifdef CVD_DW_PROFILE
__device__ unsigned long long g_dwProf[4096 * 8];
#define DW_STAMP(slot) do { if (threadIdx.x == 0) g_dwProf[(blockIdx.x & 4095) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DW_STAMP(slot) do {} while (0)
#endif

template <int KD>
inline __global__ __launch_bounds__(kDwThreads) void k_dense_walk(Layout L, Table T, DenseWalkList wl, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc, double* __restrict__ records,
                                                           double* __restrict__ ggOut) {
  static_assert(KD == 4, "bilinear depth grids (the explicit-block scope of the dense mode)");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD, gx = L.gx;
  const int recN = dwRecordDoubles(G);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = kDwThreads / 64;
  double* acc = sm;
  double* xs = acc + recN;
  double* xt = xs + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(xt + B);
  double* scr = reinterpret_cast<double*>(fcs + 2) + wave * (64 * kDwLd);
  double* sideS = acc + 256;
  double* sideT = sideS + kDwCols * G;
  double* bandS = acc + 256 + 30 * G;
  double* bandT = bandS + 5 * G;
  const int rec = blockIdx.x;
  const int p = wl.pair[rec];
  const int fs = T.pairA[p], ft = T.pairB[p];
  DW_STAMP(0);
  for (int i = tid; i < recN; i += kDwThreads) acc[i] = 0.0;
  for (int i = tid; i < B; i += kDwThreads) {
    xs[i] = x[static_cast<size_t>(fs) * B + i];
    xt[i] = x[static_cast<size_t>(ft) * B + i];
  }
  constexpr int FCW = sizeof(FrameConst) / 8;
  for (int i = tid; i < 2 * FCW; i += kDwThreads)
    reinterpret_cast<double*>(fcs)[i] = reinterpret_cast<const double*>(fc + (i < FCW ? fs : ft))[i % FCW];
  scr[lane * kDwLd + 15] = 0.0;   // (the 16th column of the staging tile: never written again)
  __syncthreads();
  DW_STAMP(1);
  // (wave-uniform read-only global data at an address that depends on blockIdx only: scalar loads, as in the hot product)
  const FrameConst& Fs = fc[fs];
  const FrameConst& Ft = fc[ft];
  const long long pixBase = T.pairOff[p];
  const long long cb = pixBase + wl.range[rec * 2], ce = pixBase + wl.range[rec * 2 + 1];

  cvd_d4 tile0 = {0.0, 0.0, 0.0, 0.0}, tile1 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  const int mk = lane >> 4, mc = lane & 15;  // MFMA operand element [k = lane >> 4][column = lane & 15]
  constexpr long long kUnit = 64LL * kDwRun;
  for (long long u0 = cb + wave * kUnit; u0 < ce; u0 += NW * kUnit) {
    const long long cFirst = u0 + static_cast<long long>(lane) * kDwRun;
    const int iFirst = lane * kDwRun;
    const long long rem = ce - u0;
    const int iStop = static_cast<int>(rem < (iFirst + kDwRun) ? (rem < iFirst ? iFirst : rem) : (iFirst + kDwRun));
    RecordStream<true> rs;
    rs.prime(T, u0, iFirst, iStop);
    for (int t = 0; t < kDwRun; ++t) {
      const int i = iFirst + t;
      float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
      float2 d = make_float2(0.f, 0.f);
      bool valid = false;
      if (i < iStop) valid = rs.take(T, u0, i, 1, iStop, pixBase, fs, ft, nd, d);
      if (__builtin_amdgcn_readfirstlane(__ballot(valid) == 0ull ? 1 : 0)) {
        if (i < iStop) ggOut[cFirst + t] = 0.0;
        continue;  // (wave-uniform)
      }
      DwState ch;
      DwTaps ts, tt;
      const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
      double sw = 0.0;
      if (valid) {
        dwGather(L, nd.x, nd.y, ts);
        dwGather(L, nd.z, nd.w, tt);
        dwChain(L, Fs, Ft, xs, xt, nd, da, db, ts, tt, ch);
        sw = sqrt(ch.w);
        cost += ch.rho0;
      }
      // d r_0,1 / d fy_t = -ws (u, v) / fy_t = -(m02, m12) z' / fy_t
      const double zf = valid ? -ch.zz / Ft.fy : 0.0;
      // ---- the three residual rows sqrt(rho') [Jp_s | Jp_t | r] through the wave's staging tile into the Gram tile
      double wj[3] = {0.0, 0.0, 0.0};  // rho' d r_r / d D_s
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        if (valid) {
          double J[15], jds;
          if (r == 0) dwColumns(ch, Fs, Ft, ch.m00, 0.0, ch.m02, ch.m02 * zf, ch.r[0], J, jds);
          else if (r == 1) dwColumns(ch, Fs, Ft, 0.0, ch.m11, ch.m12, ch.m12 * zf, ch.r[1], J, jds);
          else dwColumns(ch, Fs, Ft, 0.0, 0.0, ch.m22, 0.0, ch.r[2], J, jds);
#pragma unroll
          for (int c = 0; c < 15; ++c) scr[lane * kDwLd + c] = sw * J[c];
          wj[r] = ch.w * jds;
        } else {
#pragma unroll
          for (int c = 0; c < 15; ++c) scr[lane * kDwLd + c] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const double a0 = scr[(4 * j + mk) * kDwLd + mc];
          const double a1 = scr[(4 * j + 4 + mk) * kDwLd + mc];
          tile0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, tile0, 0, 0, 0);
          tile1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, tile1, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (valid) {
        const double wt = ch.w * ch.JDT2;   // rho' d r_2 / d D_t
        ggOut[cFirst + t] = wj[2] * ch.JDT2 * da * db;
        double fa[4], fb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { fa[k] = ts.Wt(k) * da; fb[k] = tt.Wt(k) * db; }
        double sSS;
        {  // theta_s x {pose_s, pose_t, gradient}: mu = sum_r wj_r M_r
          double vS[15], jds;
          dwColumns(ch, Fs, Ft, wj[0] * ch.m00, wj[1] * ch.m11, wj[0] * ch.m02 + wj[1] * ch.m12 + wj[2] * ch.m22,
                    (wj[0] * ch.m02 + wj[1] * ch.m12) * zf, wj[0] * ch.r[0] + wj[1] * ch.r[1] + wj[2] * ch.r[2], vS, jds);
          sSS = jds;   // sum_r rho' (d r_r / d D_s)^2
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double* col = sideS + ts.i0 + (k & 1) + (k >> 1) * gx;
#pragma unroll
            for (int c = 0; c < 15; ++c) atomicAdd(&col[c * G], vS[c] * fa[k]);
            __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise forms all products before the first atomic)
          }
        }
        {  // theta_t x {pose_s, pose_t, gradient}: d r / d D_t has the one row 2
          double vT[15], jds;
          dwColumns(ch, Fs, Ft, 0.0, 0.0, wt * ch.m22, 0.0, wt * ch.r[2], vT, jds);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            double* col = sideT + tt.i0 + (k & 1) + (k >> 1) * gx;
#pragma unroll
            for (int c = 0; c < 15; ++c)
              if (c != 13) atomicAdd(&col[c * G], vT[c] * fb[k]);  // (d r_2 / d fy_t = 0)
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        // vertex pairs of the cell, lower vertex first: offsets {0, 1, gx - 1, gx, gx + 1} -> band index {0, 1, 2, 3, 4}
        const double sTT = wt * ch.JDT2;
        const int i0 = ts.i0, j0 = tt.i0;
        const double a0 = sSS * fa[0], a1 = sSS * fa[1], a2 = sSS * fa[2], a3 = sSS * fa[3];
        atomicAdd(&bandS[0 * G + i0], a0 * fa[0]);
        atomicAdd(&bandS[1 * G + i0], a0 * fa[1]);
        atomicAdd(&bandS[3 * G + i0], a0 * fa[2]);
        atomicAdd(&bandS[4 * G + i0], a0 * fa[3]);
        atomicAdd(&bandS[0 * G + i0 + 1], a1 * fa[1]);
        atomicAdd(&bandS[2 * G + i0 + 1], a1 * fa[2]);
        atomicAdd(&bandS[3 * G + i0 + 1], a1 * fa[3]);
        atomicAdd(&bandS[0 * G + i0 + gx], a2 * fa[2]);
        atomicAdd(&bandS[1 * G + i0 + gx], a2 * fa[3]);
        atomicAdd(&bandS[0 * G + i0 + gx + 1], a3 * fa[3]);
        __builtin_amdgcn_sched_barrier(0);
        const double b0 = sTT * fb[0], b1 = sTT * fb[1], b2 = sTT * fb[2], b3 = sTT * fb[3];
        atomicAdd(&bandT[0 * G + j0], b0 * fb[0]);
        atomicAdd(&bandT[1 * G + j0], b0 * fb[1]);
        atomicAdd(&bandT[3 * G + j0], b0 * fb[2]);
        atomicAdd(&bandT[4 * G + j0], b0 * fb[3]);
        atomicAdd(&bandT[0 * G + j0 + 1], b1 * fb[1]);
        atomicAdd(&bandT[2 * G + j0 + 1], b1 * fb[2]);
        atomicAdd(&bandT[3 * G + j0 + 1], b1 * fb[3]);
        atomicAdd(&bandT[0 * G + j0 + gx], b2 * fb[2]);
        atomicAdd(&bandT[1 * G + j0 + gx], b2 * fb[3]);
        atomicAdd(&bandT[0 * G + j0 + gx + 1], b3 * fb[3]);
      } else if (i < iStop) {
        ggOut[cFirst + t] = 0.0;
      }
    }
  }
  DW_STAMP(2);
  // ---- the waves' Gram tiles and cost into the record (D: column = lane & 15, row = (lane >> 4) + 4 reg)
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicAdd(&acc[((lane >> 4) + 4 * q) * 16 + (lane & 15)], tile0[q] + tile1[q]);
  cost = waveSum(cost);
  if (lane == 0) atomicAdd(&acc[256 + 40 * G], cost);
  __syncthreads();
  double* out = records + static_cast<size_t>(rec) * recN;
  for (int i = tid; i < recN; i += kDwThreads) out[i] = acc[i];
  DW_STAMP(3);
}

// Pose rows / columns of X_ab (rows = fa's unknowns, columns = fb's) from the records of a -> b and b -> a.
inline __global__ __launch_bounds__(256) void k_dense_fold_cross(Layout L, CrossPairs cp, const int* __restrict__ xDir,
                                                          const int* __restrict__ recOff, const double* __restrict__ records,
                                                          double* __restrict__ X) {
  const int B = L.B, G = L.nD;
  const int recN = dwRecordDoubles(G);
  const int pair = blockIdx.x, tid = threadIdx.x;
  double* Xp = X + static_cast<size_t>(pair) * B * B;
  const int nOut = 49 + 14 * G;
  for (int o = tid; o < nOut; o += 256) {
    double s = 0.0;
    int row, col;
    for (int dir = 0; dir < 2; ++dir) {
      const int p = xDir[pair * 2 + dir];
      if (p < 0) continue;
      for (int q = recOff[p]; q < recOff[p + 1]; ++q) {
        const double* R = records + static_cast<size_t>(q) * recN;
        if (o < 49) {
          const int i = o / 7, j = o - 7 * i;   // pose_a[i] x pose_b[j]
          s += dir == 0 ? R[i * 16 + 7 + j] : R[j * 16 + 7 + i];
        } else if (o < 49 + 7 * G) {
          const int e = o - 49, j = e / G, v = e - j * G;  // theta_a[v] x pose_b[j]
          s += dir == 0 ? R[256 + (7 + j) * G + v] : R[256 + 15 * G + j * G + v];
        } else {
          const int e = o - 49 - 7 * G, i = e / G, v = e - i * G;  // pose_a[i] x theta_b[v]
          s += dir == 0 ? R[256 + 15 * G + i * G + v] : R[256 + (7 + i) * G + v];
        }
      }
    }
    if (o < 49) { row = o / 7; col = o % 7; }
    else if (o < 49 + 7 * G) { const int e = o - 49; col = e / G; row = 7 + (e - col * G); }
    else { const int e = o - 49 - 7 * G; row = e / G; col = 7 + (e - row * G); }
    Xp[static_cast<size_t>(row) * B + col] = s;
  }
}

// Grid x grid part of X_ab: sum over the pixels of both directions of gg fac-free tap products,
//   X[7 + v_a][7 + v_b] += gg * w_a[k] * w_b[l]        (gg = rho' JD_s,2 JD_t,2 d_s d_t of k_dense_walk; 0 = no constraint)
// One workgroup per (pair, panel of columns); lane = run of pixels as in the walk.
constexpr int kGgThreads = CVD_DETERMINISTIC ? 64 : 512;
template <int KD>
inline __global__ __launch_bounds__(kGgThreads) void k_dense_gg(Layout L, Table T, CrossPairs cp, const int* __restrict__ xDir,
                                                        const double* __restrict__ gg, int panelW, double* __restrict__ X) {
  static_assert(KD == 4, "bilinear depth grids");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD;
  const int pair = blockIdx.x, panel = blockIdx.y;
  const int v0 = panel * panelW, v1 = (v0 + panelW < G) ? v0 + panelW : G, pw = v1 - v0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kGgThreads / 64;
  double* GG = sm;
  for (int i = tid; i < G * panelW; i += kGgThreads) GG[i] = 0.0;
  __syncthreads();
  for (int dir = 0; dir < 2; ++dir) {
    const int p = xDir[pair * 2 + dir];
    if (p < 0) continue;
    const long long cb = T.pairOff[p], ce = T.pairOff[p + 1];
    constexpr long long kUnit = 64LL * kDwRun;
    for (long long u0 = cb + wave * kUnit; u0 < ce; u0 += NW * kUnit) {
      const long long cFirst = u0 + static_cast<long long>(lane) * kDwRun;
      const long long cStop = cFirst + kDwRun < ce ? cFirst + kDwRun : ce;
      for (long long c = cFirst; c < cStop; ++c) {
        const double g = gg[c];
        if (g == 0.0) continue;
        float4 nd;
        if (!denseNdcFromFlow(T, static_cast<int>(c - cb), T.flow[c], nd)) continue;
        FastTaps<KD> ts, tt;
        fastGather<KD>(L, nd.x, nd.y, ts);
        fastGather<KD>(L, nd.z, nd.w, tt);
        const FastTaps<KD>& tr = dir ? tt : ts;   // rows = frame a's vertices
        const FastTaps<KD>& tc = dir ? ts : tt;   // columns = frame b's
#pragma unroll
        for (int k = 0; k < KD; ++k) {
          const int ir = tr.I(k);
          const double fr = g * tr.Wt(k);
#pragma unroll
          for (int l = 0; l < KD; ++l) {
            const int jc = tc.I(l) - v0;
            if (jc >= 0 && jc < pw) atomicAdd(&GG[ir * panelW + jc], fr * tc.Wt(l));
          }
        }
      }
    }
  }
  __syncthreads();
  double* Xp = X + static_cast<size_t>(pair) * B * B;
  for (int i = tid; i < G * pw; i += kGgThreads) {
    const int r = i / pw, cidx = i - r * pw;
    Xp[static_cast<size_t>(7 + r) * B + 7 + v0 + cidx] = GG[r * panelW + cidx];
  }
}

}  // namespace cvd
