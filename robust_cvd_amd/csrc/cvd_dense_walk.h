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
//   k_dense_gg         per (undirected pair, panel of source vertices, direction): grid x grid of X_ab from the per-pixel scalars and the flow (taps only:
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
constexpr int kDwLd = 17;            // row stride (doubles) of a 16-column MFMA staging tile [64 constraints][16 columns] (k_coarse_edges_mfma)
constexpr int kDwLdF = 10;           // ... of the walk's 9-feature tile (conflict-free: 20 i mod 64 is injective over a 16-lane group); the
                                     // lanes of the seven unused columns read a zero word
constexpr int kDwBlk = 8;            // pixels of a lane's run that are loaded together (one 64 B line of flow)
constexpr int kDwIoStride = 9;       // per-run stride of the wave's input tiles [64 runs][8 pixels] (9 is odd: conflict-free by lane)
// input tiles of a wave: flow (float2, reused for the grid x grid scalars: double) | source depth | target depth | mask
constexpr int kDwIoBytes = 64 * kDwIoStride * (8 + 4 + 4 + 1);
// (dwRecordDoubles, DenseRecords: cvd_kernels.h, beside the frame-major kernel that folds the records)
constexpr int kDwCols = 15;          // side-block columns of a record

// Lane map of the walks.  A lane's LDS atomics go to the vertices of the cell its pixel lies in: lanes of one wave instruction that
// sit in the same cell hit the same addresses and are serialised (SQ_LDS_BANK_CONFLICT was 62 % of the LDS time with lanes 16
// pixels apart along a row and 2.7 rows per wave: ~4 lanes per cell of the 17 x 10 grid).  Here a wave works on `rowGroups` image
// rows that lie a band height (H / rowGroups) apart, `lanesPerRow` lanes per row, each walking its own run of `run` consecutive
// pixels: at 384 x 224 that is 16 x 4 lanes with runs of 24 pixels -- one lane per cell column, the row groups 56 rows (> 2 cells)
// apart.  A unit = one such set of rows; a pair has `bandH` units.
struct DenseLaneMap {
  int run, lanesPerRow, rowGroups, bandH;
};
inline DenseLaneMap denseLaneMap(int W, int H) {
  DenseLaneMap m;
  m.run = (W + 15) / 16;
  m.lanesPerRow = (W + m.run - 1) / m.run;
  m.rowGroups = 64 / m.lanesPerRow;
  if (m.rowGroups > H) m.rowGroups = H;
  m.bandH = (H + m.rowGroups - 1) / m.rowGroups;
  return m;
}
// first pixel of the lane's run in unit u (row-major index within the image) and the run's length (0: the lane idles)
__device__ __forceinline__ int denseLaneRunRC(const DenseLaneMap& m, int W, int H, int lane, int u, int& len, int& row, int& col) {
  const int rg = lane / m.lanesPerRow, cb = lane - rg * m.lanesPerRow;
  row = rg * m.bandH + u;
  col = cb * m.run;
  const int rowEnd = (rg + 1) * m.bandH < H ? (rg + 1) * m.bandH : H;
  len = (rg < m.rowGroups && row < rowEnd) ? (col + m.run <= W ? m.run : W - col) : 0;
  return row * W + col;
}
__device__ __forceinline__ int denseLaneRun(const DenseLaneMap& m, int W, int H, int lane, int u, int& len) {
  int row, col;
  return denseLaneRunRC(m, W, H, lane, u, len, row, col);
}

// Work list: one record per directed pair.
struct DenseWalkList {
  const int* pair;          // directed pair of the record
  int count;
  DenseLaneMap map;
};

// ndc of both end points of a dense-mode constraint from its flow vector, WITHOUT the depth fetch (k_dense_gg: taps only); false:
// no candidate.  (densePixelGeometry, cvd_kernels.h: the one statement of this arithmetic.)
__device__ __forceinline__ bool denseNdcFromFlow(const Table& T, int ix, int iy, float2 f, float4& n) {
  float4 loc;
  int ai, bi;
  return densePixelGeometry(T, ix, iy, f, loc, n, ai, bi);
}

// One constraint s -> t.  What stays in registers between its phases is the minimum the Jacobian can be re-expanded from:
//   d r / d q = M = [m00 0 m02; 0 m11 m12; 0 0 m22], the depth z' = -q_z, the residual, rho', d r_2 / d D_t, D_s and R_s c
// (16 doubles).  Rotation columns in cross-product form (dR/dw_i = [a_i]x R, FrameConst::Jl): d X / d w_s,i = a_i x y with
// y = D_s R_s c, d q / d w_t,i = R_t^T (v x a_i), v = X - t_t.
struct DwState {
  double m00, m02, m11, m12, m22;
  double zz;
  double r[3];
  double w, sw, rho0, JDT2, Da;   // rho', sqrt(rho'), rho
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

// The pixel whose source depth a constraint at pixel (ix, iy) reads (the reference's Observation constructor truncates the float
// location times the raster size, lib/PoseOptimizer.cpp:104-116: not always (ix, iy) itself), and the target's for flow f; -1: no candidate.
__device__ __forceinline__ int denseSourceDepthIndex(const Table& T, int ix, int iy) {
  return densePixelOfLoc(T, __fmul_rn(static_cast<float>(ix), T.sx), __fmul_rn(static_cast<float>(iy), T.sy));
}
__device__ __forceinline__ int denseTargetDepthIndex(const Table& T, int ix, int iy, float2 f) {
  float4 loc, n;
  int ai, bi;
  return densePixelGeometry(T, ix, iy, f, loc, n, ai, bi) ? bi : -1;
}

// Constants of the directed pair, wave-uniform (SGPRs: a VALU instruction takes one scalar operand).
struct DwPairConst {
  double Rs[9], Rt[9];
  double dT[3];            // t_s - t_t
  double fxs, fys, ifys;   // source focal lengths
  double ifxt, ifyt;       // 1 / target focal lengths
};

__device__ __forceinline__ void dwChain(const Layout& L, const DwPairConst& P, const double* __restrict__ xs,
                                        const double* __restrict__ xt, const float4& nd, double da, double db, const DwTaps& ts,
                                        const DwTaps& tt, DwState& o) {
  constexpr double eps = 1e-6;
  const int gx = L.gx, gMax = L.nD - 1;
  // (the same tap order and summation order as fastGather / the other fast kernels)
  double Da = 0.0, Db = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    Da += da * xs[7 + min(ts.i0 + (k & 1) + (k >> 1) * gx, gMax)] * ts.Wt(k);   // (clamped: the 1 x 1 grid of a Global transform)
    Db += db * xt[7 + min(tt.i0 + (k & 1) + (k >> 1) * gx, gMax)] * tt.Wt(k);
  }
  o.Da = Da;
  const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
  const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
  const double ca[3] = {pax * P.fxs, pay * P.fys, -1.0};
  double v[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o.Rca[i] = dot3(P.Rs + 3 * i, ca);
    v[i] = P.dT[i] + o.Rca[i] * Da;
  }
  const double q0 = P.Rt[0] * v[0] + P.Rt[3] * v[1] + P.Rt[6] * v[2];
  const double q1 = P.Rt[1] * v[0] + P.Rt[4] * v[1] + P.Rt[7] * v[2];
  const double q2 = P.Rt[2] * v[0] + P.Rt[5] * v[1] + P.Rt[8] * v[2];
  const double zz = -q2;
  const double iz = rcpFast(zz);   // (1 - 2 ulp: cvd_kernels.h; the IEEE division sequence is twice the instructions)
  const double u = q0 * iz * P.ifxt;
  const double vv = q1 * iz * P.ifyt;
  o.zz = zz;
  o.r[0] = (u - pbx) * L.ws;
  o.r[1] = (vv - pby) * L.ws;
  double dr2dA, dr2dDb;
  if (L.lossType == kLossDisparity) {
    const bool zo = !(zz < eps), bo = !(Db < eps);
    const double izc = zo ? iz : 1.0 / eps, ibc = rcpFast(bo ? Db : eps);
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
  {
    const double sq = o.r[0] * o.r[0] + o.r[1] * o.r[1] + o.r[2] * o.r[2];
    if (L.robustKind == kRobustCauchy) {   // (robustRho with the reciprocal and the square root of rho' in their fast forms)
      const double sum = 1.0 + sq * L.cauchyC;
      o.rho0 = L.cauchyB * log(sum);
      o.sw = rsqrtFast(sum);
      o.w = o.sw * o.sw;
    } else {
      robustRho(L, sq, o.rho0, o.w);
      o.sw = sqrt(o.w);
    }
  }
  const double wiz = L.ws * iz;
  o.m00 = wiz * P.ifxt;
  o.m11 = wiz * P.ifyt;
  o.m02 = wiz * u;
  o.m12 = wiz * vv;
  o.m22 = -dr2dA;
  o.JDT2 = dr2dDb;
}

// The Jacobian row that d r / d q = mu stands for, in FEATURE form.  With g = R_t mu, y = D_s R_s c, n = y x g, v = dT + y the 15
// columns [Jp_s (7) | Jp_t (7) | r] are
//   J[0..2] = g, J[3 + i] = a_s,i . n, J[6] = (g . y + D_s g . R_s e_z) / fy_s, J[7..9] = -g,
//   J[10 + i] = a_t,i . (g x v) = (dT x a_t,i) . g - a_t,i . n, J[13] = d r / d fy_t, J[14] = r:
// linear, with coefficients that are constants of the frame pair, in the NINE features  g (3) | n (3) | J[6] | J[13] | J[14].
// The matrix pipe accumulates the Gram matrix of the features; the workgroup maps it to the 15 columns once, at the end
// (dwFeatureMap) -- 9 staged values per row instead of 15, and the rotation Jacobians a_i never enter the pixel loop.
constexpr int kDwFeat = 9;
__device__ __forceinline__ void dwFeatures(const DwState& c, const DwPairConst& P, const double (&y)[3], double mu0, double mu1,
                                           double mu2, double j13, double rr, double (&F)[kDwFeat], double& jds) {
  double g[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) g[i] = P.Rt[3 * i] * mu0 + P.Rt[3 * i + 1] * mu1 + P.Rt[3 * i + 2] * mu2;
  F[0] = g[0]; F[1] = g[1]; F[2] = g[2];
  F[3] = y[1] * g[2] - y[2] * g[1];
  F[4] = y[2] * g[0] - y[0] * g[2];
  F[5] = y[0] * g[1] - y[1] * g[0];
  jds = dot3(g, c.Rca);
  F[6] = (c.Da * P.ifys) * (jds + g[0] * P.Rs[2] + g[1] * P.Rs[5] + g[2] * P.Rs[8]);
  F[7] = j13;
  F[8] = rr;
}
// column n of the 15 as a combination of the 9 features: coefficient of feature a
__device__ __forceinline__ double dwFeatureMap(const FrameConst& Fs, const FrameConst& Ft, const double (&dT)[3], int a, int n) {
  if (n < 3) return a == n ? 1.0 : 0.0;
  if (n < 6) return (a >= 3 && a < 6) ? Fs.Jl[3 * (n - 3) + (a - 3)] : 0.0;
  if (n == 6) return a == 6 ? 1.0 : 0.0;
  if (n < 10) return a == n - 7 ? -1.0 : 0.0;
  if (n < 13) {
    const double* at = Ft.Jl + 3 * (n - 10);
    if (a < 3) {  // (dT x a_t)[a]
      const int i1 = (a + 1) % 3, i2 = (a + 2) % 3;
      return dT[i1] * at[i2] - dT[i2] * at[i1];
    }
    return a < 6 ? -at[a - 3] : 0.0;
  }
  if (n == 13) return a == 7 ? 1.0 : 0.0;
  if (n == 14) return a == 8 ? 1.0 : 0.0;
  return 0.0;
}

#ifdef CVD_DW_PROFILE
__device__ unsigned long long g_dwProf[4096 * 8];
#define DW_STAMP(slot) do { if (threadIdx.x == 0) g_dwProf[(blockIdx.x & 4095) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DW_STAMP(slot) do {} while (0)
#endif

// LDS accumulators of k_dense_walk (doubles): the Gram tile, then MOMENT blocks instead of the 15-column side blocks.  The columns
// [Jp_s | Jp_t | r] contracted with a vector mu are linear in a few per-pixel FEATURES with coefficients that are constants of the
// frame pair (dwColumns spelled out):
//   g = R_t mu, y = D_s R_s c, n = y x g:   J[0..2] = g, J[3 + i] = a_s,i . n, J[6] = (g . y + D_s g . R_s e_z) / fy_s, J[7..9] = -g,
//                                           J[10 + i] = a_t,i . (g x v) = g . (dT x a_t,i) - a_t,i . n   (v = dT + y), J[13], J[14]
//   side S (mu = sum_r rho' (d r_r / d D_s) M_r, any direction):  the 9 features of dwFeatures
//   side T (mu = rho' (d r_2 / d D_t) m22 e_z, i.e. g = s rho with rho = R_t e_z a constant of the pair):
//                                                                  6 features   s | s y (3) | s D_s | gradient
// so a pixel sends 6 x 4 atomics to its target vertices instead of 14 x 4, and the source side -- whose cell and vertical tap
// weights are the same along a lane's run of pixels of one image row -- accumulates 2 x 9 feature sums in REGISTERS and flushes
// them once per run (36 atomics per run instead of 60 per pixel).  The workgroup expands the moments into the record's side blocks
// when it writes the record.
constexpr int kDwFeatS = kDwFeat, kDwFeatT = 6;
__host__ __device__ inline int dwAccDoubles(int G) { return 256 + (kDwFeatS + kDwFeatT + 10) * G + 8; }
__host__ __device__ inline size_t dwLdsBytes(int G, int B, int threads) {
  return (static_cast<size_t>(dwAccDoubles(G)) + 2 * B + 2 + static_cast<size_t>(threads / 64) * 64 * kDwLdF) * 8 +
         static_cast<size_t>(threads / 64) * ((kDwIoBytes + 15) / 16 * 16);
}

#ifndef DW_WAVES
#define DW_WAVES 2
#endif
template <int KD>
inline __global__ __launch_bounds__(kDwThreads) __attribute__((amdgpu_waves_per_eu(DW_WAVES))) void k_dense_walk(Layout L, Table T, DenseWalkList wl, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc, double* __restrict__ records,
                                                           double* __restrict__ ggOut) {
  static_assert(KD == 4, "bilinear depth grids (the explicit-block scope of the dense mode)");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B, G = L.nD, gx = L.gx;
  // A Global depth transform (one value parameter per frame) is walked as a 1 x 1 grid: dwGather returns vertex 0 with weight one
  // and three taps of weight exactly zero that do not exist -- nothing is accumulated for them and the chain's gathers clamp their index.
  const int nTap = G == 1 ? 1 : 4;
  const int accN = dwAccDoubles(G);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = kDwThreads / 64;
  double* acc = sm;                       // [0, 256) Gram tile
  double* MS = acc + 256;                 // [12][G] source-side moments
  double* MT = MS + kDwFeatS * G;         // [6][G]  target-side moments
  double* bandS = MT + kDwFeatT * G;      // [5][G]
  double* bandT = bandS + 5 * G;          // [5][G], then the cost
  double* xs = acc + accN;
  double* xt = xs + B;
  double* zeroWord = xt + B;                                   // what the lanes of the unused MFMA columns read
  double* scr = zeroWord + 2 + wave * (64 * kDwLdF);
  // the wave's input tiles [64 runs][8 pixels] (stride 9): one 64 B line of flow per run and block, loaded by eight adjacent lanes
  unsigned char* ioBase = reinterpret_cast<unsigned char*>(zeroWord + 2 + NW * (64 * kDwLdF)) + wave * ((kDwIoBytes + 15) / 16 * 16);
  float2* ioF = reinterpret_cast<float2*>(ioBase);             // flow; after a pixel's trip: its grid x grid scalar (double)
  double* ioG = reinterpret_cast<double*>(ioBase);
  float* ioDa = reinterpret_cast<float*>(ioBase + 64 * kDwIoStride * 8);
  float* ioDb = ioDa + 64 * kDwIoStride;
  unsigned char* ioM = reinterpret_cast<unsigned char*>(ioDb + 64 * kDwIoStride);
  const int rec = blockIdx.x;
  const int p = wl.pair[rec];
  const int fs = T.pairA[p], ft = T.pairB[p];
  DW_STAMP(0);
  for (int i = tid; i < accN; i += kDwThreads) acc[i] = 0.0;
  for (int i = tid; i < B; i += kDwThreads) {
    xs[i] = x[static_cast<size_t>(fs) * B + i];
    xt[i] = x[static_cast<size_t>(ft) * B + i];
  }
  if (tid < 2) zeroWord[tid] = 0.0;
  __syncthreads();
  DW_STAMP(1);
  // (wave-uniform read-only global data at an address that depends on blockIdx only: scalar loads, as in the hot product)
  const FrameConst& Fs = fc[fs];
  const FrameConst& Ft = fc[ft];
  DwPairConst P;
#pragma unroll
  for (int i = 0; i < 9; ++i) { P.Rs[i] = uniformValue(Fs.R[i]); P.Rt[i] = uniformValue(Ft.R[i]); }
#pragma unroll
  for (int i = 0; i < 3; ++i) P.dT[i] = uniformValue(Fs.t[i] - Ft.t[i]);
  P.fys = uniformValue(Fs.fy);
  P.fxs = uniformValue(Fs.fy * L.aspect);
  P.ifys = uniformValue(1.0 / Fs.fy);
  P.ifyt = uniformValue(1.0 / Ft.fy);
  P.ifxt = uniformValue(1.0 / (Ft.fy * L.aspect));
  const long long pixBase = T.pairOff[p];
  const bool havePixels = T.pairOff[p + 1] > pixBase;

  cvd_d4 tile0 = {0.0, 0.0, 0.0, 0.0};
  double cost = 0.0;
  const int mk = lane >> 4, mc = lane & 15;  // MFMA operand element [k = lane >> 4][column = lane & 15]
  const DenseLaneMap map = wl.map;
  const size_t npxImg = static_cast<size_t>(T.W) * T.H;
  const float* depthS = T.depth + static_cast<size_t>(fs) * npxImg;
  const float* depthT = T.depth + static_cast<size_t>(ft) * npxImg;
  const int mfmaRead = mc < kDwFeat ? mk * kDwLdF + mc : -1;   // (columns 9..15 of the Gram tile's operand are zero)
  for (int u = wave; u < (havePixels ? map.bandH : 0); u += NW) {
    int len, iy, ix0;
    const int iFirst = denseLaneRunRC(map, T.W, T.H, lane, u, len, iy, ix0);
    // source-side sums of the lane's run (one image row: one cell row, one pair of vertical tap weights)
    double AS[2][kDwFeatS], BS[3];
#pragma unroll
    for (int f = 0; f < kDwFeatS; ++f) { AS[0][f] = 0.0; AS[1][f] = 0.0; }
    BS[0] = BS[1] = BS[2] = 0.0;
    int curI0 = -1;
    double curRy = 0.0;
    // (one trip past the run: the flush of the last cell's sums has ONE copy, at the top of the trip)
#pragma unroll 1
    for (int t = 0; t <= map.run; ++t) {
      const int tb = t & (kDwBlk - 1);
      if (tb == 0 && t < map.run) {
        // ---- the next eight pixels of all 64 runs of the wave, by line: eight adjacent lanes load one run's 64 B of flow, its 32 B
        // of source depth, its 8 mask bytes into the wave's input tiles; then every lane fetches the target depths of its own eight
        // pixels, whose addresses follow from the flow.  (A lane that walked its own run alone touched every line in eight trips,
        // 20 k cycles apart: 18 GB fetched per launch against 2.6 GB of images, 7.7 GB written against 1.2.)
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
          const int run = (lane >> 3) + 8 * q, px = lane & 7;
          const int firstR = __shfl(iFirst, run), lenR = __shfl(len, run), rowR = __shfl(iy, run), colR = __shfl(ix0, run);
          float2 f = make_float2(0.f, 0.f);
          float da = 0.f;
          unsigned char m = 0;
          if (t + px < lenR) {
            const long long c = pixBase + firstR + t + px;
            f = T.flow[c];
            m = T.fmask[c];
            da = depthS[denseSourceDepthIndex(T, colR + t + px, rowR)];
          }
          ioF[run * kDwIoStride + px] = f;
          ioDa[run * kDwIoStride + px] = da;
          ioM[run * kDwIoStride + px] = m;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int h = 0; h < kDwBlk; h += 4) {
          float dbv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int px = h + k;
            dbv[k] = 0.f;
            if (t + px < len && ioM[lane * kDwIoStride + px]) {
              const int at = denseTargetDepthIndex(T, ix0 + t + px, iy, ioF[lane * kDwIoStride + px]);
              if (at >= 0) dbv[k] = depthT[at];
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) ioDb[lane * kDwIoStride + h + k] = dbv[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      const int i = iFirst + t;
      const bool inRun = t < len;
      // the source end point's cell follows from the pixel alone (the float arithmetic of denseConstraintFromFlow)
      DwTaps ts;
      ts.i0 = -2; ts.rx = 0.0; ts.ry = 0.0;
      if (inRun) {
        const float lx0 = __fmul_rn(static_cast<float>(ix0 + t), T.sx), ly0 = __fmul_rn(static_cast<float>(iy), T.sy);
        dwGather(L, __fadd_rn(-1.f, __fmul_rn(2.f, lx0)), __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly0), T.invAspect)), ts);
      }
      if (curI0 >= 0 && ts.i0 != curI0) {
        // ---- flush the sums of the cell the lane has left: 4 taps x 9 features, 10 vertex pairs
        const double wy0 = 1.0 - curRy, wy1 = curRy;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k >= nTap) continue;   // (uniform: a Global transform is ONE vertex, its cell's other three taps do not exist)
          double* col = MS + curI0 + (k & 1) + (k >> 1) * gx;
          const double wy = (k >> 1) ? wy1 : wy0;
#pragma unroll
          for (int f = 0; f < kDwFeatS; ++f) atomicAdd(&col[f * G], AS[k & 1][f] * wy);
        }
        // vertex pairs of the cell, lower vertex first: offsets {0, 1, gx - 1, gx, gx + 1} -> band index {0, 1, 2, 3, 4}
        const double y00 = wy0 * wy0, y01 = wy0 * wy1, y11 = wy1 * wy1;
        atomicAdd(&bandS[0 * G + curI0], BS[0] * y00);
        if (nTap == 4) {
          atomicAdd(&bandS[1 * G + curI0], BS[1] * y00);
          atomicAdd(&bandS[3 * G + curI0], BS[0] * y01);
          atomicAdd(&bandS[4 * G + curI0], BS[1] * y01);
          atomicAdd(&bandS[0 * G + curI0 + 1], BS[2] * y00);
          atomicAdd(&bandS[2 * G + curI0 + 1], BS[1] * y01);
          atomicAdd(&bandS[3 * G + curI0 + 1], BS[2] * y01);
          atomicAdd(&bandS[0 * G + curI0 + gx], BS[0] * y11);
          atomicAdd(&bandS[1 * G + curI0 + gx], BS[1] * y11);
          atomicAdd(&bandS[0 * G + curI0 + gx + 1], BS[2] * y11);
        }
#pragma unroll
        for (int f = 0; f < kDwFeatS; ++f) { AS[0][f] = 0.0; AS[1][f] = 0.0; }
        BS[0] = BS[1] = BS[2] = 0.0;
        curI0 = -1;
      }
      if (t == map.run) break;   // (the extra trip only flushes)
      // pixel t: flow, mask and both depths from the wave's input tiles
      const float2 fl = ioF[lane * kDwIoStride + tb];
      const float2 d = make_float2(ioDa[lane * kDwIoStride + tb], ioDb[lane * kDwIoStride + tb]);
      float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool valid = inRun && ioM[lane * kDwIoStride + tb] != 0 && isfinite(d.x) && d.x > 0.f && isfinite(d.y) && d.y > 0.f &&
                         denseNdcFromFlow(T, ix0 + t, iy, fl, nd);
      // (this block's scalars go out together, by line, when its last pixel is done)
      const bool lastOfBlock = tb == kDwBlk - 1 || t == map.run - 1;
      auto storeScalars = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int t0 = t - tb;
#pragma unroll 2
        for (int q = 0; q < 8; ++q) {
          const int run = (lane >> 3) + 8 * q, px = lane & 7;
          const int firstR = __shfl(iFirst, run), lenR = __shfl(len, run);
          if (t0 + px < lenR) ggOut[pixBase + firstR + t0 + px] = ioG[run * kDwIoStride + px];
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      };
      if (__builtin_amdgcn_readfirstlane(__ballot(valid) == 0ull ? 1 : 0)) {
        ioG[lane * kDwIoStride + tb] = 0.0;
        if (lastOfBlock) storeScalars();
        continue;  // (wave-uniform)
      }
      DwState ch;
      double sw = 0.0, zf = 0.0;
      double y[3] = {0.0, 0.0, 0.0};
      if (valid) {
        DwTaps tt;
        const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
        dwGather(L, nd.z, nd.w, tt);
        dwChain(L, P, xs, xt, nd, da, db, ts, tt, ch);
        sw = ch.sw;
        cost += ch.rho0;
        zf = -ch.zz * P.ifyt;   // d r_0,1 / d fy_t = -ws (u, v) / fy_t = -(m02, m12) z' / fy_t
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) y[i2] = ch.Rca[i2] * ch.Da;
        // rho' d r_r / d D_s = rho' M_r . (R_t^T R_s c)
        const double k0 = P.Rt[0] * ch.Rca[0] + P.Rt[3] * ch.Rca[1] + P.Rt[6] * ch.Rca[2];
        const double k1 = P.Rt[1] * ch.Rca[0] + P.Rt[4] * ch.Rca[1] + P.Rt[7] * ch.Rca[2];
        const double k2 = P.Rt[2] * ch.Rca[0] + P.Rt[5] * ch.Rca[1] + P.Rt[8] * ch.Rca[2];
        const double wj0 = ch.w * (ch.m00 * k0 + ch.m02 * k2), wj1 = ch.w * (ch.m11 * k1 + ch.m12 * k2), wj2 = ch.w * ch.m22 * k2;
        const double wt = ch.w * ch.JDT2;   // rho' d r_2 / d D_t
        ioG[lane * kDwIoStride + tb] = wj2 * ch.JDT2 * da * db;
        {  // ---- source side: features of mu = sum_r wj_r M_r into the run's sums
          const double mu0 = wj0 * ch.m00, mu1 = wj1 * ch.m11, mu2 = wj0 * ch.m02 + wj1 * ch.m12 + wj2 * ch.m22;
          double ft12[kDwFeatS], sSS;   // sSS = sum_r rho' (d r_r / d D_s)^2
          dwFeatures(ch, P, y, mu0, mu1, mu2, (wj0 * ch.m02 + wj1 * ch.m12) * zf, wj0 * ch.r[0] + wj1 * ch.r[1] + wj2 * ch.r[2], ft12, sSS);
          curI0 = ts.i0;
          curRy = ts.ry;
          const double fa0 = da * (1.0 - ts.rx), fa1 = da * ts.rx;
#pragma unroll
          for (int f = 0; f < kDwFeatS; ++f) {
            AS[0][f] += ft12[f] * fa0;
            AS[1][f] += ft12[f] * fa1;
          }
          BS[0] += sSS * fa0 * fa0;
          BS[1] += sSS * fa0 * fa1;
          BS[2] += sSS * fa1 * fa1;
        }
        {  // ---- target side: d r / d D_t has the one row 2, g = s R_t e_z
          const double s = wt * ch.m22;
          const double mt[kDwFeatT] = {s, s * y[0], s * y[1], s * y[2], s * ch.Da, wt * ch.r[2]};
          double fb[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) fb[k] = tt.Wt(k) * db;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            if (k >= nTap) continue;
            double* col = MT + tt.i0 + (k & 1) + (k >> 1) * gx;
#pragma unroll
            for (int f = 0; f < kDwFeatT; ++f) atomicAdd(&col[f * G], mt[f] * fb[k]);
          }
          const double sTT = wt * ch.JDT2;
          const int j0 = tt.i0;
          const double b0 = sTT * fb[0], b1 = sTT * fb[1], b2 = sTT * fb[2], b3 = sTT * fb[3];
          atomicAdd(&bandT[0 * G + j0], b0 * fb[0]);
          if (nTap == 4) {
            atomicAdd(&bandT[1 * G + j0], b0 * fb[1]);
            atomicAdd(&bandT[3 * G + j0], b0 * fb[2]);
            atomicAdd(&bandT[4 * G + j0], b0 * fb[3]);
            atomicAdd(&bandT[0 * G + j0 + 1], b1 * fb[1]);
            atomicAdd(&bandT[2 * G + j0 + 1], b1 * fb[2]);
            atomicAdd(&bandT[3 * G + j0 + 1], b1 * fb[3]);
            atomicAdd(&bandT[0 * G + j0 + gx], b2 * fb[2]);
            atomicAdd(&bandT[1 * G + j0 + gx], b2 * fb[3]);
            atomicAdd(&bandT[0 * G + j0 + gx + 1], b3 * fb[3]);
          }
        }
      } else {
        ioG[lane * kDwIoStride + tb] = 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- the three residual rows sqrt(rho') x features through the wave's staging tile into the Gram tile.  A REAL loop: unrolled,
      // the three rows' features are all formed before the first one is staged (their arithmetic moves freely across the fences).
#pragma unroll 1
      for (int r = 0; r < 3; ++r) {
        if (valid) {
          const double mu0 = r == 0 ? ch.m00 : 0.0, mu1 = r == 1 ? ch.m11 : 0.0;
          const double mu2 = r == 0 ? ch.m02 : (r == 1 ? ch.m12 : ch.m22);
          const double j13 = r == 2 ? 0.0 : mu2 * zf;
          const double rr = r == 0 ? ch.r[0] : (r == 1 ? ch.r[1] : ch.r[2]);
          double F[kDwFeat], jds;
          dwFeatures(ch, P, y, mu0, mu1, mu2, j13, rr, F, jds);
#pragma unroll
          for (int c = 0; c < kDwFeat; ++c) scr[lane * kDwLdF + c] = sw * F[c];
        } else {
#pragma unroll
          for (int c = 0; c < kDwFeat; ++c) scr[lane * kDwLdF + c] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const double* rd = mfmaRead >= 0 ? scr + mfmaRead : zeroWord;
        const int rstep = mfmaRead >= 0 ? 4 * kDwLdF : 0;
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
          const double a0 = rd[j * rstep];
          const double a1 = rd[(j + 1) * rstep];
          tile0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, tile0, 0, 0, 0);
          tile0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, tile0, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (lastOfBlock) storeScalars();
    }
  }
  DW_STAMP(2);
  // ---- the waves' Gram tiles and cost (D: column = lane & 15, row = (lane >> 4) + 4 reg)
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicAdd(&acc[((lane >> 4) + 4 * q) * 16 + (lane & 15)], tile0[q]);
  cost = waveSum(cost);
  if (lane == 0) atomicAdd(&bandT[5 * G], cost);
  __syncthreads();
  // ---- the record: tile, the side blocks expanded from the moments with the pair's constants, bands, cost
  const int recN = dwRecordDoubles(G);
  double* out = records + static_cast<size_t>(rec) * recN;
  const double rho[3] = {Ft.R[2], Ft.R[5], Ft.R[8]};          // R_t e_z
  const double rs2[3] = {Fs.R[2], Fs.R[5], Fs.R[8]};          // R_s e_z
  const double dT[3] = {Fs.t[0] - Ft.t[0], Fs.t[1] - Ft.t[1], Fs.t[2] - Ft.t[2]};
  const double ifys = 1.0 / Fs.fy;
  for (int i = tid; i < 256; i += kDwThreads) {
    // D[m][n] = sum_ab C[a][m] Gram9[a][b] C[b][n]
    const int m = i >> 4, n = i & 15;
    double v = 0.0;
    if (m < 15 && n < 15) {
      for (int a = 0; a < kDwFeat; ++a) {
        const double cm = dwFeatureMap(Fs, Ft, dT, a, m);
        if (cm == 0.0) continue;
        double row = 0.0;
        for (int b = 0; b < kDwFeat; ++b) row += acc[a * 16 + b] * dwFeatureMap(Fs, Ft, dT, b, n);
        v += cm * row;
      }
    }
    out[i] = v;
  }
  for (int i = tid; i < 10 * G; i += kDwThreads) out[256 + 30 * G + i] = bandS[i];
  if (tid == 0) out[256 + 40 * G] = bandT[5 * G];
  for (int idx = tid; idx < 2 * kDwCols * G; idx += kDwThreads) {
    const int side = idx >= kDwCols * G ? 1 : 0;
    const int e = idx - side * kDwCols * G;
    const int c = e / G, v = e - c * G;
    double val;
    if (side == 0) {
      const double* m = MS + v;   // m[f * G]
      val = 0.0;
      for (int a = 0; a < kDwFeat; ++a) val += dwFeatureMap(Fs, Ft, dT, a, c) * m[a * G];
    } else {
      const double* m = MT + v;
      if (c < 3) val = rho[c] * m[0];
      else if (c < 6) {
        const double* a = Fs.Jl + 3 * (c - 3);
        const double b0 = rho[1] * a[2] - rho[2] * a[1], b1 = rho[2] * a[0] - rho[0] * a[2], b2 = rho[0] * a[1] - rho[1] * a[0];  // rho x a
        val = b0 * m[G] + b1 * m[2 * G] + b2 * m[3 * G];
      } else if (c == 6) val = (rho[0] * m[G] + rho[1] * m[2 * G] + rho[2] * m[3 * G] + dot3(rho, rs2) * m[4 * G]) * ifys;
      else if (c < 10) val = -rho[c - 7] * m[0];
      else if (c < 13) {
        const double* a = Ft.Jl + 3 * (c - 10);
        const double b0 = a[1] * rho[2] - a[2] * rho[1], b1 = a[2] * rho[0] - a[0] * rho[2], b2 = a[0] * rho[1] - a[1] * rho[0];  // a x rho
        const double q0 = rho[1] * dT[2] - rho[2] * dT[1], q1 = rho[2] * dT[0] - rho[0] * dT[2], q2 = rho[0] * dT[1] - rho[1] * dT[0];  // rho x dT
        val = (a[0] * q0 + a[1] * q1 + a[2] * q2) * m[0] + b0 * m[G] + b1 * m[2 * G] + b2 * m[3 * G];
      } else if (c == 13) val = 0.0;
      else val = m[5 * G];
    }
    out[256 + idx] = val;
  }
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

// Grid x grid part of X_ab: sum over the pixels of both directions of tap products,
//   X[7 + v_a][7 + v_b] += gg * w_a[k] * w_b[l]        (gg = rho' JD_s,2 JD_t,2 d_s d_t of k_dense_walk; 0 = no constraint)
// G^2 doubles do not fit the LDS: the block is built in panels of SOURCE vertices -- rows of X for direction a -> b, columns for
// b -> a, one launch per direction (the second adds).  A pixel's source taps follow from its position alone, so a panel reads only the
// image rows whose cell row touches it: every pixel is visited once (the rows of the one cell row two panels share: twice), where
// panels of target vertices -- rounds 6a -- walked all pixels of both directions for every panel.  One workgroup per (pair, panel);
// lane = run of pixels as in the walk, over the panel's rows.
constexpr int kGgThreads = CVD_DETERMINISTIC ? 64 : 1024;
template <int KD>
inline __global__ __launch_bounds__(kGgThreads) void k_dense_gg(Layout L, Table T, CrossPairs cp, const int* __restrict__ xDir,
                                                        const double* __restrict__ gg, int panelW, int dir, double* __restrict__ X) {
  static_assert(KD == 4, "bilinear depth grids");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int yRange[2];
  const int B = L.B, G = L.nD;
  const int pair = blockIdx.x, panel = blockIdx.y;
  const int v0 = panel * panelW, v1 = (v0 + panelW < G) ? v0 + panelW : G, pw = v1 - v0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kGgThreads / 64;
  double* GG = sm;   // [source vertex - v0][target vertex]
  for (int i = tid; i < G * pw; i += kGgThreads) GG[i] = 0.0;
  if (tid == 0) { yRange[0] = T.H; yRange[1] = -1; }
  __syncthreads();
  const int p = xDir[pair * 2 + dir];
  const long long cb = p >= 0 ? T.pairOff[p] : 0;
  const bool havePixels = p >= 0 && T.pairOff[p + 1] > cb;
  if (!havePixels && dir != 0) return;   // (uniform; the first direction's launch writes the block even when it is zero)
  if (havePixels) {
    // the image rows whose source cell row has a vertex in [v0, v1) (cell row cy: vertices [cy gx, (cy + 2) gx))
    const int cyLo = (v0 + L.gx) / L.gx - 2, cyHi = (v1 - 1) / L.gx;
    for (int y = tid; y < T.H; y += kGgThreads) {
      const float ly0 = __fmul_rn(static_cast<float>(y), T.sy);
      const float ny = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly0), T.invAspect));
      int cy;
      double ry;
      gridCellFast(ny, 0.5 * static_cast<double>(L.gy - 1), L.maxcy, cy, ry);
      if (cy >= cyLo && cy <= cyHi) {
        atomicMin(&yRange[0], y);
        atomicMax(&yRange[1], y);
      }
    }
  }
  __syncthreads();
  const int yFirst = yRange[0], Hs = yRange[1] - yRange[0] + 1;
  if (havePixels && Hs > 0) {
    DenseLaneMap map;   // (denseLaneMap of the panel's rows)
    map.run = (T.W + 15) / 16;
    map.lanesPerRow = (T.W + map.run - 1) / map.run;
    map.rowGroups = 64 / map.lanesPerRow < Hs ? 64 / map.lanesPerRow : Hs;
    map.bandH = (Hs + map.rowGroups - 1) / map.rowGroups;
    for (int u = wave; u < map.bandH; u += NW) {
      int len, row, ix0;
      (void)denseLaneRunRC(map, T.W, Hs, lane, u, len, row, ix0);
      const int iy = yFirst + row;
      const long long cFirst = cb + static_cast<long long>(iy) * T.W + ix0;
      // (gg and flow of the lane's next FOUR pixels in flight: the trip is short and the kernel waits on its loads otherwise)
      constexpr int kBatch = 4;
      double gN[kBatch];
      float2 fN[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        gN[q] = 0.0;
        fN[q] = make_float2(0.f, 0.f);
        if (q < len) { gN[q] = gg[cFirst + q]; fN[q] = T.flow[cFirst + q]; }
      }
      // The 4 x 4 tap products are summed in registers while BOTH end points stay in their cells (a lane's run is one cell wide and
      // the flow is smooth: a few flushes per run instead of 16 atomics per pixel).
      double acc[KD][KD];
      int curS = -1, curT = -1;   // first vertex of the source / target cell the sums belong to
      auto flush = [&]() {
        if (curS < 0) return;
#pragma unroll
        for (int k = 0; k < KD; ++k) {
          const int is = curS + (k & 1) + (k >> 1) * L.gx - v0;
          if (is < 0 || is >= pw) continue;
#pragma unroll
          for (int l = 0; l < KD; ++l) {
            const int it = curT + (l & 1) + (l >> 1) * L.gx;
            if (it < G) atomicAdd(&GG[is * G + it], acc[k][l]);   // (a Global transform's 1 x 1 grid: only tap 0 exists)
          }
        }
      };
      for (int t0 = 0; t0 < len; t0 += kBatch) {
        double gC[kBatch];
        float2 fC[kBatch];
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
          gC[q] = gN[q];
          fC[q] = fN[q];
          gN[q] = 0.0;
          if (t0 + kBatch + q < len) { gN[q] = gg[cFirst + t0 + kBatch + q]; fN[q] = T.flow[cFirst + t0 + kBatch + q]; }
        }
#pragma unroll
        for (int q = 0; q < kBatch; ++q) {
          const int t = t0 + q;
          const double g = gC[q];
          const float2 f = fC[q];
          if (g == 0.0) continue;
          float4 nd;
          if (!denseNdcFromFlow(T, ix0 + t, iy, f, nd)) continue;
          FastTaps<KD> ts, tt;
          fastGather<KD>(L, nd.x, nd.y, ts);
          fastGather<KD>(L, nd.z, nd.w, tt);
          if (ts.I(0) != curS || tt.I(0) != curT) {
            flush();
            curS = ts.I(0);
            curT = tt.I(0);
#pragma unroll
            for (int k = 0; k < KD; ++k)
#pragma unroll
              for (int l = 0; l < KD; ++l) acc[k][l] = 0.0;
          }
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            const double fr = g * ts.Wt(k);
#pragma unroll
            for (int l = 0; l < KD; ++l) acc[k][l] += fr * tt.Wt(l);
          }
        }
      }
      flush();
    }
  }
  __syncthreads();
  double* Xp = X + static_cast<size_t>(pair) * B * B;
  if (dir == 0) {   // rows = frame a's vertices = this direction's sources
    for (int i = tid; i < pw * G; i += kGgThreads) {
      const int is = i / G, it = i - is * G;
      Xp[static_cast<size_t>(7 + v0 + is) * B + 7 + it] = GG[i];
    }
  } else {          // columns = frame b's vertices = this direction's sources; added to what the first launch wrote
    for (int i = tid; i < pw * G; i += kGgThreads) {
      const int it = i / pw, is = i - it * pw;
      Xp[static_cast<size_t>(7 + it) * B + 7 + v0 + is] += GG[is * G + it];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Dense mode OUTSIDE the scope of the kernels above (spatial transforms, Shared intrinsics, ScaleShift, bicubic grids, smoothness
// triplets, the pair loop of normalizeDepth): the constraint LIST the images stand for -- what the reference's
// FlowConstraintsCollection::compute builds with matchSeparation = 0 (lib/FlowConstraints.cpp:436-460: mask, target int(x + flow + 0.5)
// in bounds; scaling :371) -- is materialised ON THE DEVICE, in row-major pixel order per pair, and the solve runs on the list-mode
// kernels (every residual configuration).  24 B per constraint of table + 20 B of list: 6.4 GB for the 144 M constraints of the
// benchmarked video -- what 288 GB of HBM are for; until round 5 such a solve was refused and lib_python built the list on the host.
constexpr int kDlChunk = 4096;   // pixels per workgroup
__device__ __forceinline__ bool denseCandidate(const Table& T, int pix, unsigned int m, float2 f, float4& loc) {
  if (!m) return false;
  const int iy = pix / T.W, ix = pix - iy * T.W;
  float4 n;
  int ai, bi;
  return densePixelGeometry(T, ix, iy, f, loc, n, ai, bi);
}
// pass 1: candidates per (pair, chunk); pass 2 (offsets != nullptr): write them at the chunk's offset in pixel order
inline __global__ __launch_bounds__(256) void k_dense_list(Table T, int chunksPerPair, int* __restrict__ counts,
                                                    const long long* __restrict__ offsets, float4* __restrict__ loc,
                                                    int* __restrict__ cpair, unsigned char* __restrict__ isStatic) {
  __shared__ int waveCount[4];
  __shared__ int base;
  const int p = blockIdx.x / chunksPerPair, ch = blockIdx.x - p * chunksPerPair;
  const int npx = T.W * T.H;
  const long long pixBase = static_cast<long long>(p) * npx;
  const int p0 = ch * kDlChunk, p1 = min(npx, p0 + kDlChunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int total = 0;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  for (int q0 = p0; q0 < p1; q0 += 256) {
    const int pix = q0 + threadIdx.x;
    float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
    bool ok = false;
    if (pix < p1) ok = denseCandidate(T, pix, (T.fmask + pixBase)[pix], (T.flow + pixBase)[pix], l);
    const unsigned long long b = __ballot(ok);
    if (lane == 0) waveCount[wave] = __popcll(b);
    __syncthreads();
    int before = base;
    for (int w = 0; w < wave; ++w) before += waveCount[w];
    if (offsets != nullptr && ok) {
      const long long at = offsets[blockIdx.x] + before + __popcll(b & ((1ull << lane) - 1ull));
      loc[at] = l;
      cpair[at] = p;
      isStatic[at] = 1;
    }
    total = base + waveCount[0] + waveCount[1] + waveCount[2] + waveCount[3];
    __syncthreads();
    if (threadIdx.x == 0) base = total;
    __syncthreads();
  }
  if (offsets == nullptr && threadIdx.x == 0) counts[blockIdx.x] = total;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// LIST MODE: the off-diagonal blocks of the pose-graph level, E_ab = Z_a^T X_ab Z_b with Z = [I_7 0; 0 1] (cvd_coarse.h), on the
// matrix pipe.  k_coarse_edges_fast keeps the 8 x 8 block in 64 per-lane accumulators (128 VGPRs) beside both sides' projected
// Jacobians: 288 us per build of the level on the benchmarked list, on the critical path of every solve's first iteration.  The 16
// projected columns [J_s Z (8) | J_t Z (8)] of a constraint's three residual rows are exactly one v_mfma_f64_16x16x4 tile wide: the
// rows sqrt(rho') [..] go through the wave's staging tile and the Gram tile's off-diagonal quadrant IS the block (dir 0: rows = modes
// of the source; dir 1, source = fb: transposed).  Scope: bilinear depth grid with one value parameter (KD = 4, N = 1), every pair
// kept; elsewhere k_coarse_edges_fast / k_coarse_edges.
inline __global__ __launch_bounds__(256) void k_coarse_edges_mfma(Layout L, Table T, Items it, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc, const int* __restrict__ itemEdge,
                                                           double* __restrict__ edgeOut) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double* xa = sm;
  double* xb = xa + B;
  double* Es = xb + B;                        // [2][256] the two directions' Gram tiles, summed over the waves
  double* scr = Es + 512 + wave * (64 * kDwLd);
  const int item = blockIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  for (int i = tid; i < B; i += 256) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  for (int i = tid; i < 512; i += 256) Es[i] = 0.0;
  __syncthreads();
  const int mk = lane >> 4, mc = lane & 15;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    if (cb >= ce) continue;   // (workgroup-uniform)
    const FrameConst& Fs = fc[dir ? fb : fa];
    const FrameConst& Ft = fc[dir ? fa : fb];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    DwPairConst P;
#pragma unroll
    for (int i = 0; i < 9; ++i) { P.Rs[i] = uniformValue(Fs.R[i]); P.Rt[i] = uniformValue(Ft.R[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) P.dT[i] = uniformValue(Fs.t[i] - Ft.t[i]);
    P.fys = uniformValue(Fs.fy);
    P.fxs = uniformValue(Fs.fy * L.aspect);
    P.ifys = uniformValue(1.0 / Fs.fy);
    P.ifyt = uniformValue(1.0 / Ft.fy);
    P.ifxt = uniformValue(1.0 / (Ft.fy * L.aspect));
    cvd_d4 tile = {0.0, 0.0, 0.0, 0.0};
    const int n = static_cast<int>(ce - cb);
    for (int k0 = 0; k0 < n; k0 += 256) {
      const int k = k0 + tid;
      float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
      float2 d = make_float2(0.f, 0.f);
      bool valid = false;
      if (k < n) valid = loadConstraint<false>(T, cb + k, 0, 0, 0, nd, d);
      if (__builtin_amdgcn_readfirstlane(__ballot(valid) == 0ull ? 1 : 0)) continue;
      DwState ch;
      double sw = 0.0, zf = 0.0, da = 0.0, db = 0.0;
      double y[3] = {0.0, 0.0, 0.0};
      if (valid) {
        DwTaps ts, tt;
        da = static_cast<double>(d.x);
        db = static_cast<double>(d.y);
        dwGather(L, nd.x, nd.y, ts);
        dwGather(L, nd.z, nd.w, tt);
        dwChain(L, P, xs, xt, nd, da, db, ts, tt, ch);
        sw = ch.sw;
        zf = -ch.zz * P.ifyt;
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) y[i2] = ch.Rca[i2] * ch.Da;
      }
#pragma unroll 1
      for (int r = 0; r < 3; ++r) {
        if (valid) {
          const double mu0 = r == 0 ? ch.m00 : 0.0, mu1 = r == 1 ? ch.m11 : 0.0;
          const double mu2 = r == 0 ? ch.m02 : (r == 1 ? ch.m12 : ch.m22);
          double F[kDwFeat], jds;
          dwFeatures(ch, P, y, mu0, mu1, mu2, r == 2 ? 0.0 : mu2 * zf, 0.0, F, jds);
          const double* g = F;
          const double* nn = F + 3;   // y x g
          double J[16];
          J[0] = g[0]; J[1] = g[1]; J[2] = g[2];
          J[8] = -g[0]; J[9] = -g[1]; J[10] = -g[2];
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double* as = Fs.Jl + 3 * i;
            const double* at = Ft.Jl + 3 * i;
            J[3 + i] = as[0] * nn[0] + as[1] * nn[1] + as[2] * nn[2];
            // a_t . (g x v) = g . (dT x a_t) - a_t . (y x g)
            const double c0 = P.dT[1] * at[2] - P.dT[2] * at[1], c1 = P.dT[2] * at[0] - P.dT[0] * at[2], c2 = P.dT[0] * at[1] - P.dT[1] * at[0];
            J[11 + i] = (g[0] * c0 + g[1] * c1 + g[2] * c2) - (at[0] * nn[0] + at[1] * nn[1] + at[2] * nn[2]);
          }
          J[6] = F[6];
          J[7] = jds * da;                                  // uniform depth-scale mode of the source: sum_k d r / d theta_k = JD d_src
          J[14] = F[7];
          J[15] = r == 2 ? ch.JDT2 * db : 0.0;              // ... of the target
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) scr[lane * kDwLd + cc] = sw * J[cc];
        } else {
#pragma unroll
          for (int cc = 0; cc < 16; ++cc) scr[lane * kDwLd + cc] = 0.0;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const double a0 = scr[(4 * j + mk) * kDwLd + mc];
          tile = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, tile, 0, 0, 0);
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(&Es[dir * 256 + ((lane >> 4) + 4 * q) * 16 + (lane & 15)], tile[q]);
  }
  __syncthreads();
  // rows = modes of fa, columns = modes of fb: dir 0 (source = fa) the quadrant [0:8, 8:16], dir 1 (source = fb) [8:16, 0:8]
  const int edge = itemEdge[item];
  if (tid < kCBB && edge >= 0) {
    const int i = tid >> 3, j = tid & 7;
    atomicAdd(&edgeOut[static_cast<size_t>(edge) * kCBB + tid], Es[i * 16 + 8 + j] + Es[256 + (8 + i) * 16 + j]);
  }
}

}  // namespace cvd
