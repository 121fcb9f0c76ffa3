// Coarse (pose-graph) level of the additive preconditioner of the PCG solve.  (Round 4: cvd_temporal.h adds a temporally coarse
// level for the depth grid and a temporally coarse FORM of this level -- the same 8 modes per frame x hat functions in time, built
// from the very blocks below -- which replaces the dense inverse wherever the exact factor's elimination is over budget.)
//
// The block-Jacobi preconditioner inverts every frame's own block exactly, but the slowly converging error of
// this problem lives BETWEEN frames: low-frequency drift of the camera trajectory and of the per-frame depth
// scale (the unknowns of the reference's "Global" level, lib/PoseOptimizer.cpp:1141-1226 with a Global depth
// transform).  The coarse space spans exactly these: kCB = 8 modes per frame,
//     Z_f = [ I_7  0 ;  0  1 ]    (t, w, fy | every depth-scale vertex of the frame moves together),
// and the preconditioner becomes additive two-level
//     M^-1 = blockdiag(A_ff)^-1 + Z (Z^T A Z)^-1 Z^T,       A = J^T J + diag(lam).
// A_c = Z^T A Z is block-sparse on the frame graph (8x8 blocks, one per frame and per undirected frame pair):
//   * off-diagonal blocks  sum_c rho' (J_a Z_a)^T (J_b Z_b)           k_coarse_edges   (once per linearisation)
//   * diagonal blocks      Z_f^T (H_ff + diag(lam_f)) Z_f             k_coarse_diag    (once per LM iteration)
//   * block-sparse Cholesky A_c = L L^T on a host-computed plan        k_coarse_factor_mw (32 workgroups, grid barrier per level)
//   * W = L^-1, block-sparse: W_ij != 0 only if i is an ancestor of j in the elimination tree; its columns are
//     independent chains, one wave each                                k_coarse_winv
//   * per PCG iteration y = W Z^T r (k_coarse_apply_w, which also closes the PCG scalars: r^T Z c = |y|^2); the
//     consumers form c_f = sum_t W_tf^T y_t for the frames they touch (coarseFrameCorrection, cvd_device.h)
// The regularisers enter through H_ff only (their inter-frame part, the position regulariser, is left to
// the fine level), so A_c stays SPD.  Everything is deterministic (no atomics in the solves) so that the ranks
// of the pair-sharded multi-GPU mode stay bit-identical.
#pragma once

#include "cvd_kernels.h"

namespace cvd {

constexpr size_t kCoarseMaxUnknowns = 65536;  // 8192 frames (the plan is built on the host in O(F^2))
constexpr int kCBB = kCB * kCB;   // doubles per coarse block (kCB = 8 coarse unknowns per frame, cvd_device.h)

// LDS hand-off between the lanes of ONE wave (LDS operations of a wave complete in order; the fences only pin
// the compiler).
#define CVD_WAVE_SYNC()                                      \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

// Elimination plan (device pointers, built by the host in buildCoarsePlan).  Indices are elimination POSITIONS
// unless stated otherwise.  Block ids: [0, F) diagonal block of position j, F + e off-diagonal block e of L.
struct CoarsePlan {
  int F, nBlocks, nLevels, nEdges;
  const int* order;      // position -> frame
  const int* pos;        // frame -> position
  const int* levelPtr;   // nLevels + 1, into levelCols
  const int* levelCols;  // positions grouped by level (columns of one level are mutually independent)
  const int* lvlBlkPtr;  // nLevels + 1, into lvlBlks: every block (diagonal and below) of the level's columns
  const int* lvlBlks;    // block ids
  const int* blkCol;     // block id -> column position j
  const int* blkRow;     // block id -> row position i (>= j)
  const int* colPtr;     // F + 1: off-diagonal blocks below the diagonal of column j are ids F + [colPtr[j], colPtr[j+1])
  const int* rowPtr;     // F + 1: off-diagonal blocks of ROW j (left of the diagonal)
  const int* rowBlk;     //   their block ids
  const int* updPtr;     // nBlocks + 1: left-looking update list of block (i, j): pairs L(i,k), L(j,k), k < j
  const int* updA;       //   block id of L(i, k)
  const int* updB;       //   block id of L(j, k)
  const int* edgeBlk;    // nEdges: (block id << 1) | transposed   (edge block is stored rows = fa, cols = fb)
  const int* edgeFa;     // nEdges
  const int* edgeFb;     // nEdges
  // W = L^-1: column j holds blocks at rows path(j) = j, parent(j), parent(parent(j)), ... (W block id = wPtr[j] + t)
  const int* wPtr;       // F + 1
  const int* wRow;       // row position of every W block
  const int* wtPtr;      // F + 1: W blocks of ROW i (transpose structure), ordered by column
  const int* wtBlk;      //   W block id
  const int* wtCol;      //   column position
  const int* wtFrame;    //   frame of that column (order[wtCol])
  const int* wuPtr;      // nW + 1: gather list of W block (i, j): pairs L(i,k) W(k,j), k on the path below i
  const int* wuL;        //   block id of L(i, k)
  const int* wuW;        //   W block id of W(k, j)
  int nW;                // number of W blocks
  const int* updBlk;     // per update entry: the block id it belongs to (inverse of updPtr)
};
// *flag <- (*flag != 0): the dense level's fail word (bit 0 pivot failure, bit 30 barrier timeout) before it is summed over
// the ranks of a sharded solve (a sum of bit-30 values could wrap to zero).
inline __global__ void k_flag_to_bool(int* __restrict__ flag) { *flag = (*flag != 0) ? 1 : 0; }

// ---------------------------------------------------------------------------------------------------------
// Off-diagonal coarse blocks: C_e[i][j] = sum over the constraints between frames (fa, fb), both directions,
// of rho' * sum_r (J_fa Z)[r][i] (J_fb Z)[r][j].  One workgroup per work item (k_matvec_pairs' decomposition).
// ---------------------------------------------------------------------------------------------------------
// A work item whose frame pair was left out of the (sparsified) coarse graph -- itemEdge[item] < 0, see
// sparsifyCoarseGraph in cvd_hip.hip -- contributes nothing to the coarse matrix: instead of the cross block its two
// DIAGONAL contributions sum rho' (J_f Z)^T (J_f Z), f = fa and f = fb, are accumulated into dropDiag[f], which
// k_coarse_diag subtracts from Z^T H_ff Z.  The coarse matrix is then the Galerkin operator of the kept constraints
// (+ regularisers + damping): still SPD, and consistent on the smooth inter-frame modes it exists for.
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_coarse_edges(Layout L, Table T, Items it, const double* __restrict__ x,
                                                      const FrameConst* __restrict__ fc,
                                                      const int* __restrict__ itemEdge, double* __restrict__ edgeOut,
                                                      double* __restrict__ dropDiag) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* xa = sm;
  double* xb = sm + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(sm + 2 * B);
  double* Cs = reinterpret_cast<double*>(fcs + 2);  // 64
  const int item = blockIdx.x;
  const int tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  for (int i = tid; i < B; i += 256) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  if (tid < 2 * (sizeof(FrameConst) / 8)) {
    const int which = tid / (sizeof(FrameConst) / 8);
    const int k = tid % (sizeof(FrameConst) / 8);
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  if (tid < kCBB) Cs[tid] = 0.0;
  __syncthreads();
  const bool haveScale = L.N >= 1;
  const int edge = itemEdge[item];
  // kept pair: one pass, the cross block (mode 0); dropped pair: two passes, the self blocks of fa (1) and fb (2)
  for (int mode = (edge >= 0 ? 0 : 1); mode <= (edge >= 0 ? 0 : 2); ++mode) {
  double Cacc[kCBB];  // rows: modes of fa, columns: modes of fb
#pragma unroll
  for (int i = 0; i < kCBB; ++i) Cacc[i] = 0.0;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fs = fcs[dir];
    const FrameConst& Ft = fcs[dir ^ 1];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    // rows take the Jacobian of fa's side (mode 0, 1) or fb's (mode 2); columns fb's (mode 0, 2) or fa's (mode 1);
    // fa is the source side of direction 0 and the target side of direction 1
    const bool rowIsSrc = (mode == 2) ? (dir == 1) : (dir == 0);
    const bool colIsSrc = (mode == 1) ? (dir == 0) : (dir == 1);
    for (long long c = cb + tid; c < ce; c += 256) {
      const float2 d = T.dsrc[c];
      if (!(d.x > 0.f)) continue;
      Sample<KD, KS> s;
      evalSample<KD, KS, true>(L, Fs, Ft, xs, xt, T.ndc[c], d, s);
      // Z-projected Jacobians of the two sides: 7 pose columns + the uniform depth-scale column
      // (d r / d scale_k = JD w_k d_src and the interpolation weights sum to one)
      double Jr[3][kCB], Jc[3][kCB];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          Jr[r][i] = rowIsSrc ? s.a.Jp[r][i] : s.b.Jp[r][i];
          Jc[r][i] = colIsSrc ? s.a.Jp[r][i] : s.b.Jp[r][i];
        }
        const double js = haveScale ? s.a.JD[r] * s.a.d : 0.0, jt = haveScale ? s.b.JD[r] * s.b.d : 0.0;
        Jr[r][7] = rowIsSrc ? js : jt;
        Jc[r][7] = colIsSrc ? js : jt;
      }
      const double w = s.rho1;
#pragma unroll
      for (int i = 0; i < kCB; ++i) {
        const double a0 = w * Jr[0][i], a1 = w * Jr[1][i], a2 = w * Jr[2][i];
#pragma unroll
        for (int j = 0; j < kCB; ++j) Cacc[i * kCB + j] += a0 * Jc[0][j] + a1 * Jc[1][j] + a2 * Jc[2][j];
      }
    }
  }
  __syncthreads();
  // (one slot per wave, folded in wave order: the block does not depend on which wave finishes first)
#pragma unroll
  for (int i = 0; i < kCBB; ++i) {
    const double v = waveSum(Cacc[i]);
    if ((tid & 63) == 0) Cs[(tid >> 6) * kCBB + i] = v;
  }
  __syncthreads();
  // several chunk items may share one frame pair: the outputs are zeroed by the host before the launch
  // (two items at most for a kept pair -- a two-term sum is order-free; the self blocks of DROPPED pairs collect many items in
  // arrival order: the sparsified level is outside the deterministic build's scope)
  double* out = mode == 0 ? edgeOut + static_cast<size_t>(edge) * kCBB : dropDiag + static_cast<size_t>(mode == 1 ? fa : fb) * kCBB;
  if (tid < kCBB) atomicAdd(&out[tid], (Cs[tid] + Cs[kCBB + tid]) + (Cs[2 * kCBB + tid] + Cs[3 * kCBB + tid]));
  __syncthreads();
  }  // mode
}

// Fast path of k_coarse_edges (scope of the fast kernels: identity spatial transform, reprojection losses, per-frame
// or fixed intrinsics): the residual / Jacobian chain of k_assemble_fast with register-resident taps, both sides of the
// constraint at once.  The generic kernel above goes through Sample<KD, KS>, whose dynamically indexed tap arrays live in
// scratch memory.
template <int KD, bool DENSE = false>
inline __global__ __launch_bounds__(256) void k_coarse_edges_fast(Layout L, Table T, Items it, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc,
                                                           const int* __restrict__ itemEdge, double* __restrict__ edgeOut,
                                                           double* __restrict__ dropDiag) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr double eps = 1e-6;
  const int B = L.B;
  double* xa = sm;
  double* xb = sm + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(sm + 2 * B);
  double* Cs = reinterpret_cast<double*>(fcs + 2);  // 64
  const int item = blockIdx.x;
  const int tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  for (int i = tid; i < B; i += 256) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  if (tid < 2 * (sizeof(FrameConst) / 8)) {
    const int which = tid / (sizeof(FrameConst) / 8);
    const int k = tid % (sizeof(FrameConst) / 8);
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  if (tid < kCBB) Cs[tid] = 0.0;
  __syncthreads();
  const int N = L.N;
  const double A = L.aspect;
  const bool haveScale = N >= 1;
  const int edge = itemEdge[item];
  // kept pair: one pass, the cross block (mode 0); dropped pair (see k_coarse_edges): the self blocks of fa (1), fb (2)
  for (int mode = (edge >= 0 ? 0 : 1); mode <= (edge >= 0 ? 0 : 2); ++mode) {
  double Cacc[kCBB];  // rows: modes of fa, columns: modes of fb
#pragma unroll
  for (int i = 0; i < kCBB; ++i) Cacc[i] = 0.0;
  for (int dir = 0; dir < 2; ++dir) {
    const bool rowIsSrc = (mode == 2) ? (dir == 1) : (dir == 0);
    const bool colIsSrc = (mode == 1) ? (dir == 0) : (dir == 1);
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fa = fcs[dir];      // source frame of this direction
    const FrameConst& Fb = fcs[dir ^ 1];  // target frame
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    const double fya = Fa.fy, fxa = Fa.fy * A;
    const double fyb = Fb.fy;
    const double ifyb = 1.0 / fyb, ifxb = 1.0 / (fyb * A);
    const int fsrc = dir ? fb : fa, ftgt = dir ? fa : fb;
    const long long pixBase = DENSE ? (cb / (static_cast<long long>(T.W) * T.H)) * (static_cast<long long>(T.W) * T.H) : 0;
    for (long long c = cb + tid; c < ce; c += 256) {
      float4 nd;
      float2 d;
      if (!loadConstraint<DENSE>(T, c, pixBase, fsrc, ftgt, nd, d)) continue;
      const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
      double Da, Db;
      if (N == 0) {
        Da = da;
        Db = db;
      } else {
        FastTaps<KD> ta, tb;
        fastGather<KD>(L, nd.x, nd.y, ta);
        fastGather<KD>(L, nd.z, nd.w, tb);
        Da = 0.0;
        Db = 0.0;
#pragma unroll
        for (int k = 0; k < KD; ++k) {
          if (ta.ok(k)) {
            const int ia = ta.I(k);
            Da += (N == 2 ? da * xs[7 + ia * 2] + xs[7 + ia * 2 + 1] : da * xs[7 + ia]) * ta.Wt(k);
          }
          if (tb.ok(k)) {
            const int ib = tb.I(k);
            Db += (N == 2 ? db * xt[7 + ib * 2] + xt[7 + ib * 2 + 1] : db * xt[7 + ib]) * tb.Wt(k);
          }
        }
      }
      const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
      const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
      const double ca[3] = {pax * fxa, pay * fya, -1.0};
      const double Rca[3] = {dot3(Fa.R, ca), dot3(Fa.R + 3, ca), dot3(Fa.R + 6, ca)};
      const double v[3] = {Fa.t[0] + Rca[0] * Da - Fb.t[0], Fa.t[1] + Rca[1] * Da - Fb.t[1], Fa.t[2] + Rca[2] * Da - Fb.t[2]};
      const double q0 = Fb.R[0] * v[0] + Fb.R[3] * v[1] + Fb.R[6] * v[2];
      const double q1 = Fb.R[1] * v[0] + Fb.R[4] * v[1] + Fb.R[7] * v[2];
      const double q2 = Fb.R[2] * v[0] + Fb.R[5] * v[1] + Fb.R[8] * v[2];
      const double zz = -q2;
      const double iz = 1.0 / zz;
      const double u = q0 * iz * ifxb;
      const double vv = q1 * iz * ifyb;
      const double r0 = (u - pbx) * L.ws;
      const double r1 = (vv - pby) * L.ws;
      double r2, dr2dA, dr2dDb;
      if (L.lossType == kLossDisparity) {
        const bool zo = !(zz < eps), bo = !(Db < eps);
        const double izc = zo ? iz : 1.0 / eps, ibc = 1.0 / (bo ? Db : eps);
        r2 = (izc - ibc) * L.wd;
        dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
        dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
      } else {
        const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
        const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
        if (L.lossType == kLossRatio) {
          r2 = (mx / mn - 1.0) * L.wd;
          const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
          dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
          dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
        } else {
          r2 = log(mn / mx) * L.wd;
          const double dmn = 1.0 / mn, dmx = -1.0 / mx;
          dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
          dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
        }
      }
      const double w = robustRho1(L, r0 * r0 + r1 * r1 + r2 * r2);
      // d r / d q (rows): M0 = (m00, 0, m02), M1 = (0, m11, m12), M2 = (0, 0, m22)
      const double wiz = L.ws * iz;
      const double m00 = wiz * ifxb, m11 = wiz * ifyb, m02 = wiz * u, m12 = wiz * vv, m22 = -dr2dA;
      // Z-projected Jacobians: 7 pose columns + the uniform depth-scale column (d r / d scale_k = JD w_k d_src and the
      // interpolation weights sum to one) of the source side (Js) and of the target side (Jt)
      double Js[3][kCB], Jt[3][kCB];
      {
        // source: G = M R_b^T ; columns: t -> G, w_i -> G (D_a dR_a,i c_a), fy -> G (D_a R_a cf), D -> G R c_a
        double G[3][3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          G[0][i] = m00 * Fb.R[i * 3 + 0] + m02 * Fb.R[i * 3 + 2];
          G[1][i] = m11 * Fb.R[i * 3 + 1] + m12 * Fb.R[i * 3 + 2];
          G[2][i] = m22 * Fb.R[i * 3 + 2];
        }
        const double cf[3] = {pax * A, pay, 0.0};
        const double dXdf[3] = {Da * (Fa.R[0] * cf[0] + Fa.R[1] * cf[1]), Da * (Fa.R[3] * cf[0] + Fa.R[4] * cf[1]),
                                Da * (Fa.R[6] * cf[0] + Fa.R[7] * cf[1])};
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
          Js[rr][0] = G[rr][0];
          Js[rr][1] = G[rr][1];
          Js[rr][2] = G[rr][2];
          Js[rr][6] = dot3(G[rr], dXdf);
          Js[rr][7] = haveScale ? dot3(G[rr], Rca) * da : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double dX[3] = {Da * dot3(Fa.dR[i], ca), Da * dot3(Fa.dR[i] + 3, ca), Da * dot3(Fa.dR[i] + 6, ca)};
          Js[0][3 + i] = dot3(G[0], dX);
          Js[1][3 + i] = dot3(G[1], dX);
          Js[2][3 + i] = dot3(G[2], dX);
        }
        // target
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          Jt[0][i] = -G[0][i];
          Jt[1][i] = -G[1][i];
          Jt[2][i] = -G[2][i];
          const double* D = Fb.dR[i];  // d q / d w_b,i = dR_b,i^T v
          const double dq0 = D[0] * v[0] + D[3] * v[1] + D[6] * v[2];
          const double dq1 = D[1] * v[0] + D[4] * v[1] + D[7] * v[2];
          const double dq2 = D[2] * v[0] + D[5] * v[1] + D[8] * v[2];
          Jt[0][3 + i] = m00 * dq0 + m02 * dq2;
          Jt[1][3 + i] = m11 * dq1 + m12 * dq2;
          Jt[2][3 + i] = m22 * dq2;
        }
        Jt[0][6] = -L.ws * u * ifyb;
        Jt[1][6] = -L.ws * vv * ifyb;
        Jt[2][6] = 0.0;
        Jt[0][7] = 0.0;
        Jt[1][7] = 0.0;
        Jt[2][7] = haveScale ? dr2dDb * db : 0.0;
      }
      // (uniform selects: rows take fa's side, columns fb's for the cross block; fa is the source side of direction 0)
#pragma unroll
      for (int i = 0; i < kCB; ++i) {
        const double a0 = w * (rowIsSrc ? Js[0][i] : Jt[0][i]), a1 = w * (rowIsSrc ? Js[1][i] : Jt[1][i]),
                     a2 = w * (rowIsSrc ? Js[2][i] : Jt[2][i]);
#pragma unroll
        for (int j = 0; j < kCB; ++j)
          Cacc[i * kCB + j] += a0 * (colIsSrc ? Js[0][j] : Jt[0][j]) + a1 * (colIsSrc ? Js[1][j] : Jt[1][j]) +
                               a2 * (colIsSrc ? Js[2][j] : Jt[2][j]);
      }
    }
  }
  __syncthreads();
  // (one slot per wave, folded in wave order: the block does not depend on which wave finishes first)
#pragma unroll
  for (int i = 0; i < kCBB; ++i) {
    const double v = waveSum(Cacc[i]);
    if ((tid & 63) == 0) Cs[(tid >> 6) * kCBB + i] = v;
  }
  __syncthreads();
  // several chunk items may share one frame pair: the outputs are zeroed by the host before the launch
  // (two items at most for a kept pair -- a two-term sum is order-free; the self blocks of DROPPED pairs collect many items in
  // arrival order: the sparsified level is outside the deterministic build's scope)
  double* out = mode == 0 ? edgeOut + static_cast<size_t>(edge) * kCBB : dropDiag + static_cast<size_t>(mode == 1 ? fa : fb) * kCBB;
  if (tid < kCBB) atomicAdd(&out[tid], (Cs[tid] + Cs[kCBB + tid]) + (Cs[2 * kCBB + tid] + Cs[3 * kCBB + tid]));
  __syncthreads();
  }  // mode
}

// ---------------------------------------------------------------------------------------------------------
// Diagonal coarse blocks D_f = Z_f^T (H_ff + diag(lam_f)) Z_f and the mode activity flags.  Inactive modes
// (masked unknowns, frames outside the range, the shared focal length) become identity rows.
// ---------------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void k_coarse_diag(Layout L, const double* __restrict__ hBlocks,
                                                     const double* __restrict__ lam, const double* __restrict__ mask,
                                                     double* __restrict__ diagOut, unsigned char* __restrict__ modeActive,
                                                     double lamScale, const double* __restrict__ dropDiag) {
  __shared__ double u[520];  // u[r] = sum over scale vertices v of H[r][v] (+ lam on the diagonal); B <= 512
  __shared__ double red[4];
  __shared__ int anyScale;
  const int B = L.B, f = blockIdx.x, tid = threadIdx.x;
  const double* hf = hBlocks + static_cast<size_t>(f) * B * B;
  const double* lf = lam + static_cast<size_t>(f) * B;
  const double* mf = mask + static_cast<size_t>(f) * B;
  const int N = L.N;
  const int nV = (N >= 1 && L.depthType != kDepthIdentity) ? L.nD / N : 0;
  if (tid == 0) anyScale = 0;
  __syncthreads();
  for (int r = tid; r < B; r += 256) {
    double a = 0.0;
    for (int v = 0; v < nV; ++v) {
      const int cidx = 7 + v * N;
      a += hf[static_cast<size_t>(cidx) * B + r] + (r == cidx ? lamScale * lf[r] : 0.0);  // (symmetric block: coalesced over r)
    }
    u[r] = a;
    if (r >= 7 && r < 7 + L.nD && ((r - 7) % (N > 0 ? N : 1)) == 0 && mf[r] != 0.0) anyScale = 1;
  }
  __syncthreads();
  double s77 = 0.0;
  for (int v = tid; v < nV; v += 256) s77 += u[7 + v * N];
  s77 = waveSum(s77);
  if ((tid & 63) == 0) red[tid >> 6] = s77;
  __syncthreads();
  if (tid < kCBB) {
    const int i = tid / kCB, j = tid % kCB;
    auto active = [&](int m) -> bool {
      if (m < 7) return mf[m] != 0.0 && !(m == 6 && L.intrOpt == kIntrShared);
      return nV > 0 && anyScale != 0;
    };
    double v;
    if (i < 7 && j < 7) v = hf[static_cast<size_t>(i) * B + j] + (i == j ? lamScale * lf[i] : 0.0);
    else if (i < 7) v = u[i];
    else if (j < 7) v = u[j];
    else v = red[0] + red[1] + red[2] + red[3];
    if (dropDiag != nullptr) v -= dropDiag[static_cast<size_t>(f) * kCBB + tid];  // pairs left out of the coarse graph
    const bool ai = active(i), aj = active(j);
    if (!(ai && aj)) v = (i == j) ? 1.0 : 0.0;
    diagOut[static_cast<size_t>(f) * kCBB + tid] = v;
    if (j == 0) modeActive[f * kCB + i] = ai ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Block-sparse Cholesky A_c = L L^T, left-looking by levels of the elimination plan (per column: gather
// A_ij - sum_k L_ik L_jk^T over the columns of lower levels, dense 8x8 Cholesky of the diagonal block and the inverse of
// its factor, L_ij = G Linv_jj^T for the off-diagonal blocks), on several workgroups.  A column's work (gather for its blocks, diagonal Cholesky,
// scaling of its off-diagonal blocks) depends only on columns of LOWER levels, so a workgroup takes whole columns
// and the workgroups meet once per level at a grid barrier (all kCoarseFactorGroups workgroups are co-resident:
// far fewer than CUs).  Inside a column the 16 waves split every update list and fold their partial sums in wave
// order (deterministic).  Lb must be zeroed and *barrier set to 0 by the host before the launch.
// ---------------------------------------------------------------------------------------------------------
constexpr int kCoarseFactorGroups = 32;

__device__ __forceinline__ void coarseGridBarrier(unsigned int* counter, unsigned int target, int* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1u << 24)) {  // never hang the device: give up and report (the level is switched off)
        atomicAdd(fail, 1);
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

inline __global__ __launch_bounds__(1024) void k_coarse_factor_mw(CoarsePlan P, const double* __restrict__ diag,
                                                           const double* __restrict__ edges,
                                                           const unsigned char* __restrict__ modeActive,
                                                           double* __restrict__ Lb, double* __restrict__ Linv,
                                                           int* __restrict__ fail, unsigned int* __restrict__ barrier) {
  __shared__ double scratch[16][2 * kCBB];
  __shared__ double fold[16][kCBB];
  __shared__ double foldLast[16][kCBB];
  __shared__ int edgeBlkOf[16][2];
  constexpr int kMaxColBlk = 40;  // staging of one column: diagonal + off-diagonal blocks (20 KB)
  __shared__ double colAcc[kMaxColBlk * kCBB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane >> 3, c = lane & 7;
  const int nG = gridDim.x, g = blockIdx.x;
  unsigned int phase = 0;
  // load: diagonal blocks by position, edge blocks (masked by the mode flags, transposed if needed)
  for (int j = g * 16 + wv; j < P.F; j += nG * 16) Lb[static_cast<size_t>(j) * kCBB + lane] = diag[static_cast<size_t>(P.order[j]) * kCBB + lane];
  for (int e = g * 16 + wv; e < P.nEdges; e += nG * 16) {
    const int code = P.edgeBlk[e];
    const int b = code >> 1, tr = code & 1;
    const int fa = P.edgeFa[e], fb = P.edgeFb[e];
    const int ra = tr ? c : r, cb2 = tr ? r : c;
    double v = edges[static_cast<size_t>(e) * kCBB + ra * kCB + cb2];
    if (!modeActive[fa * kCB + ra] || !modeActive[fb * kCB + cb2]) v = 0.0;
    Lb[static_cast<size_t>(b) * kCBB + lane] = v;
  }
  coarseGridBarrier(barrier, ++phase * nG, fail);
  double* sA = scratch[wv];
  double* sB = scratch[wv] + kCBB;
  for (int lv = 0; lv < P.nLevels; ++lv) {
    for (int q = P.levelPtr[lv] + g; q < P.levelPtr[lv + 1]; q += nG) {
      const int j = P.levelCols[q];
      const int nOff = P.colPtr[j + 1] - P.colPtr[j];
      // ---- gather.  The update lists of the column's blocks (diagonal: ids of block j; off-diagonal: one contiguous
      // range, their block ids are consecutive) are cut into 16 equal slices, one per wave, whatever block an entry
      // belongs to: a wave accumulates in registers while the block stays the same and flushes when it changes.  A block in
      // the MIDDLE of a wave's slice belongs to that wave alone (plain store into the column's LDS staging); the partial
      // sums of the slice's FIRST and LAST block -- blocks that neighbouring slices may share -- are parked per wave and
      // folded in wave order afterwards.  No atomics: the factor is bit-identical from run to run and from rank to rank
      // (the ranks of a sharded solve each build it and must take the same PCG decisions).  The serial depth per column is
      // (updates / 16) products and two barriers instead of one pair of barriers per block.
      const int d0 = P.updPtr[j], nd = P.updPtr[j + 1] - d0;
      const int ob = P.F + P.colPtr[j];
      const int o0 = P.updPtr[ob], no = P.updPtr[ob + nOff] - o0;
      const int U = nd + no;
      if (U > 0 && nOff + 1 <= kMaxColBlk) {
        for (int i = tid; i < (nOff + 1) * kCBB; i += 1024) colAcc[i] = 0.0;
        if (lane < 2) edgeBlkOf[wv][lane] = -1;
        __syncthreads();
        const int per = (U + 15) >> 4;
        const int i0 = wv * per, i1 = min(U, i0 + per);
        if (i0 < i1) {
          bool firstFlush = true;
          auto entry = [&](int idx, int& u, int& blk) {
            if (idx < nd) { u = d0 + idx; blk = 0; }
            else { u = o0 + (idx - nd); blk = P.updBlk[u] - ob + 1; }
          };
          int u, blk;
          entry(i0, u, blk);
          double na = Lb[static_cast<size_t>(P.updA[u]) * kCBB + lane];
          double nb = Lb[static_cast<size_t>(P.updB[u]) * kCBB + lane];
          double acc = 0.0;
          for (int idx = i0; idx < i1; ++idx) {
            sA[lane] = na;
            sB[lane] = nb;
            CVD_WAVE_SYNC();
            int nblk = -1;
            if (idx + 1 < i1) {
              int un;
              entry(idx + 1, un, nblk);
              na = Lb[static_cast<size_t>(P.updA[un]) * kCBB + lane];
              nb = Lb[static_cast<size_t>(P.updB[un]) * kCBB + lane];
            }
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < kCB; ++m) s += sA[r * kCB + m] * sB[c * kCB + m];
            acc += s;
            CVD_WAVE_SYNC();
            if (nblk != blk) {  // wave-uniform
              if (firstFlush) {
                fold[wv][lane] = acc;
                if (lane == 0) edgeBlkOf[wv][0] = blk;
              } else if (nblk < 0) {
                foldLast[wv][lane] = acc;
                if (lane == 0) edgeBlkOf[wv][1] = blk;
              } else {
                colAcc[blk * kCBB + lane] = acc;
              }
              firstFlush = false;
              acc = 0.0;
              blk = nblk;
            }
          }
        }
        __syncthreads();
        if (wv == 0) {
          for (int w2 = 0; w2 < 16; ++w2) {
            const int b0 = edgeBlkOf[w2][0], b1 = edgeBlkOf[w2][1];
            if (b0 >= 0) colAcc[b0 * kCBB + lane] += fold[w2][lane];
            if (b1 >= 0) colAcc[b1 * kCBB + lane] += foldLast[w2][lane];
          }
        }
        __syncthreads();
        for (int k = wv; k <= nOff; k += 16) {
          const int b = (k == 0) ? j : ob + k - 1;
          Lb[static_cast<size_t>(b) * kCBB + lane] -= colAcc[k * kCBB + lane];
        }
        __syncthreads();
      } else
      for (int k = 0; k <= nOff; ++k) {
        const int b = (k == 0) ? j : P.F + P.colPtr[j] + k - 1;
        const int u0 = P.updPtr[b], u1 = P.updPtr[b + 1];
        if (u0 == u1) continue;  // uniform
        double acc = 0.0, na = 0.0, nb = 0.0;
        int uidx = u0 + wv;
        if (uidx < u1) {
          na = Lb[static_cast<size_t>(P.updA[uidx]) * kCBB + lane];
          nb = Lb[static_cast<size_t>(P.updB[uidx]) * kCBB + lane];
        }
        for (; uidx < u1; uidx += 16) {
          sA[lane] = na;
          sB[lane] = nb;
          CVD_WAVE_SYNC();
          if (uidx + 16 < u1) {
            na = Lb[static_cast<size_t>(P.updA[uidx + 16]) * kCBB + lane];
            nb = Lb[static_cast<size_t>(P.updB[uidx + 16]) * kCBB + lane];
          }
          double s = 0.0;
#pragma unroll
          for (int m = 0; m < kCB; ++m) s += sA[r * kCB + m] * sB[c * kCB + m];
          acc += s;
          CVD_WAVE_SYNC();
        }
        fold[wv][lane] = acc;
        __syncthreads();
        if (wv == 0) {
          double t = 0.0;
#pragma unroll
          for (int w = 0; w < 16; ++w) t += fold[w][lane];
          Lb[static_cast<size_t>(b) * kCBB + lane] -= t;
        }
        __syncthreads();
      }
      // ---- diagonal block: dense 8x8 Cholesky and the inverse of its factor (wave 0)
      if (wv == 0) {
        double* S = scratch[0];
        S[lane] = Lb[static_cast<size_t>(j) * kCBB + lane];
        for (int k = 0; k < kCB; ++k) {
          CVD_WAVE_SYNC();
          double d = S[k * kCB + k];
          if (!(d > 0.0)) {
            if (lane == 0) atomicAdd(fail, 1);
            d = 1.0;
          }
          const double sd = sqrt(d);
          const double lrk = S[r * kCB + k] / sd, lck = S[c * kCB + k] / sd;
          CVD_WAVE_SYNC();
          if (c == k && r >= k) S[lane] = (r == k) ? sd : lrk;
          else if (r > k && c > k && c <= r) S[lane] -= lrk * lck;
        }
        CVD_WAVE_SYNC();
        if (c > r) S[lane] = 0.0;
        CVD_WAVE_SYNC();
        Lb[static_cast<size_t>(j) * kCBB + lane] = S[lane];
        double* Iv = Linv + static_cast<size_t>(j) * kCBB;
        if (lane < kCB) {
          double col[kCB];
#pragma unroll
          for (int i = 0; i < kCB; ++i) {
            double v = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int m = 0; m < kCB; ++m)
              if (m < i) v -= S[i * kCB + m] * col[m];
            col[i] = v / S[i * kCB + i];
          }
#pragma unroll
          for (int i = 0; i < kCB; ++i) Iv[i * kCB + lane] = col[i];
        }
      }
      __syncthreads();
      // ---- off-diagonal blocks: L_ij = G Linv_jj^T
      for (int k = wv; k < nOff; k += 16) {
        const int b = P.F + P.colPtr[j] + k;
        sA[lane] = Lb[static_cast<size_t>(b) * kCBB + lane];
        sB[lane] = Linv[static_cast<size_t>(j) * kCBB + lane];
        CVD_WAVE_SYNC();
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < kCB; ++m) s += sA[r * kCB + m] * sB[c * kCB + m];
        Lb[static_cast<size_t>(b) * kCBB + lane] = s;
        CVD_WAVE_SYNC();
      }
      __syncthreads();
    }
    coarseGridBarrier(barrier, ++phase * nG, fail);
  }
}

// ---------------------------------------------------------------------------------------------------------
// W = L^-1.  Column j of W is the solution of L w = E_j; it is non-zero only on the path from j to the root of
// the elimination tree, and the columns do not depend on each other: one wave per column walks up its path,
//   W_jj = Linv_jj,    W_ij = -Linv_ii sum_{k on the path below i, L_ik != 0} L_ik W_kj.
// lane = (r, c) of the 8x8 block.  The (L_ik, W_kj) gather list of every W block is built on the host.
// ---------------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(256) void k_coarse_winv(CoarsePlan P, const double* __restrict__ Lb,
                                                     const double* __restrict__ Linv, double* __restrict__ Wb) {
  __shared__ double scratch[4][3 * kCBB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane >> 3, c = lane & 7;
  const int j = blockIdx.x * 4 + wv;
  if (j >= P.F) return;
  double* sA = scratch[wv];
  double* sB = sA + kCBB;
  double* sC = sB + kCBB;
  const int w0 = P.wPtr[j], len = P.wPtr[j + 1] - w0;
  Wb[static_cast<size_t>(w0) * kCBB + lane] = Linv[static_cast<size_t>(j) * kCBB + lane];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (int t = 1; t < len; ++t) {
    const int wb = w0 + t;
    const int i = P.wRow[wb];
    const double ivr = Linv[static_cast<size_t>(i) * kCBB + lane];  // staged below: Linv_ii
    // gather list of this block (host-built): pairs (L_ik, W_kj) with k on the path below i
    const int u0 = P.wuPtr[wb], u1 = P.wuPtr[wb + 1];
    double acc = 0.0, na = 0.0, nb = 0.0;
    if (u0 < u1) {
      na = Lb[static_cast<size_t>(P.wuL[u0]) * kCBB + lane];
      nb = Wb[static_cast<size_t>(P.wuW[u0]) * kCBB + lane];
    }
    for (int u = u0; u < u1; ++u) {
      sA[lane] = na;
      sB[lane] = nb;
      CVD_WAVE_SYNC();
      if (u + 1 < u1) {
        na = Lb[static_cast<size_t>(P.wuL[u + 1]) * kCBB + lane];
        nb = Wb[static_cast<size_t>(P.wuW[u + 1]) * kCBB + lane];
      }
      double s2 = 0.0;
#pragma unroll
      for (int m = 0; m < kCB; ++m) s2 += sA[r * kCB + m] * sB[m * kCB + c];
      acc += s2;
      CVD_WAVE_SYNC();
    }
    sC[lane] = acc;
    sA[lane] = ivr;
    CVD_WAVE_SYNC();
    double v = 0.0;
#pragma unroll
    for (int m = 0; m < kCB; ++m) v += sA[r * kCB + m] * sC[m * kCB + c];
    CVD_WAVE_SYNC();
    Wb[static_cast<size_t>(wb) * kCBB + lane] = -v;
    // later steps of this wave read the block back through global memory: make it visible
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

// PCG scalars: see pcgFinishScalars (cvd_kernels.h).
// ---------------------------------------------------------------------------------------------------------
// y = W (Z^T r): row i gathers W_ij rc_j over the columns j of its subtree (fixed order).  One workgroup per
// row, the four waves take interleaved quarters of the list; coarse indices are frame * 8 + mode.
// ---------------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(1024) void k_coarse_apply_w(CoarsePlan P, const double* __restrict__ Wb,
                                                         const double* __restrict__ rc, double* __restrict__ y,
                                                         double* __restrict__ dotPart, double* __restrict__ scal,
                                                         unsigned int* __restrict__ counter,
                                                         const int* __restrict__ fail, int init, double tol2,
                                                         double* __restrict__ hostMirror) {
  __shared__ double part[16][kCB];
  __shared__ double red[16];
  __shared__ int flag;
  if (!init && scal[S_DONE] != 0.0) return;
  const int i = blockIdx.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane >> 3, c = lane & 7;
  // the rows near the root of the elimination tree gather from hundreds of columns: 16 waves share the list
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const int e1 = P.wtPtr[i + 1];
  int e = P.wtPtr[i] + wv;
  for (; e + 48 < e1; e += 64) {  // four independent gathers in flight per wave
    a0 += Wb[static_cast<size_t>(P.wtBlk[e]) * kCBB + lane] * rc[P.wtFrame[e] * kCB + c];
    a1 += Wb[static_cast<size_t>(P.wtBlk[e + 16]) * kCBB + lane] * rc[P.wtFrame[e + 16] * kCB + c];
    a2 += Wb[static_cast<size_t>(P.wtBlk[e + 32]) * kCBB + lane] * rc[P.wtFrame[e + 32] * kCB + c];
    a3 += Wb[static_cast<size_t>(P.wtBlk[e + 48]) * kCBB + lane] * rc[P.wtFrame[e + 48] * kCB + c];
  }
  for (; e < e1; e += 16) a0 += Wb[static_cast<size_t>(P.wtBlk[e]) * kCBB + lane] * rc[P.wtFrame[e] * kCB + c];
  double acc = (a0 + a1) + (a2 + a3);
  acc += dppMove<0xB1>(acc);
  acc += dppMove<0x4E>(acc);
  acc += dppMove<0x141>(acc);
  if (c == 0) part[wv][r] = acc;
  __syncthreads();
  if (tid < kCB) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][tid];
    y[i * kCB + tid] = t;
    part[0][tid] = t * t;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kCB; ++k) t += part[0][k];
    dotPart[i] = t;
  }
  // r^T Z A_c^-1 Z^T r = |W Z^T r|^2 = |y|^2: the last workgroup adds it to the block-Jacobi part of r^T z
  // (S_RZPART, left by k_cg_update) and finishes the PCG scalars of this iteration
  if (!lastBlockArrives(counter, gridDim.x, &flag)) return;
  double t = 0.0;
  for (int b = tid; b < static_cast<int>(gridDim.x); b += 1024) t += dotPart[b];
  t = waveSum(t);
  if (lane == 0) red[wv] = t;
  __syncthreads();
  if (tid == 0) {
    double dot = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) dot += red[w];
    if (*fail != 0) dot = 0.0;  // a broken-down factorisation switches the level off (the consumers use c = 0)
    pcgFinishScalars(scal, init, scal[S_RZPART] + dot, scal[S_RR], tol2, hostMirror);
  }
}

// ---------------------------------------------------------------------------------------------------------
// c = W^T y for ALL frames into global memory.  The PCG kernels form the c_f they need themselves
// (coarseFrameCorrection); this kernel only runs with the position regulariser (whose rows read the neighbours'
// directions) and for the test hook.
// ---------------------------------------------------------------------------------------------------------
// One workgroup (4 waves) per frame: the frame's column of W (its elimination-tree path, <= tree depth blocks) is dealt
// over the waves, four blocks per wave in flight (clamped index, zero weight), so the kernel is one or two dependent
// round trips (row index -> y gather) instead of one per four blocks of a single wave walking the whole path.
inline __global__ __launch_bounds__(256) void k_coarse_apply_wt(CoarseView V, int F, double* __restrict__ cOut,
                                                         const double* __restrict__ scal, int init) {
  __shared__ double part[4][kCB];
  const double sDone = init ? 0.0 : scal[S_DONE];  // (tested at the store: the flag rides on the first round trip)
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int f = blockIdx.x;
  const int r = lane >> 3;
  const int j = V.pos[f];
  const int w0 = V.wPtr[j], len = V.wPtr[j + 1] - w0;
  double a0 = 0.0, a1 = 0.0;
  for (int t0 = wv; t0 < len; t0 += 16) {
    int row[4];
    double wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int tc = w0 + min(t0 + 4 * u, len - 1);
      row[u] = V.wRow[tc];
      wb[u] = V.Wb[static_cast<size_t>(tc) * 64 + lane];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const double yv = V.y[row[u] * kCB + r];
      const double term = (t0 + 4 * u < len) ? wb[u] * yv : 0.0;
      if (u & 1) a1 += term; else a0 += term;
    }
  }
  double acc = a0 + a1;
  acc += __shfl_xor(acc, 8, 64);   // sum over r: lanes with equal c are 8 apart
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (lane < kCB) part[wv][lane] = acc;
  __syncthreads();
  if (tid < kCB && sDone == 0.0) {
    const double c = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    cOut[f * kCB + tid] = (*V.fail == 0 && V.modeActive[f * kCB + tid]) ? c : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// DENSE coarse level.  The sparse factorisation above follows the fill of the frame graph: fine for the reference
// sampler's hierarchical flow list (~10 k block updates at 300 frames), hopeless for a flow list with long-range pairs
// from nearly every frame (the "~4k pairs" list of BASELINE.json: ~10^6 updates, 44 ms per factorisation).  For such
// graphs the coarse matrix (8 F unknowns: 2400 at 300 frames) is simply treated as dense: assembled from the same
// diagonal / edge blocks, inverted by ONE persistent kernel on the f64 matrix cores (k_dense_spd_inverse,
// cvd_dense_inverse.h) and applied as an f64 matrix-vector product per PCG iteration (k_coarse_dense_apply: c = A_c^-1 Z^T r and its share of
// r^T z, 46 MB streamed).  Unknowns in FRAME order (8 f + mode); inactive modes are identity rows.
// ---------------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(64) void k_coarse_dense_assemble(int F, int nEdges, const double* __restrict__ diag,
                                                              const double* __restrict__ edges,
                                                              const int* __restrict__ edgeFa, const int* __restrict__ edgeFb,
                                                              const unsigned char* __restrict__ modeActive,
                                                              double* __restrict__ A, double shift) {
  const size_t n = static_cast<size_t>(F) * kCB;
  const int b = blockIdx.x, t = threadIdx.x, i = t >> 3, j = t & 7;
  if (b < F) {
    // (the diagonal carries a relative shift, cvd_solver_options::coarse_dense_shift: A_c is singular along the 7 gauge
    // directions of the trajectory up to the LM damping, i.e. up to 1e-10 ... 1e-16 of its norm once the trust region has
    // grown.  An explicit inverse of such a matrix is dominated by those directions and its rounding errors -- of relative
    // size cond(A_c) eps -- swamp its small eigenvalues: the applied "inverse" is indefinite and PCG runs into its
    // iteration cap, in which LM iteration depended on rounding.  f64 storage alone moves the limit from cond ~ 1e7 to
    // ~ 1e8; the shift keeps cond of the Jacobi-scaled matrix below ~ 1e7.  The level is a preconditioner: the shift is
    // invisible on every direction that carries gradient.)
    const double v = diag[static_cast<size_t>(b) * kCBB + t];
    A[(static_cast<size_t>(b) * kCB + i) * n + b * kCB + j] = (i == j) ? v * (1.0 + shift) : v;
  } else if (b - F < nEdges) {
    const int e = b - F, fa = edgeFa[e], fb = edgeFb[e];  // block stored rows = fa, columns = fb
    double v = edges[static_cast<size_t>(e) * kCBB + t];
    if (!modeActive[fa * kCB + i] || !modeActive[fb * kCB + j]) v = 0.0;
    A[(static_cast<size_t>(fa) * kCB + i) * n + fb * kCB + j] = v;
    A[(static_cast<size_t>(fb) * kCB + j) * n + fa * kCB + i] = v;
  }
}

// Frame blocks beyond the register-resident inverses (B > 256: ScaleShift value transforms on the 17x10 grid, B = 347): the
// block-Jacobi inverses go through rocSOLVER's strided-batched potrf / potri.  k_blocks_add_diag forms H_ff + diag(lam) in
// a scratch copy, k_blocks_pack mirrors the inverse
// (rocSOLVER's potri(lower) on the column-major view leaves the inverse in the UPPER triangle of the row-major array)
// into the f32 blocks; a block whose factorisation failed becomes the inverse of its diagonal and is counted in `fail`.
inline __global__ __launch_bounds__(256) void k_blocks_add_diag(int B, size_t total, const double* __restrict__ H,
                                                         const double* __restrict__ lam, double* __restrict__ out) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t f = idx / (static_cast<size_t>(B) * B), rem = idx - f * B * B;
  const int i = static_cast<int>(rem / B), j = static_cast<int>(rem - static_cast<size_t>(i) * B);
  double v = H[idx];
  if (i == j) {
    v += lam[f * B + i];
    if (!(v > 0.0)) v = 1.0;  // (masked unknown without damping: identity row, as the register-resident kernels treat it)
  }
  out[idx] = v;
}
inline __global__ __launch_bounds__(256) void k_blocks_pack(int B, size_t total, const double* __restrict__ A, const double* __restrict__ H,
                                                     const double* __restrict__ lam, const int* __restrict__ info,
                                                     float* __restrict__ out, int* __restrict__ fail) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (idx >= total) return;
  const size_t bb = static_cast<size_t>(B) * B;
  const size_t f = idx / bb, rem = idx - f * bb;
  const int i = static_cast<int>(rem / B), j = static_cast<int>(rem - static_cast<size_t>(i) * B);
  const int nf = static_cast<int>(total / bb);
  if (info[f] != 0 || info[nf + f] != 0) {
    if (rem == 0) atomicAdd(fail, 1);
    const double d = H[f * bb + static_cast<size_t>(i) * B + i] + lam[f * B + i];
    out[idx] = (i == j) ? static_cast<float>(d > 0.0 ? 1.0 / d : 1.0) : 0.f;
    return;
  }
  out[idx] = static_cast<float>(j >= i ? A[f * bb + static_cast<size_t>(i) * B + j] : A[f * bb + static_cast<size_t>(j) * B + i]);
}

// c_f = (A_c^-1 Z^T r)_f for the 8 modes of frame f (one workgroup per frame: 8 rows x n, 32 threads per row) and this
// frame's share of r^T Z A_c^-1 Z^T r; the last workgroup closes the PCG scalars exactly as k_coarse_apply_w does.
inline __global__ __launch_bounds__(256) void k_coarse_dense_apply(int F, const double* __restrict__ Ainv,
                                                            const double* __restrict__ rc, double* __restrict__ cOut,
                                                            const unsigned char* __restrict__ modeActive,
                                                            double* __restrict__ dotPart, double* __restrict__ scal,
                                                            unsigned int* __restrict__ counter, const int* __restrict__ fail,
                                                            int init, double tol2, double* __restrict__ hostMirror) {
  __shared__ double cs[kCB];
  __shared__ double red[4];
  __shared__ int flag;
  if (!init && scal[S_DONE] != 0.0) return;
  const int f = blockIdx.x, tid = threadIdx.x;
  const int m = tid >> 5, part = tid & 31;
  const size_t n = static_cast<size_t>(F) * kCB;
  // 16-byte loads (n = 8 F: every row starts on a 16-byte boundary), four in flight per thread: the kernel streams 8 n^2
  // bytes (46 MB at 300 frames) and is latency-bound otherwise
  const double2* row = reinterpret_cast<const double2*>(Ainv + (static_cast<size_t>(f) * kCB + m) * n);
  const double2* rc2 = reinterpret_cast<const double2*>(rc);
  const size_t n2 = n / 2;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  size_t j = part;
  for (; j + 96 < n2; j += 128) {
    const double2 w0 = row[j], w1 = row[j + 32], w2 = row[j + 64], w3 = row[j + 96];
    const double2 r0 = rc2[j], r1 = rc2[j + 32], r2 = rc2[j + 64], r3 = rc2[j + 96];
    a0 += w0.x * r0.x + w0.y * r0.y;
    a1 += w1.x * r1.x + w1.y * r1.y;
    a2 += w2.x * r2.x + w2.y * r2.y;
    a3 += w3.x * r3.x + w3.y * r3.y;
  }
  for (; j < n2; j += 32) {
    const double2 w0 = row[j], r0 = rc2[j];
    a0 += w0.x * r0.x + w0.y * r0.y;
  }
  double acc = (a0 + a1) + (a2 + a3);
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 16, 64);
  if (part == 0) {
    const bool on = *fail == 0 && modeActive[f * kCB + m];
    const double c = on ? acc : 0.0;
    cs[m] = c * rc[static_cast<size_t>(f) * kCB + m];
    cOut[f * kCB + m] = c;
  }
  __syncthreads();
  if (tid == 0) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kCB; ++k) t += cs[k];
    dotPart[f] = t;
  }
  if (!lastBlockArrives(counter, gridDim.x, &flag)) return;
  double t = 0.0;
  for (int b = tid; b < static_cast<int>(gridDim.x); b += 256) t += dotPart[b];
  t = waveSum(t);
  if ((tid & 63) == 0) red[tid >> 6] = t;
  __syncthreads();
  if (tid == 0) pcgFinishScalars(scal, init, scal[S_RZPART] + ((red[0] + red[1]) + (red[2] + red[3])), scal[S_RR], tol2, hostMirror);
}

}  // namespace cvd
