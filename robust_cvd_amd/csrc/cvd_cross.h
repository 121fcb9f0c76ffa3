// robust_cvd_amd/csrc/cvd_cross.h
//
// DENSE MODE with explicit cross blocks (reference lib/FlowConstraints.cpp:315-329,381-465 with
// matchSeparation = 0: every masked in-bounds pixel of a frame pair is a constraint).
//
// The matrix-free product re-derives every constraint's Jacobian row in every PCG iteration.  That is the right trade for
// the SAMPLED constraint lists (~600 constraints per frame pair against B^2 = 31 k block entries), and the wrong one in
// dense mode: ~81 k pixel constraints per directed pair, 83 products per LM iteration.  Here the off-diagonal blocks
//     X_ab = sum over the constraints of {a -> b, b -> a} of rho' J_a^T J_b          (B x B doubles, a < b)
// are assembled ONCE per Jacobian evaluation (cvd_dense_walk.h) and a product streams them (k_cross_matvec: 2 B^2 flop
// per 8 B^2 bytes -- HBM-bound); the frame-diagonal part of J^T J p comes from H_ff, which the assembly kernels form anyway
// (k_matvec_finish, Hdiag).  Scope: the fast kernels' (identity spatial transform, reprojection losses, one value
// parameter per vertex, bilinear grids, Fixed / PerFrame intrinsics), single GPU, no triplets.
//
// The blocks are assembled by the one-walk kernels of cvd_dense_walk.h (round 6; rounds 2-5 had two more walks over the pixels
// here, k_cross_assemble<.., GRID>).  This header keeps their consumers: the product, the coarse edge blocks.
#pragma once

#include "cvd_kernels.h"

namespace cvd {

// Undirected frame pairs of the explicit-block mode: ranges of both directions' pixel slots (either may be empty) and the
// two rows of the partial-product buffer.
struct CrossPairs {
  const int* fa;            // fa < fb
  const int* fb;
  const long long* range;   // 4 per pair: [a -> b begin, end, b -> a begin, end) pixel slots
  const int* slot;          // 2 per pair: rows of the partial buffer (frame-major)
  int count;
};

constexpr int kCrossThreads = 512;
// q rows of one undirected pair from its block: y_a = X p_b, y_b = X^T p_a with p = (z + Z c + beta p_old) * mask (the
// search direction, formed here exactly as the matrix-free product forms it).  Each wave streams its rows once, fully
// coalesced: a row's dot product with p_b gives y_a[row], the same loads scaled by p_a[row] accumulate y_b per column.
// Workgroups beyond the pairs (diagSlot != nullptr: one per frame) stream the frame's own block: y_f = H_ff p_f into a row of its own.
// (Until round 6 the finish half of the PCG tail formed H_ff p_f -- 75 MB of the same streaming, but inside a latency chain of one
// workgroup per frame in front of a grid barrier: the dense mode's tail took 58 us against the list mode's 28.)
inline __global__ __launch_bounds__(kCrossThreads) void k_cross_matvec(Layout L, CrossPairs cp, const double* __restrict__ X,
                                                                const double* __restrict__ Hd, const int* __restrict__ diagSlot,
                                                                const double* __restrict__ mask, const double* __restrict__ z,
                                                                const double* __restrict__ pOld, const double* __restrict__ scal,
                                                                int useBeta, double* __restrict__ qPart, CoarseView V) {
  const double sDone = scal[S_DONE];
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kCrossThreads / 64;
  double* pa = sm;
  double* pb = pa + B;
  double* ya = pb + B;
  double* ybw = ya + B;          // NW x B partial column sums
  double* cl = ybw + NW * B;     // 2 x kCB coarse corrections
  double* tls = cl + 2 * kCB;    // third level (CoarseView::tl): the two frames' coefficients, 2 x tlS
  const bool diag = pair >= cp.count;   // (uniform)
  const int fa = diag ? pair - cp.count : cp.fa[pair], fb = diag ? fa : cp.fb[pair];
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  const bool tlOn = V.tl != nullptr;
  if (tid < 2 * kCB) cl[tid] = (V.cF != nullptr) ? V.cF[(tid < kCB ? fa : fb) * kCB + (tid & (kCB - 1))] : 0.0;
  if (tlOn)
    for (int i = tid; i < 2 * V.tlS; i += kCrossThreads) {
      const int which = i >= V.tlS ? 1 : 0;
      tls[i] = V.tl[static_cast<size_t>(which ? fb : fa) * V.tlS + (i - which * V.tlS)];
    }
  __syncthreads();
  if (sDone != 0.0) return;
  for (int i = tid; i < B; i += kCrossThreads) {
    const size_t ia = static_cast<size_t>(fa) * B + i, ib = static_cast<size_t>(fb) * B + i;
    double ca = coarseAtLds(cl, L, i), cb2 = coarseAtLds(cl + kCB, L, i);
    if (tlOn) {
      TlTaps tt;
      tlLoadTaps(V, i, L.nD, tt);
      ca += tlAt(tls, tt);
      cb2 += tlAt(tls + V.tlS, tt);
    }
    pa[i] = (z[ia] + ca + (useBeta ? beta * pOld[ia] : 0.0)) * mask[ia];
    pb[i] = (z[ib] + cb2 + (useBeta ? beta * pOld[ib] : 0.0)) * mask[ib];
  }
  __syncthreads();
  const double* Xp = diag ? Hd + static_cast<size_t>(fa) * B * B : X + static_cast<size_t>(pair) * B * B;
  constexpr int NC = 4;  // column chunks of 64 per lane: B <= 256
  double cb[NC] = {0.0, 0.0, 0.0, 0.0};
  double pbv[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) pbv[k] = (lane + 64 * k < B) ? pb[lane + 64 * k] : 0.0;
  for (int r0 = wave; r0 < B; r0 += 2 * NW) {
    // two rows per trip: their loads are in flight together
    const int r1 = r0 + NW;
    double xv0[NC], xv1[NC];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const int j = lane + 64 * k;
      xv0[k] = (j < B) ? Xp[static_cast<size_t>(r0) * B + j] : 0.0;
      xv1[k] = (j < B && r1 < B) ? Xp[static_cast<size_t>(r1) * B + j] : 0.0;
    }
    const double a0 = pa[r0], a1 = (r1 < B) ? pa[r1] : 0.0;
    double d0 = 0.0, d1 = 0.0;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      d0 += xv0[k] * pbv[k];
      d1 += xv1[k] * pbv[k];
      if (!diag) cb[k] += xv0[k] * a0 + xv1[k] * a1;   // (uniform; a frame's own block is symmetric: its row products are everything)
    }
    d0 = waveSum(d0);
    d1 = waveSum(d1);
    if (lane == 0) {
      ya[r0] = d0;
      if (r1 < B) ya[r1] = d1;
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k)
    if (lane + 64 * k < B) ybw[wave * B + lane + 64 * k] = cb[k];
  __syncthreads();
  if (diag) {   // (symmetric block: the row products are the whole answer)
    double* outD = qPart + static_cast<size_t>(diagSlot[fa]) * B;
    for (int i = tid; i < B; i += kCrossThreads) outD[i] = ya[i];
    return;
  }
  double* outA = qPart + static_cast<size_t>(cp.slot[pair * 2]) * B;
  double* outB = qPart + static_cast<size_t>(cp.slot[pair * 2 + 1]) * B;
  for (int i = tid; i < B; i += kCrossThreads) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += ybw[w * B + i];
    outA[i] = ya[i];
    outB[i] = s;
  }
}

// Coarse edge blocks of the two-level preconditioner from the explicit blocks: E_ab = Z_a^T X_ab Z_b with Z = [I_7 0; 0 1]
// (the 8th mode moves every depth-scale vertex together), i.e. the pose x pose corner, row / column sums over the grid part
// and its total -- a reduction of the 250 KB block instead of a third walk over the pair's pixels (k_coarse_edges_fast:
// 9.9 ms per rebuild at 300 frames).  One workgroup per pair; rows coalesced, one wave per row.  Stored rows = fa, columns = fb
// like k_coarse_edges; `pairEdge` < 0: the pair has no edge block.
inline __global__ __launch_bounds__(256) void k_coarse_edges_cross(Layout L, CrossPairs cp, const double* __restrict__ X,
                                                            const int* __restrict__ pairEdge, double* __restrict__ edgeOut) {
  __shared__ double rowG[256];      // sum over the grid columns of row r
  __shared__ double colP[4][8];     // per wave: sum over the grid rows of pose column j
  const int B = L.B;
  const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int edge = pairEdge[pair];
  if (edge < 0) return;
  const double* Xp = X + static_cast<size_t>(pair) * B * B;
  double cp0 = 0.0;  // lanes 0..6: column sums over the rows >= 7 handled by this wave
  for (int r = wave; r < B; r += 4) {
    double s = 0.0;
    for (int j = lane; j < B; j += 64) {
      const double v = Xp[static_cast<size_t>(r) * B + j];
      if (j >= 7) s += v;
      else if (r >= 7) cp0 += v;  // (j == lane < 7 here)
    }
    s = waveSum(s);
    if (lane == 0) rowG[r] = s;
  }
  if (lane < 8) colP[wave][lane] = lane < 7 ? cp0 : 0.0;
  __syncthreads();
  if (tid < kCBB) {
    const int i = tid >> 3, j = tid & 7;
    double v;
    if (i < 7 && j < 7) v = Xp[static_cast<size_t>(i) * B + j];
    else if (i < 7) v = rowG[i];
    else if (j < 7) v = (colP[0][j] + colP[1][j]) + (colP[2][j] + colP[3][j]);
    else {
      v = 0.0;
      for (int r = 7; r < B; ++r) v += rowG[r];
    }
    edgeOut[static_cast<size_t>(edge) * kCBB + tid] = v;
  }
}

}  // namespace cvd
