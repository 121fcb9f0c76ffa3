// Coarse (pose-graph) level of the two-level preconditioner of the PCG solve.
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
//   * block-sparse Cholesky on a host-computed elimination plan       k_coarse_factor  (one workgroup)
//   * explicit dense inverse, 8 columns per workgroup                 k_coarse_inverse (F workgroups)
//   * c = A_c^-1 (Z^T r) per PCG iteration: one dense symmetric product k_coarse_apply
// The regularisers enter through H_ff only (their inter-frame part, the position regulariser, is left to
// the fine level), so A_c stays SPD.  Everything is deterministic (no atomics in the solves) so that the ranks
// of the pair-sharded multi-GPU mode stay bit-identical.
#pragma once

#include "cvd_kernels.h"

namespace cvd {

constexpr size_t kCoarseMaxUnknowns = 4096;  // dense inverse of A_c: n^2 doubles (128 MiB at the cap)
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
};

// ---------------------------------------------------------------------------------------------------------
// Off-diagonal coarse blocks: C_e[i][j] = sum over the constraints between frames (fa, fb), both directions,
// of rho' * sum_r (J_fa Z)[r][i] (J_fb Z)[r][j].  One workgroup per work item (k_matvec_pairs' decomposition).
// ---------------------------------------------------------------------------------------------------------
template <int KD, int KS>
__global__ __launch_bounds__(256) void k_coarse_edges(Layout L, Table T, Items it, const double* __restrict__ x,
                                                      const FrameConst* __restrict__ fc,
                                                      const int* __restrict__ itemEdge, double* __restrict__ edgeOut) {
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
  double Cacc[kCBB];  // rows: modes of fa, columns: modes of fb
#pragma unroll
  for (int i = 0; i < kCBB; ++i) Cacc[i] = 0.0;
  const bool haveScale = L.N >= 1;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fs = fcs[dir];
    const FrameConst& Ft = fcs[dir ^ 1];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    for (long long c = cb + tid; c < ce; c += 256) {
      const float2 d = T.dsrc[c];
      if (!(d.x > 0.f)) continue;
      Sample<KD, KS> s;
      evalSample<KD, KS, true>(L, Fs, Ft, xs, xt, T.ndc[c], d, s);
      // Z-projected Jacobians of the two sides: 7 pose columns + the uniform depth-scale column
      // (d r / d scale_k = JD w_k d_src and the interpolation weights sum to one)
      double Js[3][kCB], Jt[3][kCB];
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int i = 0; i < 7; ++i) { Js[r][i] = s.a.Jp[r][i]; Jt[r][i] = s.b.Jp[r][i]; }
        Js[r][7] = haveScale ? s.a.JD[r] * s.a.d : 0.0;
        Jt[r][7] = haveScale ? s.b.JD[r] * s.b.d : 0.0;
      }
      const double w = s.rho1;
      if (dir == 0) {
#pragma unroll
        for (int i = 0; i < kCB; ++i) {
          const double a0 = w * Js[0][i], a1 = w * Js[1][i], a2 = w * Js[2][i];
#pragma unroll
          for (int j = 0; j < kCB; ++j) Cacc[i * kCB + j] += a0 * Jt[0][j] + a1 * Jt[1][j] + a2 * Jt[2][j];
        }
      } else {  // source = fb, target = fa: rows (fa) take the target side
#pragma unroll
        for (int i = 0; i < kCB; ++i) {
          const double a0 = w * Jt[0][i], a1 = w * Jt[1][i], a2 = w * Jt[2][i];
#pragma unroll
          for (int j = 0; j < kCB; ++j) Cacc[i * kCB + j] += a0 * Js[0][j] + a1 * Js[1][j] + a2 * Js[2][j];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < kCBB; ++i) {
    const double v = waveSum(Cacc[i]);
    if ((tid & 63) == 0) atomicAdd(&Cs[i], v);
  }
  __syncthreads();
  // several chunk items may share one frame pair: the edge block is zeroed by the host before the launch
  if (tid < kCBB) atomicAdd(&edgeOut[static_cast<size_t>(itemEdge[item]) * kCBB + tid], Cs[tid]);
}

// ---------------------------------------------------------------------------------------------------------
// Diagonal coarse blocks D_f = Z_f^T (H_ff + diag(lam_f)) Z_f and the mode activity flags.  Inactive modes
// (masked unknowns, frames outside the range, the shared focal length) become identity rows.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coarse_diag(Layout L, const double* __restrict__ hBlocks,
                                                     const double* __restrict__ lam, const double* __restrict__ mask,
                                                     double* __restrict__ diagOut, unsigned char* __restrict__ modeActive) {
  __shared__ double u[264];  // u[r] = sum over scale vertices v of H[r][v] (+ lam on the diagonal)
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
      a += hf[static_cast<size_t>(r) * B + cidx] + (r == cidx ? lf[r] : 0.0);
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
    if (i < 7 && j < 7) v = hf[static_cast<size_t>(i) * B + j] + (i == j ? lf[i] : 0.0);
    else if (i < 7) v = u[i];
    else if (j < 7) v = u[j];
    else v = red[0] + red[1] + red[2] + red[3];
    const bool ai = active(i), aj = active(j);
    if (!(ai && aj)) v = (i == j) ? 1.0 : 0.0;
    diagOut[static_cast<size_t>(f) * kCBB + tid] = v;
    if (j == 0) modeActive[f * kCB + i] = ai ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Block-sparse Cholesky A_c = L L^T, one workgroup, left-looking by levels of the elimination plan:
//   A: every block (i, j) of the level's columns gathers  A_ij - sum_k L_ik L_jk^T   (complete: k is in a lower level)
//   B: diagonal blocks: dense 8x8 Cholesky and the inverse of its factor (Linv)
//   C: off-diagonal blocks: L_ij = (gathered) Linv_jj^T
// One wave per block, lane = (row, column) of the 8x8 block.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_coarse_factor(CoarsePlan P, const double* __restrict__ diag,
                                                        const double* __restrict__ edges,
                                                        const unsigned char* __restrict__ modeActive,
                                                        double* __restrict__ Lb, double* __restrict__ Linv,
                                                        int* __restrict__ fail) {
  __shared__ double scratch[16][kCBB];
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane >> 3, c = lane & 7;
  const int nW = blockDim.x >> 6;
  // load: diagonal blocks by position, edge blocks (masked by the mode flags, transposed if needed), fill = 0
  for (int b = wv; b < P.nBlocks; b += nW) Lb[static_cast<size_t>(b) * kCBB + lane] = 0.0;
  __syncthreads();
  for (int j = wv; j < P.F; j += nW) Lb[static_cast<size_t>(j) * kCBB + lane] = diag[static_cast<size_t>(P.order[j]) * kCBB + lane];
  for (int e = wv; e < P.nEdges; e += nW) {
    const int code = P.edgeBlk[e];
    const int b = code >> 1, tr = code & 1;
    const int fa = P.edgeFa[e], fb = P.edgeFb[e];
    // stored rows = fa, columns = fb; block (i, j) has rows = frame of position i
    const int ra = tr ? c : r, cb2 = tr ? r : c;  // element of the stored block that lands at (r, c)
    double v = edges[static_cast<size_t>(e) * kCBB + ra * kCB + cb2];
    if (!modeActive[fa * kCB + ra] || !modeActive[fb * kCB + cb2]) v = 0.0;
    Lb[static_cast<size_t>(b) * kCBB + lane] = v;
  }
  __syncthreads();
  for (int lv = 0; lv < P.nLevels; ++lv) {
    // ---- A: gather updates
    for (int q = P.lvlBlkPtr[lv] + wv; q < P.lvlBlkPtr[lv + 1]; q += nW) {
      const int b = P.lvlBlks[q];
      double acc = Lb[static_cast<size_t>(b) * kCBB + lane];
      for (int uidx = P.updPtr[b]; uidx < P.updPtr[b + 1]; ++uidx) {
        const double* A = Lb + static_cast<size_t>(P.updA[uidx]) * kCBB + r * kCB;
        const double* Bm = Lb + static_cast<size_t>(P.updB[uidx]) * kCBB + c * kCB;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < kCB; ++m) s += A[m] * Bm[m];
        acc -= s;
      }
      Lb[static_cast<size_t>(b) * kCBB + lane] = acc;
    }
    __syncthreads();
    // ---- B: diagonal blocks of the level
    for (int q = P.levelPtr[lv] + wv; q < P.levelPtr[lv + 1]; q += nW) {
      const int j = P.levelCols[q];
      double* S = scratch[wv];
      S[lane] = Lb[static_cast<size_t>(j) * kCBB + lane];
      // in-place Cholesky (lower), one wave, lane = (r, c)
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
      // inverse of the lower-triangular factor: lane c < 8 solves column c by forward substitution
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
    // ---- C: off-diagonal blocks of the level's columns: L_ij = G Linv_jj^T
    for (int q = P.lvlBlkPtr[lv] + wv; q < P.lvlBlkPtr[lv + 1]; q += nW) {
      const int b = P.lvlBlks[q];
      if (b < P.F) continue;
      const int j = P.blkCol[b];
      const double* G = Lb + static_cast<size_t>(b) * kCBB + r * kCB;
      const double* Iv = Linv + static_cast<size_t>(j) * kCBB + c * kCB;
      double s = 0.0;
#pragma unroll
      for (int m = 0; m < kCB; ++m) s += G[m] * Iv[m];
      CVD_WAVE_SYNC();  // every lane has read its row of G before the block is overwritten
      scratch[wv][lane] = s;
      CVD_WAVE_SYNC();
      Lb[static_cast<size_t>(b) * kCBB + lane] = scratch[wv][lane];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// Explicit inverse.  Workgroup jb solves L L^T X = E_jb (the 8 unit columns of position jb) with the level
// schedule, in place in its 8 rows of the dense inverse (row = coarse index of the right-hand side, column =
// coarse index of the solution entry; coarse index = frame * 8 + mode).  Gather form only: deterministic.
//   forward : Y_j = Linv_jj ( E_j - sum_{k in row(j)} L_jk Y_k )
//   backward: X_j = Linv_jj^T ( Y_j - sum_{i in col(j)} L_ij^T X_i )
// lane = (r, c): r = mode of the solution entry, c = right-hand side column.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_coarse_inverse(CoarsePlan P, const double* __restrict__ Lb,
                                                        const double* __restrict__ Linv, const int* __restrict__ posLevel,
                                                        double* __restrict__ Ainv) {
  __shared__ double scratch[4][kCBB];
  const int jb = blockIdx.x;
  const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
  const int r = lane >> 3, c = lane & 7;
  const size_t n = static_cast<size_t>(P.F) * kCB;
  double* rows = Ainv + static_cast<size_t>(P.order[jb]) * kCB * n;  // 8 rows of n
  auto at = [&](int posIdx, int rr, int cc) -> double& { return rows[static_cast<size_t>(cc) * n + P.order[posIdx] * kCB + rr]; };
  for (size_t i = tid; i < kCB * n; i += 256) rows[i] = 0.0;
  __syncthreads();
  const int lv0 = posLevel[jb];
  for (int lv = lv0; lv < P.nLevels; ++lv) {
    for (int q = P.levelPtr[lv] + wv; q < P.levelPtr[lv + 1]; q += 4) {
      const int j = P.levelCols[q];
      double acc = (j == jb && r == c) ? 1.0 : 0.0;
      for (int e = P.rowPtr[j]; e < P.rowPtr[j + 1]; ++e) {
        const int b = P.rowBlk[e];
        const int k = P.blkCol[b];
        const double* Lr = Lb + static_cast<size_t>(b) * kCBB + r * kCB;
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < kCB; ++m) s += Lr[m] * at(k, m, c);
        acc -= s;
      }
      scratch[wv][lane] = acc;
      CVD_WAVE_SYNC();
      const double* Iv = Linv + static_cast<size_t>(j) * kCBB + r * kCB;
      double y = 0.0;
#pragma unroll
      for (int m = 0; m < kCB; ++m) y += Iv[m] * scratch[wv][m * kCB + c];
      CVD_WAVE_SYNC();
      at(j, r, c) = y;
    }
    __syncthreads();
  }
  for (int lv = P.nLevels - 1; lv >= 0; --lv) {
    for (int q = P.levelPtr[lv] + wv; q < P.levelPtr[lv + 1]; q += 4) {
      const int j = P.levelCols[q];
      double acc = at(j, r, c);
      for (int e = P.colPtr[j]; e < P.colPtr[j + 1]; ++e) {
        const int b = P.F + e;
        const int i = P.blkRow[b];
        const double* Lc = Lb + static_cast<size_t>(b) * kCBB + r;  // column r of L_ij: L[m][r]
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < kCB; ++m) s += Lc[m * kCB] * at(i, m, c);
        acc -= s;
      }
      scratch[wv][lane] = acc;
      CVD_WAVE_SYNC();
      const double* Iv = Linv + static_cast<size_t>(j) * kCBB + r;  // column r of Linv: Linv[m][r]
      double xv = 0.0;
#pragma unroll
      for (int m = 0; m < kCB; ++m) xv += Iv[m * kCB] * scratch[wv][m * kCB + c];
      CVD_WAVE_SYNC();
      at(j, r, c) = xv;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------
// c = A_c^-1 rc (dense, symmetric: column access = coalesced), grid (row chunks of 256) x (kCoarseSlabs column
// slabs).  The last workgroup to arrive folds the slab partials in a fixed order, adds rc . c to r^T z and
// finishes the PCG scalars that k_cg_update left open (S_RZPART holds the block-Jacobi part of r^T z).
// ---------------------------------------------------------------------------------------------------------
constexpr int kCoarseSlabs = 16;

__global__ __launch_bounds__(256) void k_coarse_apply(int n, const double* __restrict__ Ainv,
                                                      const double* __restrict__ rc, double* __restrict__ part,
                                                      double* __restrict__ cOut, double* __restrict__ scal,
                                                      unsigned int* __restrict__ counter, const int* __restrict__ fail,
                                                      const unsigned char* __restrict__ modeActive, int init, double tol2) {
  __shared__ double rs[512];
  __shared__ double red[4];
  __shared__ int flag;
  if (!init && scal[S_DONE] != 0.0) return;
  const int tid = threadIdx.x;
  const int i = blockIdx.x * 256 + tid;
  const int slab = blockIdx.y;
  const int per = (n + kCoarseSlabs - 1) / kCoarseSlabs;
  const int j0 = slab * per, j1 = min(n, j0 + per);
  for (int j = j0 + tid; j < j1; j += 256) rs[j - j0] = rc[j];
  __syncthreads();
  if (i < n) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int j = j0;
    const double* col = Ainv + static_cast<size_t>(j0) * n + i;
    for (; j + 3 < j1; j += 4, col += 4 * static_cast<size_t>(n)) {
      a0 += col[0] * rs[j - j0];
      a1 += col[n] * rs[j + 1 - j0];
      a2 += col[2 * static_cast<size_t>(n)] * rs[j + 2 - j0];
      a3 += col[3 * static_cast<size_t>(n)] * rs[j + 3 - j0];
    }
    for (; j < j1; ++j, col += n) a0 += col[0] * rs[j - j0];
    part[static_cast<size_t>(slab) * n + i] = (a0 + a1) + (a2 + a3);
  }
  if (!lastBlockArrives(counter, gridDim.x * gridDim.y, &flag)) return;
  const bool ok = (*fail == 0);
  double dot = 0.0;
  for (int k = tid; k < n; k += 256) {
    double s = 0.0;
    for (int sl = 0; sl < kCoarseSlabs; ++sl) s += part[static_cast<size_t>(sl) * n + k];
    // inactive modes (identity rows of A_c) take no correction; a broken-down factorisation switches the level off
    if (!ok || !modeActive[k]) s = 0.0;
    cOut[k] = s;
    dot += s * rc[k];
  }
  dot = waveSum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  if (tid == 0) pcgFinishScalars(scal, init, scal[S_RZPART] + ((red[0] + red[1]) + (red[2] + red[3])), scal[S_RR], tol2);
}

}  // namespace cvd
