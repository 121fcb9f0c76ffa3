// cvd_dense_inverse.h -- inverse of ONE dense symmetric positive definite f64 matrix (the dense coarse level of the
// two-level preconditioner: A_c = Z^T (J^T J + D) Z, 8 unknowns per frame, n = 2400 at 300 frames) on the f64 matrix cores
// of the WHOLE device, as one persistent kernel.  Replaces rocSOLVER's potrf + potri (~250 dependent micro-kernels,
// 6.5 ms) on the product path.  Since round 4 the default path inverts the temporal levels' matrices with
// it (cvd_temporal.h: n = 8 x nodes = 312 and n = nodes x hats = 495 at 300 frames, 0.12 ms each); the 2400-unknown exact level
// remains as cvd_solver_options::coarse_over_budget = 1.
//
// Algorithm: the symmetric sweep operator of k_block_inverse_mfma (cvd_kernels.h), blocked with 16-wide pivot tiles, spread
// over the device.  Sweeping pivot tile k (P = G_kk^-1) maps
//     G_kk <- -P,   G_ik <- G_ik P,   G_kj <- P G_kj,   G_ij <- G_ij - G_ik P G_kj      (i, j != k)
// and after all nT = n / 16 steps G = -A^-1.  n^3 flop like potrf + potri, but ONE uniform step: every tile of the lower
// triangle receives a rank-16 update in every step, so the work is perfectly balanced and nothing shrinks.
//
// Layout: the lower-triangle 16x16 tiles live in MFMA ACCUMULATOR registers for the whole kernel (23 MB at n = 2400 =
// 90 KB per CU).  The triangle is cut into S x S super-tiles, one per workgroup (8 waves, TPW = ceil(S^2 / 8) tiles per
// wave): a workgroup's tiles then need only S row blocks and S column blocks of the pivot panel A(:, k).  S is the smallest
// value for which the super-tiles fit one workgroup per CU (n = 2400: S = 7, 253 workgroups), so every workgroup is
// resident and the grid barrier below cannot strand one.
//
// Per step k:  [grid barrier]  the 2 S panel tiles + (-P) come from the global panel buffer into LDS (agent-scope loads);
// -T_m = A(m, k) (-P) for the workgroup's row blocks (4 MFMAs each);  every tile:  G_ij += (-T_i) A(j, k)^T (4 MFMAs),
// tiles of block row / column k and the pivot tile are replaced;  then the owners of block column / row k + 1 PUBLISH
// their freshly updated tiles as the next panel (write-through stores into the other half of the double buffer) and the
// owner wave of tile (k + 1, k + 1) inverts it in-wave (invPivotStep: DPP / readlane / one bpermute per pivot) and
// publishes -P.  One grid barrier per step; the dependent chain of a step is panel load -> T -> update of the next pivot
// tile -> its 16 scalar pivots -> publish.
//
// Inter-workgroup visibility (cdna_hip_programming.md Guideline 16, recipe R1): payload stores are agent-scope relaxed
// atomic stores (write-through, sc1), every storing wave drains vmcnt(0) before the workgroup arrives at the barrier
// counter; payload loads are agent-scope relaxed atomic loads, so no fence is needed on either side.  The counter is
// monotonic (step k waits for (k + 1) x gridDim arrivals) and is zeroed by the host before the launch; every spin is
// bounded and a timeout raises `fail` and releases all workgroups.
#pragma once

#include "cvd_kernels.h"

namespace cvd {

constexpr int kDinvNW = 8;  // waves per workgroup

#ifdef CVD_DINV_PROFILE  // tools/dinv_bench.hip: shader-clock cycles per phase, wave and workgroup
__device__ unsigned long long g_dinvProf[256 * kDinvNW * 8];
#define CVD_DINV_T(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[slot] += t_ - tLast; tLast = t_; } while (0)
#else
#define CVD_DINV_T(slot) do { } while (0)
#endif

__device__ __forceinline__ void dinvStore(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double dinvLoad(const double* p) {
  return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}

// Arrive at the monotonic counter (counter[0]) and wait for `target` arrivals.  Returns false when the kernel is being
// abandoned (a workgroup timed out: counter[1] != 0, and bit 30 of *fail is set); uniform over the workgroup.
// (A split-phase variant -- arrive right after publishing, update the tiles nobody else needs, then wait -- was measured
// and is slower: a step is bound by three dependent memory round trips (drain of the write-through stores ~2.5 us, the
// counter, the agent-scope panel loads) plus the 16 scalar pivots, not by the rank-16 updates it would have hidden.)
__device__ __forceinline__ bool dinvGridBarrier(unsigned int* counter, unsigned int target, int* fail, int* ldsFlag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through payload stores have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int spins = 0;
    int ok = 1;
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      ++spins;
      if ((spins & 1023u) == 0 && __hip_atomic_load(counter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (spins > (1u << 22)) {  // ~1 s: a workgroup is not resident or has left -- abandon, never hang the device
        __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicOr(fail, 0x40000000);
        ok = 0;
        break;
      }
    }
    *ldsFlag = ok;
  }
  __syncthreads();
  return *ldsFlag != 0;
}

// A: n x n f64 row-major (lda = n), symmetric, only the lower triangle is read.  out: n x n f64, the full symmetric
// inverse.  A non-positive pivot (the coarse matrix is singular along the gauge directions up to the damping) leaves `out`
// untouched: when *outValid says it holds an earlier inverse that one stays in use -- any SPD approximation serves the
// preconditioner -- otherwise *fail = 1 (the level is switched off by its consumers).  Success sets *outValid = 1.
// Bit 30 of *fail = barrier timeout.
// panel: 2 x nT x 256 doubles, pinv: 2 x 256 doubles, barrier: three zeroed words (arrivals, abandon flag, bad pivots);
// *fail zeroed by the host.  gridDim.x = nS (nS + 1) / 2.
// (body: `bid` of `nGroupsArg` workgroups invert ONE matrix; the kernels below run it for one matrix or -- round 6 -- for two
// independent matrices in one launch, blockIdx.y = matrix, each with its own panel / barrier words)
template <int TPW>
__device__ __forceinline__ void dinvBody(int n, int S, int nS, const double* __restrict__ A, double* __restrict__ out, int* __restrict__ fail,
                                         double* __restrict__ panel, double* __restrict__ pinv, unsigned int* __restrict__ barrier,
                                         int* __restrict__ outValid, int bid, unsigned int nGroupsArg) {
  extern __shared__ __attribute__((aligned(16))) double dinvSmem[];
  __shared__ int barrierOk;
  // S^2 <= 8 TPW tiles per workgroup => (2 S + 1) x 256 panel elements over 512 threads
  constexpr int kPanelLoads = (TPW <= 2 ? 4 : TPW <= 5 ? 6 : TPW <= 8 ? 8 : TPW <= 13 ? 10 : TPW <= 18 ? 12 : 14) + 1;
  const int nT = (n + kInvTS - 1) / kInvTS;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, r0 = lane >> 4;
  // super-tile (SI >= SJ) of this workgroup
  int SI = static_cast<int>((sqrtf(8.f * static_cast<float>(bid) + 1.f) - 1.f) * 0.5f);
  while ((SI + 1) * (SI + 2) / 2 <= bid) ++SI;
  while (SI * (SI + 1) / 2 > bid) --SI;
  const int SJ = bid - SI * (SI + 1) / 2;
  const int rowBase = SI * S, colBase = SJ * S;
  double* arow = dinvSmem;                      // [S] A(rowBase + m, k)
  double* acol = arow + S * kInvTile;           // [S] A(colBase + m, k)
  double* tneg = acol + S * kInvTile;           // [S] -T of the row blocks
  double* tnegc = tneg + S * kInvTile;          // [S] -T of the column blocks (only when block row k lies in this super-tile)
  double* piv = tnegc + S * kInvTile;           // -P
  double* scratch = piv + kInvTile + w * kInvTile;  // one private tile per wave

  cvd_d4 acc[TPW];
  // tile slot s of this wave: (li, lj) within the super-tile, packed into ONE scalar register per slot (li | lj << 8, -1 =
  // empty slot): four unpacked index arrays of TPW entries spill the SGPR file
  int tCode[TPW];
#define DINV_LI(s) (tCode[s] & 0xff)
#define DINV_LJ(s) ((tCode[s] >> 8) & 0xff)
#define DINV_I(s) (tCode[s] < 0 ? -1 : rowBase + DINV_LI(s))
#define DINV_J(s) (tCode[s] < 0 ? -1 : colBase + DINV_LJ(s))
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    // Tile l = w + 8 s of the super-tile.  Off-diagonal super-tiles: row-major (li, lj).  DIAGONAL super-tiles hold the lower
    // triangle only and deal their S diagonal tiles -- the pivot tiles -- FIRST, one per wave (l = li < S), then the tiles
    // below the diagonal: the wave that has to invert the next pivot carries one diagonal tile and its share of the others
    // (row-major numbering put ALL diagonal tiles of an S = 7 super-tile on wave 0: l = 8 li).
    const int l = w + s * kDinvNW;
    int li, lj;
    if (SI != SJ) {
      li = l / S;
      lj = l - li * S;
    } else if (l < S) {
      li = l;
      lj = l;
    } else {
      const int m = l - S;  // strictly lower tiles: (1,0), (2,0), (2,1), (3,0) ...
      li = static_cast<int>((1.f + sqrtf(1.f + 8.f * static_cast<float>(m))) * 0.5f);
      while (li * (li - 1) / 2 > m) --li;
      while ((li + 1) * li / 2 <= m) ++li;
      lj = m - li * (li - 1) / 2;
    }
    int I = rowBase + li, J = colBase + lj;
    int code = li | (lj << 8);
    if (l >= (SI != SJ ? S * S : S * (S + 1) / 2) || li >= S || I >= nT || J >= nT || I < J) { I = -1; J = -1; code = -1; }
    tCode[s] = __builtin_amdgcn_readfirstlane(code);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kInvTS * (I < 0 ? 0 : I) + r0 + 4 * r, j = kInvTS * (J < 0 ? 0 : J) + c;
      const double v = A[static_cast<size_t>(min(i, n - 1)) * n + min(j, n - 1)];
      acc[s][r] = (I >= 0 && i < n && j < n) ? v : (i == j ? 1.0 : 0.0);
    }
  }

  // Slot of tile (kk, kk) in THIS wave, -1 when another wave or workgroup owns it.
  auto pivotSlot = [&](int kk) -> int {
    const int li = kk - rowBase;
    if (SI != SJ || li < 0 || li >= S || kk >= nT) return -1;
    return (li % kDinvNW == w) ? li / kDinvNW : -1;   // (diagonal tiles come first: l = li)
  };
  // The owner wave of pivot tile (kk, kk) inverts it in-wave (16 scalar symmetric sweeps) and publishes -P.  The 16 pivots
  // exist ONCE in the code, not once per tile slot.
  auto invertAndPublishPivot = [&](int kk, int slot) {
#pragma unroll
    for (int s = 0; s < TPW; ++s)
      if (s == slot) {
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
      }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // (a non-inlined function for the 16 sweeps -- own register allocation, away from the ~100 live scalars of this kernel --
    // was measured: no difference)
    const int row = lane & 15, cg = lane >> 4;
    double g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = scratch[row * kInvLd + 4 * cg + e];
    int bad = 0;
    invPivotStep<0>(g, row, cg, bad);   invPivotStep<1>(g, row, cg, bad);   invPivotStep<2>(g, row, cg, bad);
    invPivotStep<3>(g, row, cg, bad);   invPivotStep<4>(g, row, cg, bad);   invPivotStep<5>(g, row, cg, bad);
    invPivotStep<6>(g, row, cg, bad);   invPivotStep<7>(g, row, cg, bad);   invPivotStep<8>(g, row, cg, bad);
    invPivotStep<9>(g, row, cg, bad);   invPivotStep<10>(g, row, cg, bad);  invPivotStep<11>(g, row, cg, bad);
    invPivotStep<12>(g, row, cg, bad);  invPivotStep<13>(g, row, cg, bad);  invPivotStep<14>(g, row, cg, bad);
    invPivotStep<15>(g, row, cg, bad);
    if (bad && lane == 0) atomicAdd(barrier + 2, 1u);
    double* dst = pinv + static_cast<size_t>(kk & 1) * 256;
#pragma unroll
    for (int e = 0; e < 4; ++e) dinvStore(dst + row * 16 + 4 * cg + e, g[e]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // Publishes the panel step `k` will read: block column k (tiles I > k) and block row k transposed (tiles J < k).
  auto publish = [&](int k) {
    double* pk = panel + static_cast<size_t>(k & 1) * nT * 256;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      if (DINV_I(s) < 0) continue;
      if (DINV_J(s) == k && DINV_I(s) > k) {
        double* dst = pk + static_cast<size_t>(DINV_I(s)) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) dinvStore(dst + (r0 + 4 * r) * 16 + c, acc[s][r]);
      } else if (DINV_I(s) == k && DINV_J(s) < k) {
        // A(j, k) = A(k, j)^T: through the private LDS tile so that the global stores run along rows
#pragma unroll
        for (int r = 0; r < 4; ++r) scratch[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double* dst = pk + static_cast<size_t>(DINV_J(s)) * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) dinvStore(dst + (r0 + 4 * r) * 16 + c, scratch[c * kInvLd + r0 + 4 * r]);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
  };

#ifdef CVD_DINV_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tLast = __builtin_amdgcn_s_memtime();
#endif
  {
    const int ps = pivotSlot(0);
    if (ps >= 0) invertAndPublishPivot(0, ps);
  }
  publish(0);
  CVD_DINV_T(0);
  const unsigned int nGroups = nGroupsArg;
  bool alive = true;
  for (int k = 0; k < nT; ++k) {
    if (!dinvGridBarrier(barrier, static_cast<unsigned int>(k + 1) * nGroups, fail, &barrierOk)) { alive = false; break; }
    CVD_DINV_T(1);
    // ---- the panel tiles of this super-tile's row and column blocks, and -P, into LDS.  ALL of a thread's loads are issued
    // before the first is used: written as load-then-store per element the agent-scope loads ran one global round trip after
    // the other (8 per step, most of a step's duration)
    {
      const double* pk = panel + static_cast<size_t>(k & 1) * nT * 256;
      const double* pp = pinv + static_cast<size_t>(k & 1) * 256;
      const int total = (2 * S + 1) * 256;
      double v[kPanelLoads];
      int dstOff[kPanelLoads];
#pragma unroll
      for (int u = 0; u < kPanelLoads; ++u) {
        const int e = threadIdx.x + u * (kDinvNW * 64);
        const int t = e >> 8, q = e & 255, rr = q >> 4, cc = q & 15;
        const double* src = pp + q;                         // (always a valid address: the loads are unconditional)
        int off = (4 * S) * kInvTile + rr * kInvLd + cc;    // piv
        bool zero = false;
        if (t < 2 * S) {
          const int m = t < S ? t : t - S;
          const int blk = (t < S ? rowBase : colBase) + m;
          off = (t < S ? 0 : S * kInvTile) + m * kInvTile + rr * kInvLd + cc;   // arow / acol
          src = pk + static_cast<size_t>(min(blk, nT - 1)) * 256 + q;
          zero = blk >= nT || blk == k;                     // (no such panel tile: its slot reads as zero)
        }
        dstOff[u] = e < total ? off : -1;
        const double raw = dinvLoad(src);
        v[u] = zero ? 0.0 : raw;
      }
#pragma unroll
      for (int u = 0; u < kPanelLoads; ++u)
        if (dstOff[u] >= 0) dinvSmem[dstOff[u]] = v[u];
    }
    __syncthreads();
    CVD_DINV_T(2);
    // ---- -T_m = A(m, k) (-P): the row blocks, and the column blocks when block row k lies in this super-tile
    const bool hasRowK = k >= rowBase && k < rowBase + S;
    for (int m = w; m < (hasRowK ? 2 * S : S); m += kDinvNW) {
      const double* src = (m < S ? arow + m * kInvTile : acol + (m - S) * kInvTile);
      cvd_d4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(src[c * kInvLd + 4 * kk + r0], piv[(4 * kk + r0) * kInvLd + c], t, 0, 0, 0);
      double* dst = (m < S ? tneg + m * kInvTile : tnegc + (m - S) * kInvTile);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(r0 + 4 * r) * kInvLd + c] = t[r];
    }
    __syncthreads();
    CVD_DINV_T(3);
    // ---- rank-16 update G_ij += (-T_i) A(j, k)^T of every owned tile; tiles of block row / column k (they read zeroed panel
    // slots) and the pivot tile are replaced
    const int laneOp = c * kInvLd + r0;
    // The next pivot tile FIRST: its owner wave updates it, inverts it and issues the -P stores before anything else, so
    // that the stores' way to memory (~2.5 us until the drain in the grid barrier returns) overlaps this wave's other tiles.
    const int ps = pivotSlot(k + 1);   // wave-uniform
    if (ps >= 0) {
#pragma unroll
      for (int s = 0; s < TPW; ++s)
        if (s == ps) {
          const double* ta = tneg + DINV_LI(s) * kInvTile + laneOp;
          const double* pb = acol + DINV_LJ(s) * kInvTile + laneOp;
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[4 * kk], pb[4 * kk], acc[s], 0, 0, 0);
        }
      invertAndPublishPivot(k + 1, ps);
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      const int I = DINV_I(s), J = DINV_J(s);
      if (I < 0 || s == ps) continue;   // wave-uniform
      const int oa = DINV_LI(s) * kInvTile, ob = DINV_LJ(s) * kInvTile;
      if (I != k && J != k) {
        const double* ta = tneg + oa + laneOp;
        const double* pb = acol + ob + laneOp;
        double a[4], b[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { a[kk] = ta[4 * kk]; b[kk] = pb[4 * kk]; }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc[s], 0, 0, 0);
      } else if (I == k && J == k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = piv[(r0 + 4 * r) * kInvLd + c];
      } else if (J == k) {  // i > k: G_ik <- T_i
        const double* src = tneg + oa;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -src[(r0 + 4 * r) * kInvLd + c];
      } else {              // j < k: G_kj <- T_j^T
        const double* src = tnegc + ob;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -src[c * kInvLd + r0 + 4 * r];
      }
    }
    CVD_DINV_T(4);
    if (k + 1 < nT) publish(k + 1);
    CVD_DINV_T(5);
  }
  if (!alive) return;
  // (every pivot was inverted before the last barrier: the count is final)
  const unsigned int nBad = __hip_atomic_load(barrier + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (nBad != 0u) {
    if (bid == 0 && threadIdx.x == 0 && *outValid == 0) atomicOr(fail, 1);
    return;
  }
  if (bid == 0 && threadIdx.x == 0) *outValid = 1;

  // A^-1 = -G: the tile as it lies and its mirror image (transposed through the private LDS tile); a diagonal tile is
  // averaged with its own transpose (its two halves were updated independently and differ in the last bits), so that the
  // stored inverse is exactly symmetric
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    if (DINV_I(s) < 0) continue;
    const bool diagTile = DINV_I(s) == DINV_J(s);
#pragma unroll
    for (int r = 0; r < 4; ++r) scratch[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kInvTS * DINV_I(s) + r0 + 4 * r, j = kInvTS * DINV_J(s) + c;
      const double t = scratch[c * kInvLd + r0 + 4 * r];  // element (c, r0 + 4 r) of the tile
      if (i < n && j < n) out[static_cast<size_t>(i) * n + j] = diagTile ? -0.5 * (acc[s][r] + t) : -acc[s][r];
      const int it = kInvTS * DINV_I(s) + c, jt = kInvTS * DINV_J(s) + r0 + 4 * r;  // element (it, jt) -> out[jt][it]
      if (!diagTile && it < n && jt < n) out[static_cast<size_t>(jt) * n + it] = -t;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
#ifdef CVD_DINV_PROFILE
  CVD_DINV_T(6);
  if (lane == 0 && bid < 256)
    for (int q = 0; q < 8; ++q) g_dinvProf[(bid * kDinvNW + w) * 8 + q] = prof[q];
#endif
}

template <int TPW>
inline __global__ __launch_bounds__(kDinvNW * 64) void k_dense_spd_inverse(int n, int S, int nS, const double* __restrict__ A,
                                                                           double* __restrict__ out, int* __restrict__ fail,
                                                                           double* __restrict__ panel, double* __restrict__ pinv,
                                                                           unsigned int* __restrict__ barrier, int* __restrict__ outValid) {
  dinvBody<TPW>(n, S, nS, A, out, fail, panel, pinv, barrier, outValid, static_cast<int>(blockIdx.x), gridDim.x);
}

// Two independent matrices in one launch (the two temporal levels of the preconditioner: 312 and 495 unknowns at 300 frames, 123 + 160 us
// one after the other).  Their pivot chains are sequential in themselves and independent of each other: side by side they take the
// longer one's time.  Every workgroup of BOTH must be resident (the launcher checks the sum against the device).
struct DinvJob {
  int n, S, nS, groups;
  const double* A;
  double* out;
  int* fail;
  double* panel;
  double* pinv;
  unsigned int* barrier;
  int* outValid;
};
template <int TPW>
inline __global__ __launch_bounds__(kDinvNW * 64) void k_dense_spd_inverse_pair(DinvJob j0, DinvJob j1) {
  const DinvJob& J = blockIdx.y ? j1 : j0;
  if (static_cast<int>(blockIdx.x) >= J.groups) return;   // (workgroup-uniform; the job's barrier counts J.groups arrivals)
  dinvBody<TPW>(J.n, J.S, J.nS, J.A, J.out, J.fail, J.panel, J.pinv, J.barrier, J.outValid, static_cast<int>(blockIdx.x),
                static_cast<unsigned int>(J.groups));
}

#undef DINV_LI
#undef DINV_LJ
#undef DINV_I
#undef DINV_J

}  // namespace cvd
