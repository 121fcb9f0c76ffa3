// cvd_kernels.h -- HIP kernels of the geometric-consistency optimizer (gfx950).
//
// Kernel map (DESIGN.md has the roofline of each):
//   k_frame_consts     per frame   : R(w), dR/dw, t, fy of the evaluation point
//   k_build_table      per constr. : Observation ctor of the reference (NDC + truncating depth fetch)
//   k_cost_items       pair-major  : sum of rho(|r|^2) of the static constraints (candidate-point cost)
//   k_cost_frames      per frame   : regulariser cost
//   k_assemble         frame-major : gradient J^T r and the frame-diagonal blocks of J^T J (+ cost)
//   k_lm_diag          per unknown : Jacobi scale, LM damping (Ceres LevenbergMarquardtStrategy)
//   k_block_inverse    per frame   : (H_ff + damping)^-1 by an LDS Cholesky (block-Jacobi preconditioner)
//   k_matvec_pairs     pair-major  : partial q = J^T (rho' J p), matrix free            <- the hot kernel
//   k_matvec_finish    per frame   : reduce partials + regulariser Hessian + damping, p.q partial dots
//   k_cg_update        per frame   : x += alpha p, r -= alpha q, z = M^-1 r, partial dots
//   k_cg_scalars       1 block     : alpha/beta bookkeeping on the device (no host round trip)
#pragma once

#include "cvd_device.h"

namespace cvd {

// Compiled constraint table (the reference rebuilds `Observation`s for every solve,
// lib/PoseOptimizer.cpp:1185-1193; here it is 24 B per constraint, pair-major, resident in HBM).
struct Table {
  const float4* ndc;    // (ndc_a.xy, ndc_b.xy), f32 exactly as the reference computes them
  const float2* dsrc;   // (d_a, d_b) source depths; d_a <= 0 marks a skipped constraint
  const int* pairA;     // per pair
  const int* pairB;
  const long long* pairOff;  // P + 1
  // Dense mode (cvd_set_pair_flows: the reference's matchSeparation = 0 regime, lib/FlowConstraints.cpp:315-329, 381-465 --
  // every masked pixel whose flow target rounds into the image is a constraint): there is NO table; constraint slot c of
  // pair p is pixel c - pairOff[p] (pairOff[p] = p * W * H) and the kernels read flow / mask / depth directly,
  // 8 + 1 + 4 + 4 = 17 B per pixel pair (SURVEY.md 8d "dense mode").
  const float2* flow;           // [P][H][W] pixels
  const unsigned char* fmask;   // [P][H][W]
  const float* depth;           // [F][H][W] source depth
  int W, H;
  float sx, sy, invAspect;      // loc = pixel * (1 / W, invAspect / H), float as in the reference (:371)
};

// Dense mode: everything about a pixel's constraint that follows from its position and flow vector `f` -- ONE statement of the
// reference's float arithmetic for every kernel that reads the images (product, cost, walk, grid x grid, list materialisation):
// candidate test of FlowConstraintsCollection::compute (lib/FlowConstraints.cpp:436-460: target int(x + flow + 0.5) in bounds),
// constraint scaling (:371), the Observation constructor's pixel-edge NDC and truncating depth fetch (lib/PoseOptimizer.cpp:104-116).
// loc = (source x, y, target x, y) in [0, 1] x [0, invAspect]; n = ndc of both end points; ai / bi = the pixels whose source depths
// the two observations read (not always the constraint's own pixel: the fetch truncates loc * raster).  false: no candidate.
__device__ __forceinline__ int densePixelOfLoc(const Table& T, float lx, float ly) {
  int px = static_cast<int>(__fmul_rn(lx, static_cast<float>(T.W)));
  int py = static_cast<int>(__fmul_rn(__fdiv_rn(ly, T.invAspect), static_cast<float>(T.H)));
  px = min(max(px, 0), T.W - 1);
  py = min(max(py, 0), T.H - 1);
  return py * T.W + px;
}
__device__ __forceinline__ bool densePixelGeometry(const Table& T, int ix, int iy, float2 f, float4& loc, float4& n, int& ai, int& bi) {
  const float fx1 = __fadd_rn(static_cast<float>(ix), f.x), fy1 = __fadd_rn(static_cast<float>(iy), f.y);
  if (!(isfinite(fx1) && isfinite(fy1))) return false;
  const int ix1 = static_cast<int>(__fadd_rn(fx1, 0.5f)), iy1 = static_cast<int>(__fadd_rn(fy1, 0.5f));
  if (ix1 < 0 || ix1 >= T.W || iy1 < 0 || iy1 >= T.H) return false;
  const float lx0 = __fmul_rn(static_cast<float>(ix), T.sx), ly0 = __fmul_rn(static_cast<float>(iy), T.sy);
  const float lx1 = __fmul_rn(fx1, T.sx), ly1 = __fmul_rn(fy1, T.sy);
  loc = make_float4(lx0, ly0, lx1, ly1);
  n.x = __fadd_rn(-1.f, __fmul_rn(2.f, lx0));
  n.y = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly0), T.invAspect));
  n.z = __fadd_rn(-1.f, __fmul_rn(2.f, lx1));
  n.w = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, ly1), T.invAspect));
  ai = densePixelOfLoc(T, lx0, ly0);
  bi = densePixelOfLoc(T, lx1, ly1);
  return true;
}
// ... + the two depths, REQUESTED only (a kernel that runs its loads ahead tests the values when it uses them)
__device__ __forceinline__ bool denseConstraintRequest(const Table& T, int pix, int fa, int fb, float2 f, float4& n, float2& d) {
  const int iy = pix / T.W, ix = pix - iy * T.W;
  float4 loc;
  int ai, bi;
  d = make_float2(0.f, 0.f);
  if (!densePixelGeometry(T, ix, iy, f, loc, n, ai, bi)) return false;
  const size_t fs = static_cast<size_t>(T.W) * T.H;
  d.x = T.depth[fa * fs + ai];
  d.y = T.depth[fb * fs + bi];
  return true;
}
__device__ __forceinline__ bool denseDepthsValid(float2 d) { return isfinite(d.x) && d.x > 0.f && isfinite(d.y) && d.y > 0.f; }
// ... and tested (lib/PoseOptimizer.cpp:1190-1193: an invalid depth skips the whole constraint)
__device__ __forceinline__ bool denseConstraintFromFlow(const Table& T, long long c, long long pixBase, int fa, int fb, float2 f,
                                                        float4& n, float2& d) {
  return denseConstraintRequest(T, static_cast<int>(c - pixBase), fa, fb, f, n, d) && denseDepthsValid(d);
}

// One constraint of the pair-major / frame-major fast kernels: (ndc of both end points, the two source depths); false =
// skipped.  DENSE = false: the compiled 24 B table entry.  DENSE = true: built on the fly from the flow / mask / depth
// images with the reference's float arithmetic -- candidate test of FlowConstraintsCollection::compute (reference
// lib/FlowConstraints.cpp:436-460: mask, target pixel int(x + flow + 0.5) in bounds), constraint scaling (:371), then the
// Observation constructor (lib/PoseOptimizer.cpp:104-116, identical to k_build_table).  fa / fb = source / target frame,
// pixBase = first slot of the pair.
template <bool DENSE>
__device__ __forceinline__ bool loadConstraint(const Table& T, long long c, long long pixBase, int fa, int fb, float4& n,
                                               float2& d) {
  if constexpr (!DENSE) {
    // (both records are requested at once: testing d first made the 16-byte load a SECOND dependent round trip in every trip
    // of the constraint loops; skipped entries are rare and the 16 bytes of one are free)
    d = T.dsrc[c];
    n = T.ndc[c];
    return d.x > 0.f;
  } else {
    if (!T.fmask[c]) return false;
    return denseConstraintFromFlow(T, c, pixBase, fa, fb, T.flow[c], n, d);
  }
}

// Walk over table records c, c + step, ... < end with the NEXT record in flight (list mode; the dense mode's on-the-fly
// constraints go through loadConstraint).  Loaded at the top of its own trip a record is a global round trip -- two,
// dependent, where the compiler sinks the 16-byte half below the validity test of the 8-byte half -- that only the other
// waves of the SIMD can hide; requested one trip ahead (6 registers) it is back before it is needed (hot product
// 50.2 -> 48.6 us, 1000 frames: 213 -> 195 us).
template <bool DENSE>
struct RecordStream {
  float4 ndNext;
  float2 dNext;
  // records base + i, base + i + step, ... (i < n); base is wave-uniform: 32-bit offsets from a scalar base address
  __device__ __forceinline__ void prime(const Table& T, long long base, int i, int n) {
    ndNext = make_float4(0.f, 0.f, 0.f, 0.f);
    dNext = make_float2(0.f, 0.f);
    if (i < n) { dNext = (T.dsrc + base)[i]; ndNext = (T.ndc + base)[i]; }
  }
  // record base + i (false: skipped); requests record base + i + step
  __device__ __forceinline__ bool take(const Table& T, long long base, int i, int step, int n, long long, int, int, float4& nd,
                                       float2& d) {
    nd = ndNext;
    d = dNext;
    if (i + step < n) { dNext = (T.dsrc + base)[i + step]; ndNext = (T.ndc + base)[i + step]; }
    return d.x > 0.f;
  }
};
// Dense mode: a constraint is a chain mask byte -> flow vector -> the two depths (the target's address depends on the
// flow).  Mask and flow of the NEXT pixel are requested one trip ahead (3 registers): one dependent round trip per pixel --
// the depths -- instead of three.
template <>
struct RecordStream<true> {
  float2 fNext;
  unsigned int mNext;
  __device__ __forceinline__ void prime(const Table& T, long long base, int i, int n) {
    fNext = make_float2(0.f, 0.f);
    mNext = 0u;
    if (i < n) { mNext = (T.fmask + base)[i]; fNext = (T.flow + base)[i]; }
  }
  __device__ __forceinline__ bool take(const Table& T, long long base, int i, int step, int n, long long pixBase, int fa, int fb,
                                       float4& nd, float2& d) {
    const unsigned int m = mNext;
    const float2 f = fNext;
    if (i + step < n) { mNext = (T.fmask + base)[i + step]; fNext = (T.flow + base)[i + step]; }
    if (!m) return false;
    return denseConstraintFromFlow(T, base + i, pixBase, fa, fb, f, nd, d);
  }
};

// The same two pixels ahead (the candidate-cost pass of the dense mode, whose trip is short): mask and flow of pixel i + 2 step are
// requested while the two depths of pixel i + step -- whose address follows from its flow -- are, and pixel i is worked on from
// registers.  One pixel ahead, every trip took the depths' round trip (1.9 ms per pass over 152 M pixel slots).
struct DenseStreamAhead {
  float2 fNext;        // pixel i + step: its flow and mask are here, its depths not yet requested
  unsigned int mNext;
  float4 ndCur;        // pixel i: candidate, ndc, depths (requested; validity is tested when they are used)
  float2 dCur;
  bool candCur;
  __device__ __forceinline__ bool request(const Table& T, int pix, int fa, int fb, float2 f, float4& n, float2& d) const {
    return denseConstraintRequest(T, pix, fa, fb, f, n, d);
  }
  __device__ __forceinline__ void prime(const Table& T, long long base, int i, int step, int n, long long pixBase, int fa, int fb) {
    unsigned int m0 = 0u;
    float2 f0 = make_float2(0.f, 0.f);
    fNext = make_float2(0.f, 0.f);
    mNext = 0u;
    if (i < n) { m0 = (T.fmask + base)[i]; f0 = (T.flow + base)[i]; }
    if (i + step < n) { mNext = (T.fmask + base)[i + step]; fNext = (T.flow + base)[i + step]; }
    ndCur = make_float4(0.f, 0.f, 0.f, 0.f);
    candCur = m0 != 0u && request(T, static_cast<int>(base - pixBase) + i, fa, fb, f0, ndCur, dCur);
  }
  __device__ __forceinline__ bool take(const Table& T, long long base, int i, int step, int n, long long pixBase, int fa, int fb,
                                       float4& nd, float2& d) {
    nd = ndCur;
    d = dCur;
    const bool cand = candCur;
    const unsigned int m1 = mNext;
    const float2 f1 = fNext;
    mNext = 0u;
    if (i + 2 * step < n) { mNext = (T.fmask + base)[i + 2 * step]; fNext = (T.flow + base)[i + 2 * step]; }
    candCur = m1 != 0u && request(T, static_cast<int>(base - pixBase) + i + step, fa, fb, f1, ndCur, dCur);
    return cand && denseDepthsValid(d);
  }
};

// Valid constraints of the dense mode (what k_build_table counts for the list mode).
inline __global__ void k_dense_count(Table T, int P, const unsigned char* __restrict__ inRange, unsigned long long* __restrict__ nValid) {
  const long long npx = static_cast<long long>(T.W) * T.H;
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool ok = false;
  if (c < npx * P) {
    const int p = static_cast<int>(c / npx);
    const int fa = T.pairA[p], fb = T.pairB[p];
    float4 n;
    float2 d;
    ok = inRange[fa] && inRange[fb] && fa != fb && loadConstraint<true>(T, c, static_cast<long long>(p) * npx, fa, fb, n, d);
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(nValid, static_cast<unsigned long long>(__popcll(b)));
}

// Work items of the pair-major kernels: one UNDIRECTED frame pair {fa < fb} with a chunk of the constraints
// of the directed pair fa->fb (range[0..1]) and of fb->fa (range[2..3]); either may be empty.  Both directions
// share the two frames' parameter blocks, so one workgroup prologue / epilogue serves both.
struct Items {
  const int* fa;
  const int* fb;
  const long long* range;  // 4 per item
  const int* slot;         // 2 per item: rows of the partial-product buffer (frame-major, see k_matvec_finish)
  int count;
};

// Work list of k_assemble_fast: a frame's (pair, side) entries are cut into units of at most kAsmUnit constraints
// and the units into parts (one workgroup each) of bounded size, so that frames with many pairs (the long-range
// levels of the hierarchical flow list) do not set the kernel's duration.  Parts of a split frame publish their
// packed partial blocks and the last one to arrive folds them (lastBlockArrives on a per-frame counter).
constexpr int kDenseFramesPerGroup = 2;  // frames per dense-level workgroup of k_cg_update (DenseStep)
constexpr int kAsmUnit = 128;
// Dense mode: 1024 pixel slots per unit, dealt to the 64 lanes in runs of 16 consecutive pixels (see kDenseRun)
constexpr int kAsmUnitDense = 1024;
// Dense mode lane mapping.  Consecutive pixels fall into the same grid cell, i.e. hit the same four vertices: with lane =
// pixel every LDS atomic of a wave would collide 64-fold.  Each lane therefore walks its own run of kDenseRun = 16
// consecutive pixels (two 64-byte lines of flow per lane: the images are still streamed once).  Measured on the three
// assembly kernels together: runs of 32 pixels 25.6 ms, 16: 22.5 ms, 8: 22.7 ms, 64: 28.6 ms (150 frames).
constexpr int kDenseRun = 16;

// ---- deterministic build (lib/libcvd_hip_det.so: robust_cvd_amd.build.build_deterministic; tests/test_gpu_determinism.py) ----
// The solver kernels accumulate through LDS f64 atomics from several waves of a workgroup (grid columns of the pair-major
// product, the packed frame blocks of the assembly, the regulariser rows of the PCG tail, the Galerkin blocks of the levels):
// the order in which the waves' atomics land is a matter of timing, so two runs of one solve differ in the last bits, and the
// LM / PCG stopping rules amplify that to +-1 PCG iteration (VERDICT r4 Missing #3).  With CVD_DETERMINISTIC = 1 every such
// accumulation is issued by ONE wave (the kernels run with 64-thread workgroups, or their first wave alone walks the
// constraints) and every fold of partial results runs in index order: a wave's LDS atomics execute in program order and the
// hardware resolves same-address lanes of one instruction in a fixed order, so the sums are reproducible bit for bit.  Several
// times slower; for tests and for telling a race from rounding -- the product build keeps the concurrent form.
#ifndef CVD_DETERMINISTIC
#define CVD_DETERMINISTIC 0
#endif
constexpr int kAtomicWalkers256 = CVD_DETERMINISTIC ? 64 : 256;  // threads of a 256-thread workgroup that walk constraints into LDS atomics
struct AsmPart {
  int frame;
  int u0, u1;   // unit range
  int part;     // index within the frame
  int nParts;   // 1: the frame is not split
  int slot0;    // first partial-block slot of the frame (split frames only)
};
struct AsmWork {
  const AsmPart* parts;      // sorted by descending size (longest first)
  const int2* units;         // {pair * 2 + side, first constraint relative to the pair's offset}
  double* scratch;           // partial blocks: slots x (B(B+1)/2 + B + 2) doubles
  unsigned int* counters;    // one per frame, zero between launches
};

// Dense mode, one-walk assembly (cvd_dense_walk.h): k_dense_walk leaves one compact RECORD per directed pair (layout: that
// header); k_assemble_fast<.., FOLD> sums a frame's records into H_ff / g / cost through this view.
__host__ __device__ inline int dwRecordDoubles(int G) { return 256 + 40 * G + 8; }
struct DenseRecords {
  const double* records;
  const int* recOff;     // per directed pair: its records [recOff[p], recOff[p + 1])
  const int* fpOff;      // per frame: entries of fpList
  const int* fpList;     // directed pair * 2 + side (0: the frame is the pair's source, 1: its target)
};

// scalar slots kept on the device during PCG
enum : int { S_RZ = 0, S_RZOLD = 1, S_BETA = 2, S_PQ = 3, S_ALPHA = 4, S_RR = 5, S_RZ0 = 6, S_COST = 7,
             S_DG = 8, S_DR = 9, S_DLD = 10, S_DD = 11, S_XX = 12, S_NVALID = 13, S_GMAX = 14,
             // PCG control, owned by the device: S_DONE 0 running / 1 converged / 2 NaN, S_TARGET = tol^2 rz0,
             // S_ITERS = iterations applied.  Every kernel of an iteration returns at once when S_DONE is set, so
             // the host may enqueue iterations ahead of the convergence test without changing the result.
             S_DONE = 15, S_TARGET = 16, S_ITERS = 17,
             S_RZPART = 18,  // block-Jacobi part of r^T z while the coarse level (cvd_coarse.h) is pending
             S_NACTIVE = 19, // number of active unknowns (k_step_stats)
             S_COUNT = 20 };

// Sum of a short global array (F per-frame partials, L2-resident) by every workgroup that needs the scalar:
// cheaper than a separate 1-block reduction kernel + its launch boundary. `red` = 4 doubles of LDS.
__device__ __forceinline__ double blockSumArray(const double* __restrict__ a, int n, double* red) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += a[i];
  acc = waveSum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (blockDim.x >> 6); ++w) t += red[w];
  __syncthreads();
  return t;
}

// "Last workgroup finishes the reduction": every workgroup publishes its partials with an agent-scope release,
// takes a ticket, and the last one acquires and reduces (cdna_hip_programming.md G16: release on the producer,
// acquire on the consumer; L1 is per CU and the 8 XCD L2s are not coherent with each other).
// `flag` = one int of LDS. Returns true (for every thread of the block) in the last-arriving workgroup.
__device__ __forceinline__ bool lastBlockArrives(unsigned int* counter, unsigned int total, int* flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == total - 1);
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
    }
    *flag = last ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}

// The same hand-off WITHOUT fences, for the three kernels of a PCG iteration (their workgroups exchange a handful of
// doubles): the partials are published by write-through agent-scope stores (publishPartial), every storing wave drains
// them (s_waitcnt vmcnt(0)) before the workgroup's ticket, and the last workgroup reads them back with agent-scope loads
// (readPartial) -- cdna_hip_programming.md G16 recipe R1.  The release fence of lastBlockArrives is a buffer_wbl2 of the
// XCD's whole L2 in EVERY workgroup and the acquire an invalidate, although nothing but the partials crosses workgroups.
// The protocol leans on the gfx942 / gfx950 memory system (agent-scope relaxed atomics are write-through / L2-coherent accesses,
// s_waitcnt vmcnt(0) drains them, workgroups of one launch are ordered by the ticket's data dependency) rather than on the HIP
// memory model's release / acquire edges: this library is built for gfx950 only, and says so.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the fence-free inter-workgroup hand-off (publishPartial / readPartial / lastBlockArrivesLite) is written for gfx942 / gfx950"
#endif
__device__ __forceinline__ void publishPartial(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(__double_as_longlong(v)),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double readPartial(const double* p) {
  return __longlong_as_double(static_cast<long long>(__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
}
__device__ __forceinline__ bool lastBlockArrivesLite(unsigned int* counter, unsigned int total, int* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its published partials have landed
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = (t == total - 1);
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm for the next launch
    *flag = last ? 1 : 0;
  }
  __syncthreads();
  return *flag != 0;
}
// sum of n published partials by the whole (last) workgroup; `red` = 16 doubles of LDS
__device__ __forceinline__ double blockSumPartials(const double* __restrict__ a, int n, double* red) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += readPartial(a + i);
  acc = waveSum(acc);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (blockDim.x >> 6); ++w) t += red[w];
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------------------------------------------
inline __global__ void k_frame_consts(Layout L, const double* __restrict__ x, FrameConst* __restrict__ fc) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= L.F) return;
  FrameConst c;
  frameConstFromParams(x + static_cast<size_t>(f) * L.B, L.intrOpt, L.vFocal, x, c);
  fc[f] = c;
}

// Observation ctor, reference lib/PoseOptimizer.cpp:104-116 (float arithmetic, no FMA contraction).
inline __global__ void k_build_table(int W, int H, float invAspect, long long C, const float4* __restrict__ loc,
                              const unsigned char* __restrict__ isStatic, const int* __restrict__ cpair,
                              const int* __restrict__ pairA, const int* __restrict__ pairB,
                              const unsigned char* __restrict__ inRange, const float* __restrict__ depth,
                              float4* __restrict__ ndc, float2* __restrict__ dsrc,
                              unsigned long long* __restrict__ nValid, int ignoreStatic) {
  // ignoreStatic: normalizeDepth's pair loop takes every constraint, dynamic ones included (reference
  // lib/PoseOptimizer.cpp:1036-1052 never looks at isStatic)
  const long long c = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  bool ok = false;
  if (c < C) {
    const float4 l = loc[c];
    const int p = cpair[c];
    const int fa = pairA[p], fb = pairB[p];
    float4 n;
    n.x = __fadd_rn(-1.f, __fmul_rn(2.f, l.x));
    n.y = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, l.y), invAspect));
    n.z = __fadd_rn(-1.f, __fmul_rn(2.f, l.z));
    n.w = __fsub_rn(1.f, __fdiv_rn(__fmul_rn(2.f, l.w), invAspect));
    int ax = static_cast<int>(__fmul_rn(l.x, static_cast<float>(W)));
    int ay = static_cast<int>(__fmul_rn(__fdiv_rn(l.y, invAspect), static_cast<float>(H)));
    int bx = static_cast<int>(__fmul_rn(l.z, static_cast<float>(W)));
    int by = static_cast<int>(__fmul_rn(__fdiv_rn(l.w, invAspect), static_cast<float>(H)));
    ax = min(max(ax, 0), W - 1); ay = min(max(ay, 0), H - 1);
    bx = min(max(bx, 0), W - 1); by = min(max(by, 0), H - 1);
    const size_t fs = static_cast<size_t>(W) * H;
    float da = depth[fa * fs + static_cast<size_t>(ay) * W + ax];
    float db = depth[fb * fs + static_cast<size_t>(by) * W + bx];
    ok = (ignoreStatic || isStatic[c]) && inRange[fa] && inRange[fb] && isfinite(da) && da > 0.f && isfinite(db) && db > 0.f;
    if (!ok) { da = 0.f; db = 0.f; }
    ndc[c] = n;
    dsrc[c] = make_float2(da, db);
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(nValid, static_cast<unsigned long long>(__popcll(b)));
}

// Table order (cvd_solver_options::constraint_order).  The constraint lists arrive in raster order: the 64 lanes of a
// wave are neighbours along an image row, 2 - 4 of them inside the same cell of the depth grid, and their LDS f64 atomics
// on that cell's vertices (and the lanes of the next sample row: the same cells again) serialise -- SQ_LDS_BANK_CONFLICT is
// 88 % of the LDS-active cycles of the hot product.  Every directed pair's slice of the table is therefore re-ordered as a
// SWEEP OVER THE CELLS: first one constraint of every non-empty cell in cell order, then the second of every cell that has
// one, ...  Consecutive lanes then hit consecutive cells, i.e. distinct vertices on consecutive LDS banks (measured on the
// benchmark: product 52.6 -> 49.5 us, assembly 0.38 -> 0.34 ms; 1000 frames / 16 x 12 grid: 242 -> 212 us, 1.71 -> 1.20 ms;
// a random order: 56 us; two constraints of a cell side by side: no gain).  One wave per directed pair, windows of
// kOrderCap constraints, everything in a fixed order (the sums downstream stay reproducible).
constexpr int kOrderCap = 4096;       // constraints per window
constexpr int kOrderMaxCells = 4096;  // gx * gy
inline __global__ __launch_bounds__(64) void k_order_table(const long long* __restrict__ pairOff, int gx, int gy, double maxcx,
                                                           double maxcy, const float4* __restrict__ ndcIn,
                                                           const float2* __restrict__ dsrcIn, float4* __restrict__ ndcOut,
                                                           float2* __restrict__ dsrcOut) {
  extern __shared__ __attribute__((aligned(16))) int smo[];
  const int nCells = gx * gy;
  int* start = smo;  // nCells + 1: counts, then exclusive prefix sums
  unsigned short* cellOf = reinterpret_cast<unsigned short*>(smo + nCells + 1);
  unsigned short* rankOf = cellOf + kOrderCap;
  unsigned short* sorted = rankOf + kOrderCap;
  const int lane = threadIdx.x;
  const unsigned long long below = (1ull << lane) - 1ull;
  const long long pb = pairOff[blockIdx.x], pe = pairOff[blockIdx.x + 1];
  for (long long w0 = pb; w0 < pe; w0 += kOrderCap) {
    const int n = static_cast<int>(pe - w0 < kOrderCap ? pe - w0 : kOrderCap);
    for (int c = lane; c <= nCells; c += 64) start[c] = 0;
    __syncthreads();
    // A: cell of every constraint and its rank among the constraints of that cell (input order)
    for (int i0 = 0; i0 < n; i0 += 64) {
      const int i = i0 + lane;
      const bool valid = i < n;
      int cell = -1;
      if (valid) {
        const float4 nd = ndcIn[w0 + i];
        int ix, iy;
        double rx, ry;
        gridCell(nd.x, gx, maxcx, ix, rx);
        gridCell(nd.y, gy, maxcy, iy, ry);
        cell = ix + iy * gx;
      }
      unsigned long long todo = __ballot(valid);
      int rank = 0;
      while (todo) {
        const int leader = __ffsll(static_cast<long long>(todo)) - 1;
        const int c0 = __shfl(cell, leader, 64);
        const bool mine = valid && cell == c0;
        const unsigned long long m = __ballot(mine);
        if (mine) rank = start[c0] + __popcll(m & below);
        if (lane == leader) start[c0] += __popcll(m);   // (LDS operations of one wave execute in order)
        todo &= ~m;
      }
      if (valid) {
        cellOf[i] = static_cast<unsigned short>(cell);
        rankOf[i] = static_cast<unsigned short>(rank);
      }
    }
    __syncthreads();
    // exclusive prefix sums of the counts
    int carry = 0;
    for (int c0 = 0; c0 <= nCells; c0 += 64) {
      const int c = c0 + lane;
      const int v = c < nCells ? start[c] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d, 64);
        if (lane >= d) incl += t;
      }
      if (c <= nCells) start[c] = carry + incl - v;
      carry += __shfl(incl, 63, 64);
    }
    __syncthreads();
    // B: constraints grouped by cell
    for (int i = lane; i < n; i += 64) sorted[start[cellOf[i]] + rankOf[i]] = static_cast<unsigned short>(i);
    __syncthreads();
    // C: sweep r = 0, 1, ...: the r-th constraint of every cell that has one, in cell order
    int base = 0;
    for (int r = 0; base < n; ++r) {
      for (int c0 = 0; c0 < nCells; c0 += 64) {
        const int c = c0 + lane;
        const bool has = c < nCells && start[c + 1] - start[c] > r;
        const unsigned long long m = __ballot(has);
        if (has) {
          const long long src = w0 + sorted[start[c] + r];
          const long long dst = w0 + base + __popcll(m & below);
          ndcOut[dst] = ndcIn[src];
          dsrcOut[dst] = dsrcIn[src];
        }
        base += __popcll(m);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------
// candidate-point cost: pair-major over work items
// ---------------------------------------------------------------------------------------------------
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_cost_items(Layout L, Table T, Items it, const double* __restrict__ x,
                                                    const FrameConst* __restrict__ fc, double* __restrict__ costItem) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* xa = sm;
  double* xb = sm + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(sm + 2 * B);
  double* red = reinterpret_cast<double*>(fcs + 2);
  const int item = blockIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  if (threadIdx.x < 2 * (sizeof(FrameConst) / 8)) {
    const int which = threadIdx.x / (sizeof(FrameConst) / 8);
    const int k = threadIdx.x % (sizeof(FrameConst) / 8);
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  __syncthreads();
  double acc = 0.0;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fs = fcs[dir];
    const FrameConst& Ft = fcs[dir ^ 1];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    for (long long c = cb + threadIdx.x; c < ce; c += blockDim.x) {
      const float2 d = T.dsrc[c];
      if (d.x > 0.f) {
        Sample<KD, KS> s;
        evalSample<KD, KS, false>(L, Fs, Ft, xs, xt, T.ndc[c], d, s);
        acc += s.rho0;
      }
    }
  }
  acc = waveSum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < (blockDim.x >> 6); ++w) t += red[w];
    costItem[item] = 0.5 * t;
  }
}

template <int KD>
inline __global__ __launch_bounds__(256) void k_cost_frames(Layout L, const double* __restrict__ x,
                                                     const float* __restrict__ median,
                                                     const unsigned char* __restrict__ inRange,
                                                     const unsigned char* __restrict__ rangeFlags,
                                                     double* __restrict__ costFrame) {
  // (the frame's parameters go through LDS: every residual would otherwise start with its own dependent global load)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ double red[4];
  const int f = blockIdx.x;
  double acc = 0.0;
  if (threadIdx.x == 0 && L.positionRegSqrt > 0.0) {
    double o3[3] = {0, 0, 0}, dg = 0.0, cst = 0.0;
    if (posRegValid(L, rangeFlags, f)) posRegFrame(L, rangeFlags, f, x, nullptr, o3, dg, cst);
    acc += cst;
  }
  const bool active = inRange[f] != 0;
  if (active)
    for (int i = threadIdx.x; i < L.B; i += 256) sm[i] = x[static_cast<size_t>(f) * L.B + i];
  __syncthreads();
  if (active) {
    const float med = median[f];
    const int nr = numRegResiduals<KD>(L);
    for (int i = threadIdx.x; i < nr; i += 256) {
      double r;
      int n;
      int cols[2 * KD + 2];
      double jac[2 * KD + 2];
      regResidual<KD>(L, f, i, sm, med, r, n, cols, jac);
      acc += r * r;
    }
  }
  acc = waveSum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) costFrame[f] = 0.5 * ((red[0] + red[1]) + (red[2] + red[3]));
}

// deterministic final sum: out[slot] = sum(a[0..na)) + sum(b[0..nb))
inline __global__ __launch_bounds__(256) void k_sum2(const double* __restrict__ a, int na, const double* __restrict__ b,
                                              int nb, double* __restrict__ out, int slot) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < na; i += 256) acc += a[i];
  for (int i = threadIdx.x; i < nb; i += 256) acc += b[i];
  acc = waveSum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[slot] = red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// Frame-major assembly: one workgroup owns frame f, walks every pair it takes part in (as source or as
// target), and accumulates g_f = J_f^T r and H_ff = J_f^T J_f in LDS (packed lower triangle), then adds the
// frame's regularisers.  No global atomics, no partial buffers, output written exactly once.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int packedIdx(int i, int j) {  // i >= j
  return i * (i + 1) / 2 + j;
}

// Row panels of the packed lower triangle for k_assemble / k_assemble_triplets: the triangle of a frame block
// (B (B + 1) / 2 doubles: 162 KB at B = 201, the reference's default deferred-spatial layout 7 + 170 + 24) no longer
// fits the 160 KB of LDS beyond B = 199, so the kernels accumulate it panel by panel -- rows [row[k], row[k + 1]) per
// pass over the frame's constraints (the Jacobians are re-evaluated per pass; one pass whenever the triangle fits).
struct AsmPanels {
  int n;
  int row[9];  // n + 1 boundaries, row[0] = 0, row[n] = B
};

template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_assemble(Layout L, Table T, const double* __restrict__ x,
                                                  const FrameConst* __restrict__ fc, const double* __restrict__ mask,
                                                  const float* __restrict__ median,
                                                  const unsigned char* __restrict__ inRange,
                                                  const unsigned char* __restrict__ rangeFlags,
                                                  const int* __restrict__ fpOff, const int* __restrict__ fpList,
                                                  double* __restrict__ gOut, double* __restrict__ hOut,
                                                  double* __restrict__ costFrame, double* __restrict__ focalG,
                                                  double* __restrict__ focalH, AsmPanels panels, int panelCap) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* Hs = sm;             // one row panel of the packed lower triangle (panelCap doubles)
  double* gs = Hs + panelCap;  // B
  double* xf = gs + B;         // B
  double* xo = xf + B;         // B
  FrameConst* fcs = reinterpret_cast<FrameConst*>(xo + B);  // [0] = own frame, [1] = other frame
  double* red = reinterpret_cast<double*>(fcs + 2);          // 4 * 36
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  for (int i = tid; i < B; i += 256) {
    gs[i] = 0.0;
    xf[i] = x[static_cast<size_t>(f) * B + i];
  }
  constexpr int FCW = sizeof(FrameConst) / 8;
  if (tid < FCW) reinterpret_cast<double*>(fcs)[tid] = reinterpret_cast<const double*>(fc + f)[tid];
  const double* mf = mask + static_cast<size_t>(f) * B;
  double* hf = hOut + static_cast<size_t>(f) * B * B;
  double staticCost = 0.0, regCostTotal = 0.0;

  for (int pass = 0; pass < panels.n; ++pass) {
  const int r0 = panels.row[pass], r1 = panels.row[pass + 1];
  const int base = r0 * (r0 + 1) / 2, npk = r1 * (r1 + 1) / 2 - base;
  const bool first = pass == 0;  // gradient, cost and the shared-focal sums are taken in the first pass only
  // entry (hi, lo), hi >= lo, of the triangle: in this panel iff r0 <= hi < r1
#define CVD_PANEL_ADD(hi_, lo_, val_)                                                       \
  do {                                                                                      \
    const int hi__ = (hi_);                                                                 \
    if (hi__ >= r0 && hi__ < r1) atomicAdd(&Hs[packedIdx(hi__, (lo_)) - base], (val_));     \
  } while (0)
  __syncthreads();
  for (int i = tid; i < npk; i += 256) Hs[i] = 0.0;
  __syncthreads();

  // register accumulators of the pose-like 7x7 block + gradient (all lanes hit the same addresses)
  double PP[28];
  double gp[7];
  double cost = 0.0;
  double shG = 0.0, shH = 0.0;  // IntrinsicsOptimization::Shared: focal gradient / squared column norm
#pragma unroll
  for (int i = 0; i < 28; ++i) PP[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) gp[i] = 0.0;

  if (L.includeStatic) {
    for (int e = fpOff[f]; e < fpOff[f + 1]; ++e) {
      const int code = fpList[e];
      const int p = code >> 1;
      const int side = code & 1;  // 0: f is the source (a) of pair p, 1: f is the target (b)
      const int o = side ? T.pairA[p] : T.pairB[p];
      __syncthreads();
      for (int i = tid; i < B; i += 256) xo[i] = x[static_cast<size_t>(o) * B + i];
      if (tid < FCW) reinterpret_cast<double*>(fcs + 1)[tid] = reinterpret_cast<const double*>(fc + o)[tid];
      __syncthreads();
      const FrameConst& fa = side ? fcs[1] : fcs[0];
      const FrameConst& fb = side ? fcs[0] : fcs[1];
      const double* xa = side ? xo : xf;
      const double* xb = side ? xf : xo;
      for (long long c = T.pairOff[p] + tid; c < T.pairOff[p + 1]; c += 256) {
        const float2 d = T.dsrc[c];
        if (!(d.x > 0.f)) continue;
        Sample<KD, KS> s;
        evalSample<KD, KS, true>(L, fa, fb, xa, xb, T.ndc[c], d, s);
        const double w = s.rho1;
        if (L.intrOpt == kIntrShared) {
          // one focal length: the focal column of this constraint is (d r / d f_a + d r / d f_b); its gradient and
          // squared norm are taken once per constraint (source visit) for frame 0's slot
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) {
            const double tot = s.a.Jp[rr][6] + s.b.Jp[rr][6];
            s.a.Jp[rr][6] = tot;
            s.b.Jp[rr][6] = tot;
          }
          if (!side && first) {
            shG += w * (s.a.Jp[0][6] * s.r[0] + s.a.Jp[1][6] * s.r[1] + s.a.Jp[2][6] * s.r[2]);
            shH += w * (s.a.Jp[0][6] * s.a.Jp[0][6] + s.a.Jp[1][6] * s.a.Jp[1][6] + s.a.Jp[2][6] * s.a.Jp[2][6]);
          }
        }
        const Side<KD, KS>& me = side ? s.b : s.a;
        if (!side && first) cost += s.rho0;  // count every constraint once
        // pose-like block (rows 0..6: first panel)
        if (first) {
          int q = 0;
#pragma unroll
          for (int i = 0; i < 7; ++i) {
            gp[i] += w * (me.Jp[0][i] * s.r[0] + me.Jp[1][i] * s.r[1] + me.Jp[2][i] * s.r[2]);
#pragma unroll
            for (int j = 0; j <= i; ++j) {
              PP[q] += w * (me.Jp[0][i] * me.Jp[0][j] + me.Jp[1][i] * me.Jp[1][j] + me.Jp[2][i] * me.Jp[2][j]);
              ++q;
            }
          }
        }
        // tap columns
        const int nt = sideNumTapCols(L, me);
        for (int t = 0; t < nt; ++t) {
          int ct;
          double Jt[3];
          sideTapCol(L, me, t, ct, Jt);
          const double wj0 = w * Jt[0], wj1 = w * Jt[1], wj2 = w * Jt[2];
          if (first) atomicAdd(&gs[ct], wj0 * s.r[0] + wj1 * s.r[1] + wj2 * s.r[2]);
          if (ct >= r0 && ct < r1) {
            const int rowBase = ct * (ct + 1) / 2 - base;
#pragma unroll
            for (int i = 0; i < 7; ++i)
              atomicAdd(&Hs[rowBase + i], wj0 * me.Jp[0][i] + wj1 * me.Jp[1][i] + wj2 * me.Jp[2][i]);
          }
          for (int t2 = 0; t2 <= t; ++t2) {
            int c2;
            double J2[3];
            sideTapCol(L, me, t2, c2, J2);
            const double val = wj0 * J2[0] + wj1 * J2[1] + wj2 * J2[2];
            // tap columns of one sample are distinct but not sorted (border folding keeps row-major order,
            // spatial columns follow depth columns) -> order the pair
            const int hi = ct > c2 ? ct : c2, lo = ct > c2 ? c2 : ct;
            CVD_PANEL_ADD(hi, lo, val);
          }
        }
      }
    }
  }
  __syncthreads();
  if (first) {
    // block reduction of the register accumulators
    {
#pragma unroll
      for (int i = 0; i < 28; ++i) PP[i] = waveSum(PP[i]);
#pragma unroll
      for (int i = 0; i < 7; ++i) gp[i] = waveSum(gp[i]);
      cost = waveSum(cost);
      const int wv = tid >> 6;
      if ((tid & 63) == 0) {
#pragma unroll
        for (int i = 0; i < 28; ++i) red[wv * 36 + i] = PP[i];
#pragma unroll
        for (int i = 0; i < 7; ++i) red[wv * 36 + 28 + i] = gp[i];
        red[wv * 36 + 35] = cost;
      }
    }
    __syncthreads();
    if (tid < 28) {
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= tid) ++i;
      const int j = tid - i * (i + 1) / 2;
      Hs[packedIdx(i, j)] += red[tid] + red[36 + tid] + red[72 + tid] + red[108 + tid];  // (rows 0..6 are in panel 0: r1 >= 7)
    } else if (tid < 35) {
      gs[tid - 28] += red[tid] + red[36 + tid] + red[72 + tid] + red[108 + tid];
    }
    __syncthreads();
    staticCost = 0.5 * (red[35] + red[36 + 35] + red[72 + 35] + red[108 + 35]);
    __syncthreads();
  }
  if (L.intrOpt == kIntrShared) {
    // The focal column of every constraint belongs to frame 0's slot: publish this frame's static focal
    // gradient / diagonal for k_shared_focal_fixup and drop the entries from the frame's own block (for f != 0
    // they are off-diagonal couplings with frame 0, which the block-Jacobi preconditioner does not hold).
    if (first) {
      shG = waveSum(shG);
      shH = waveSum(shH);
      if ((tid & 63) == 0) { red[tid >> 6] = shG; red[4 + (tid >> 6)] = shH; }
      __syncthreads();
      if (tid == 0) {
        focalG[f] = red[0] + red[1] + red[2] + red[3];
        focalH[f] = red[4] + red[5] + red[6] + red[7];
        gs[6] = 0.0;
        Hs[packedIdx(6, 6)] = 0.0;
      }
    }
    if (f != 0) {
      for (int j = tid; j < B; j += 256) {
        if (j == 6) continue;
        const int hi = j > 6 ? j : 6, lo = j > 6 ? 6 : j;
        if (hi >= r0 && hi < r1) Hs[packedIdx(hi, lo) - base] = 0.0;
      }
    }
    __syncthreads();
  }

  // regularisers of this frame
  double regCost = 0.0;
  if (inRange[f]) {
    const int nr = numRegResiduals<KD>(L);
    for (int i = tid; i < nr; i += 256) {
      double r;
      int n;
      int cols[2 * KD + 2];
      double jac[2 * KD + 2];
      regResidual<KD>(L, f, i, xf, median[f], r, n, cols, jac);
      if (first) regCost += r * r;
      for (int a = 0; a < n; ++a) {
        if (first) atomicAdd(&gs[cols[a]], jac[a] * r);
        for (int b = 0; b <= a; ++b) {
          const int hi = cols[a] > cols[b] ? cols[a] : cols[b];
          const int lo = cols[a] > cols[b] ? cols[b] : cols[a];
          CVD_PANEL_ADD(hi, lo, jac[a] * jac[b]);
        }
      }
    }
  }
  if (tid == 0 && L.positionRegSqrt > 0.0 && first) {
    double o3[3] = {0, 0, 0}, dg = 0.0, cst = 0.0;
    posRegFrame(L, rangeFlags, f, x, nullptr, o3, dg, cst);
    for (int i = 0; i < 3; ++i) {
      atomicAdd(&gs[i], o3[i]);
      atomicAdd(&Hs[packedIdx(i, i)], dg);
    }
    regCost += cst;
  }
  if (first) {
    regCost = waveSum(regCost);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = regCost;
    __syncthreads();
    regCostTotal = 0.5 * (red[0] + red[1] + red[2] + red[3]);
  }
  __syncthreads();
  // write-out of this panel with the constant-parameter mask applied (constant columns drop out of J): entry (i, j),
  // j <= i, goes to both triangles of the full block
  for (int idx = tid; idx < npk; idx += 256) {
    int i = static_cast<int>((sqrt(8.0 * static_cast<double>(idx + base) + 1.0) - 1.0) * 0.5);
    while (i * (i + 1) / 2 > idx + base) --i;
    while ((i + 1) * (i + 2) / 2 <= idx + base) ++i;
    const int j = idx + base - i * (i + 1) / 2;
    const double v = Hs[idx] * mf[i] * mf[j];
    hf[static_cast<size_t>(i) * B + j] = v;
    hf[static_cast<size_t>(j) * B + i] = v;
  }
#undef CVD_PANEL_ADD
  }  // pass
  if (tid == 0) costFrame[f] = staticCost + regCostTotal;
  __syncthreads();
  for (int i = tid; i < B; i += 256) gOut[static_cast<size_t>(f) * B + i] = gs[i] * mf[i];
}

// IntrinsicsOptimization::Shared: frame 0's focal slot receives the static focal gradient / diagonal of all frames.
inline __global__ __launch_bounds__(256) void k_shared_focal_fixup(Layout L, const double* __restrict__ focalG,
                                                            const double* __restrict__ focalH,
                                                            const double* __restrict__ mask, double* __restrict__ g,
                                                            double* __restrict__ hBlocks) {
  __shared__ double red[8];
  double a = 0.0, b = 0.0;
  for (int f = threadIdx.x; f < L.F; f += 256) { a += focalG[f]; b += focalH[f]; }
  a = waveSum(a);
  b = waveSum(b);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = a; red[4 + (threadIdx.x >> 6)] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double m = mask[6];
    g[6] += (red[0] + red[1] + red[2] + red[3]) * m;
    hBlocks[static_cast<size_t>(6) * L.B + 6] += (red[4] + red[5] + red[6] + red[7]) * m * m;
  }
}

// ---------------------------------------------------------------------------------------------------
// LM diagonal (Ceres LevenbergMarquardtStrategy::ComputeStep in the unscaled variables):
//   scale_j = 1 / (1 + sqrt(h_jj)) from the FIRST Jacobian; lam_j = clamp(scale_j^2 h_jj) / (radius scale_j^2).
// Unknowns nothing depends on (h_jj == 0: constant or absent parameter blocks) get lam = 1, so that the
// damped system is the identity on them and their step is exactly 0.
// ---------------------------------------------------------------------------------------------------
// hdiag = diag(H_ff) of every frame as a flat F x B vector (k_extract_diag; all-gathered from the frames' owners in the
// pair-sharded multi-GPU mode, where a rank holds the reduced H_ff of its own frames only).
inline __global__ void k_lm_diag(Layout L, const double* __restrict__ hdiag, double* __restrict__ scale,
                          int computeScale, double radius, double* __restrict__ lam) {
  const size_t n = static_cast<size_t>(L.F) * L.B;
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double h = hdiag[i];
  if (computeScale) scale[i] = 1.0 / (1.0 + sqrt(h));
  const double s = scale[i];
  if (h == 0.0) {
    lam[i] = 1.0;
  } else {
    const double d = fmin(fmax(s * s * h, 1e-6), 1e32);
    lam[i] = d / (radius * s * s);
  }
}

// Block-Jacobi preconditioner, register-resident variant: M_f^-1 = (H_ff + diag(lam_f))^-1 by the symmetric sweep
// operator (Gauss-Jordan without pivoting, valid for SPD blocks).  Sweeping pivot k maps
//   G_kk <- -1/G_kk,  G_ik <- G_ik / G_kk,  G_ij <- G_ij - G_ik G_kj / G_kk   (i, j != k)
// and after all B pivots G = -A^-1.  The lower triangle is cut into 4x4 tiles held in REGISTERS (tile id =
// tid + t * blockDim, TPT tiles per thread); a step only needs the pivot column, which its owners publish to a
// double-buffered LDS vector, so one barrier per pivot and ~16 FMAs + 64 B of LDS reads per tile and step.
// Padding rows/columns (B not a multiple of 4) are identity and never swept.
template <int TPT, int TS = 4>
inline __global__ __launch_bounds__(TS == 4 ? 1024 : 512, 4) void k_block_inverse_sweep(Layout L, const double* __restrict__ hBlocks,
                                                              const double* __restrict__ lam,
                                                              float* __restrict__ minv, int* __restrict__ fail) {
  __shared__ __attribute__((aligned(16))) double colBuf[2][264];
  static_assert(TS % 2 == 0, "the pivot column is read as double2");
  const int B = L.B;
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const int nT = blockDim.x;
  const int nb = (B + TS - 1) / TS;
  const int nTiles = nb * (nb + 1) / 2;
  const double* hf = hBlocks + static_cast<size_t>(f) * B * B;
  const double* lf = lam + static_cast<size_t>(f) * B;
  double T[TPT][TS][TS];
  int tI[TPT], tJ[TPT];
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    const int id = tid + t * nT;
    tI[t] = -1;
    tJ[t] = -1;
    if (id < nTiles) {
      int I = static_cast<int>((sqrtf(8.f * static_cast<float>(id) + 1.f) - 1.f) * 0.5f);
      while ((I + 1) * (I + 2) / 2 <= id) ++I;
      while (I * (I + 1) / 2 > id) --I;
      tI[t] = I;
      tJ[t] = id - I * (I + 1) / 2;
    }
#pragma unroll
    for (int p = 0; p < TS; ++p)
#pragma unroll
      for (int q = 0; q < TS; ++q) {
        const int i = TS * tI[t] + p, j = TS * tJ[t] + q;
        double v = (i == j) ? 1.0 : 0.0;
        if (tI[t] >= 0 && i < B && j < B) v = hf[static_cast<size_t>(i) * B + j] + (i == j ? lf[i] : 0.0);
        T[t][p][q] = v;
      }
    if (tJ[t] == 0) {  // publish pivot column 0
#pragma unroll
      for (int p = 0; p < TS; ++p) colBuf[0][TS * tI[t] + p] = T[t][p][0];
    }
  }
  __syncthreads();
  for (int kt = 0; kt < nb; ++kt) {
#pragma unroll
    for (int a = 0; a < TS; ++a) {
      const int k = TS * kt + a;
      if (k >= B) break;  // uniform
      const double* col = colBuf[k & 1];
      double* nxt = colBuf[(k + 1) & 1];
      double d = col[k];
      if (!(d > 0.0)) {
        if (tid == 0) atomicAdd(fail, 1);
        d = 1.0;
      }
      const double id = 1.0 / d;
      const int an = (a + 1) % TS;  // compile-time after unrolling
      const int ktn = kt + (a == TS - 1 ? 1 : 0);
#pragma unroll
      for (int t = 0; t < TPT; ++t) {
        if (tI[t] < 0) continue;
        double ci[TS], cj[TS];
#pragma unroll
        for (int e = 0; e < TS; e += 2) {
          const double2 vi = *reinterpret_cast<const double2*>(col + TS * tI[t] + e);
          const double2 vj = *reinterpret_cast<const double2*>(col + TS * tJ[t] + e);
          ci[e] = vi.x * id;  // c_i / d
          ci[e + 1] = vi.y * id;
          cj[e] = vj.x;
          cj[e + 1] = vj.y;
        }
#pragma unroll
        for (int p = 0; p < TS; ++p)
#pragma unroll
          for (int q = 0; q < TS; ++q) T[t][p][q] -= ci[p] * cj[q];
        if (tI[t] == kt) {  // row k of the tile: G_kj <- c_j / d
#pragma unroll
          for (int q = 0; q < TS; ++q) T[t][a][q] = cj[q] * id;
        }
        if (tJ[t] == kt) {  // column k of the tile: G_ik <- c_i / d
#pragma unroll
          for (int p = 0; p < TS; ++p) T[t][p][a] = ci[p];
          if (tI[t] == kt) T[t][a][a] = -id;
        }
        // publish the next pivot column (row k+1 of the tiles left of / on the diagonal, column k+1 below it)
        if (k + 1 < B) {
          if (tI[t] == ktn) {
#pragma unroll
            for (int q = 0; q < TS; ++q) nxt[TS * tJ[t] + q] = T[t][an][q];
          } else if (tJ[t] == ktn) {
#pragma unroll
            for (int p = 0; p < TS; ++p) nxt[TS * tI[t] + p] = T[t][p][an];
          }
        }
      }
      __syncthreads();
    }
  }
  float* Mf = minv + static_cast<size_t>(f) * B * B;
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    if (tI[t] < 0) continue;
#pragma unroll
    for (int p = 0; p < TS; ++p)
#pragma unroll
      for (int q = 0; q < TS; ++q) {
        const int i = TS * tI[t] + p, j = TS * tJ[t] + q;
        if (i < B && j < B) {
          const float v = static_cast<float>(-T[t][p][q]);
          Mf[static_cast<size_t>(i) * B + j] = v;
          if (tI[t] != tJ[t]) Mf[static_cast<size_t>(j) * B + i] = v;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------
// Block-Jacobi preconditioner on the matrix cores: the same symmetric sweep, BLOCKED with 16-wide pivot blocks so
// that the work is dense 16x16x16 products on v_mfma_f64_16x16x4_f64 and the dependent chain is nb = ceil(B / 16)
// block steps instead of B scalar pivots.  Sweeping pivot block k (P = G_kk^-1) maps
//   G_kk <- -P,   G_ik <- G_ik P,   G_kj <- P G_kj,   G_ij <- G_ij - G_ik P G_kj        (i, j != k)
// (= the composition of the block's 16 scalar sweeps) and after all nb blocks G = -A^-1.
// Layout: the lower-triangle 16x16 tiles (i >= j) live in MFMA accumulator registers for the whole kernel, tile
// id = wave + s * NW (TPW tiles per wave; accumulator layout of the f64 MFMA: col = lane & 15, row = (lane >> 4) + 4 r).
// Per block step:
//   A  the owners of the tiles of block row / column k publish the panel A(:, k) to LDS (row tiles transposed);
//      the owner wave of the pivot tile inverts it in-wave (scalar symmetric sweep on a 4-elements-per-lane layout,
//      pivot row / column exchanged by lane shuffles) and publishes G_kk = -P;
//   B  every wave forms its share of -T_i = A(i, k) (-P) (4 MFMAs per tile) into the LDS T panel;
//   C  every wave updates its tiles: G_ij += (-T_i) A(j, k)^T (4 MFMAs per tile, operands from the two LDS panels);
//      tiles of block row / column k are replaced by T_i resp. T_j^T, the pivot tile by -P.
// Three barriers per block step, 3 nb in total (B = 177: 36 instead of 177).  Padding (B not a multiple of 16) is
// identity.  f64 in, f64 arithmetic, f32 out (what k_cg_update consumes), like the scalar kernels.
typedef double cvd_d4 __attribute__((ext_vector_type(4)));
#ifdef CVD_INV_PROFILE  // tools/inv_bench.hip: shader-clock cycles per phase and wave of workgroup 0
__device__ unsigned long long g_invProf[16 * 8];
#define CVD_INV_T(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[slot] += t_ - tLast; tLast = t_; } while (0)
#else
#define CVD_INV_T(slot) do { } while (0)
#endif
__device__ __forceinline__ double readlaneF64(double v, int srcLane) {  // srcLane uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srcLane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srcLane);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ double rowBroadcastF64(double v) {  // lane SRC of every 16-lane row to the whole row (DPP row_newbcast)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + SRC, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + SRC, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// One scalar sweep of the 16x16 pivot tile in the (row, column group) lane layout of k_block_inverse_mfma.
template <int P>
__device__ __forceinline__ void invPivotStep(double (&g)[4], int row, int cg, int& bad) {
  const double gi = __shfl(g[P & 3], row + 16 * (P >> 2), 64);
  double d = readlaneF64(g[P & 3], P + 16 * (P >> 2));
  double cj[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) cj[e] = rowBroadcastF64<P>(g[e]);
  if (!(d > 0.0)) {  // uniform
    bad = 1;
    d = 1.0;
  }
  // 1 / d by v_rcp_f64 + two Newton steps (~1 ulp; the result is stored as f32): the IEEE division sequence is twice as
  // long and sits on the dependent chain of all 16 pivots
  double id = __builtin_amdgcn_rcp(d);
  id = fma(fma(-d, id, 1.0), id, id);
  id = fma(fma(-d, id, 1.0), id, id);
  const double ci = gi * id;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    double v = g[e] - ci * cj[e];
    if (row == P) v = cj[e] * id;
    if (4 * cg + e == P) v = (row == P) ? -id : ci;
    g[e] = v;
  }
}
constexpr int kInvTS = 16;            // tile size = MFMA M = N
constexpr int kInvLd = 17;            // LDS row stride of a tile (doubles): conflict-free operand reads
constexpr int kInvTile = kInvTS * kInvLd;

template <int NW, int TPW>
inline __global__ __launch_bounds__(NW * 64, 4) void k_block_inverse_mfma(Layout L, const double* __restrict__ hBlocks,
                                                               const double* __restrict__ lam, float* __restrict__ minv,
                                                               int* __restrict__ fail) {
  extern __shared__ __attribute__((aligned(16))) double invSmem[];
  const int B = L.B;
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nb = (B + kInvTS - 1) / kInvTS;
  const int nTiles = nb * (nb + 1) / 2;
  double* panel = invSmem;                     // [nb][16][17]  A(m, k) of the current block step
  double* tneg = invSmem + nb * kInvTile;      // [nb][16][17]  -T_m = A(m, k) (-P)
  double* piv = tneg + nb * kInvTile;          // [16][17]      G_kk = -P
  const int c = lane & 15, r0 = lane >> 4;
  const double* hf = hBlocks + static_cast<size_t>(f) * B * B;
  const double* lf = lam + static_cast<size_t>(f) * B;

#ifdef CVD_INV_PROFILE
  unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tLast = __builtin_amdgcn_s_memtime();
#endif
  cvd_d4 acc[TPW];
  int tI[TPW], tJ[TPW];
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    const int id = w + s * NW;
    int I = -1, J = -1;
    if (id < nTiles) {
      I = static_cast<int>((sqrtf(8.f * static_cast<float>(id) + 1.f) - 1.f) * 0.5f);
      while ((I + 1) * (I + 2) / 2 <= id) ++I;
      while (I * (I + 1) / 2 > id) --I;
      J = id - I * (I + 1) / 2;
    }
    tI[s] = __builtin_amdgcn_readfirstlane(I);
    tJ[s] = __builtin_amdgcn_readfirstlane(J);
    // unconditional loads from clamped addresses: all of a wave's loads are in flight before the first one is used
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kInvTS * (I < 0 ? 0 : I) + r0 + 4 * r, j = kInvTS * (J < 0 ? 0 : J) + c;
      acc[s][r] = hf[static_cast<size_t>(min(i, B - 1)) * B + min(j, B - 1)];
    }
  }
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double v = acc[s][r];
      asm volatile("" : "+v"(v));  // (keeps the compiler from sinking the load into the bounds test below)
      const int i = kInvTS * tI[s] + r0 + 4 * r, j = kInvTS * tJ[s] + c;
      const bool in = tI[s] >= 0 && i < B && j < B;
      if (tI[s] == tJ[s] && i == j && in) v += lf[i];   // damping on the diagonal (diagonal tiles only: wave-uniform)
      acc[s][r] = in ? v : (i == j ? 1.0 : 0.0);
    }
  }
  CVD_INV_T(0);
  for (int k = 0; k < nb; ++k) {
    // ---- A: publish the panel of block column k; the pivot tile's owner inverts it
    bool ownsPivot = false;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      if (tI[s] < 0) continue;
      if (tJ[s] == k && tI[s] > k) {          // A(i, k) as stored
        int o = tI[s] * kInvTile;
        asm volatile("" : "+s"(o));
        double* dst = panel + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
      } else if (tI[s] == k && tJ[s] < k) {   // A(j, k) = A(k, j)^T
        int o = tJ[s] * kInvTile;
        asm volatile("" : "+s"(o));
        double* dst = panel + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[c * kInvLd + r0 + 4 * r] = acc[s][r];
      } else if (tI[s] == k && tJ[s] == k) {  // pivot tile
#pragma unroll
        for (int r = 0; r < 4; ++r) piv[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
        ownsPivot = true;
      }
    }
    CVD_INV_T(1);
    if (ownsPivot) {  // wave-uniform: in-wave inverse of the pivot tile, G_kk <- -P
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // lane = (row = lane & 15, column group cg = lane >> 4): 4 elements G[row][4 cg + e] per lane.  Pivot step p needs
      //   G[p][4 cg + e]  from lane p of the SAME 16-lane row group  -> DPP row broadcast (VALU move, no LDS round trip)
      //   G[p][p]         from one known lane                         -> v_readlane
      //   G[row][p]       from lane row + 16 (p >> 2)                 -> the one ds_bpermute of the step
      const int row = lane & 15, cg = lane >> 4;
      double g[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = piv[row * kInvLd + 4 * cg + e];
      int bad = 0;
      invPivotStep<0>(g, row, cg, bad);   invPivotStep<1>(g, row, cg, bad);   invPivotStep<2>(g, row, cg, bad);
      invPivotStep<3>(g, row, cg, bad);   invPivotStep<4>(g, row, cg, bad);   invPivotStep<5>(g, row, cg, bad);
      invPivotStep<6>(g, row, cg, bad);   invPivotStep<7>(g, row, cg, bad);   invPivotStep<8>(g, row, cg, bad);
      invPivotStep<9>(g, row, cg, bad);   invPivotStep<10>(g, row, cg, bad);  invPivotStep<11>(g, row, cg, bad);
      invPivotStep<12>(g, row, cg, bad);  invPivotStep<13>(g, row, cg, bad);  invPivotStep<14>(g, row, cg, bad);
      invPivotStep<15>(g, row, cg, bad);
      if (bad && lane == 0) atomicAdd(fail, 1);
#pragma unroll
      for (int e = 0; e < 4; ++e) piv[row * kInvLd + 4 * cg + e] = g[e];
      CVD_INV_T(2);
    }
    __syncthreads();
    CVD_INV_T(3);
    // ---- B: -T_m = A(m, k) (-P) for every m != k
    for (int m = w; m < nb; m += NW) {
      if (m == k) continue;
      cvd_d4 t = {0.0, 0.0, 0.0, 0.0};
      const double* src = panel + m * kInvTile;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(src[c * kInvLd + 4 * kk + r0], piv[(4 * kk + r0) * kInvLd + c], t, 0, 0, 0);
      double* dst = tneg + m * kInvTile;
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(r0 + 4 * r) * kInvLd + c] = t[r];
    }
    CVD_INV_T(4);
    __syncthreads();
    CVD_INV_T(5);
    // ---- C: rank-16 update G_ij += (-T_i) A(j, k)^T of EVERY owned tile, branch-free so that the operand loads of the
    // next tile overlap the MFMAs of this one (tiles of block row / column k read stale panel slots: their result is
    // discarded by the fix-up below; unused slots update tile (0, 0) into a register nobody stores)
    const int laneOp = c * kInvLd + r0;  // operand element [row = c][k = 4 kk + r0] of a panel tile
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      int oa = (tI[s] < 0 ? 0 : tI[s]) * kInvTile, ob = (tJ[s] < 0 ? 0 : tJ[s]) * kInvTile;
      asm volatile("" : "+s"(oa), "+s"(ob));  // keep the per-tile LDS addresses out of loop-invariant VGPRs (they spill)
      const double* ta = tneg + oa + laneOp;
      const double* pb = panel + ob + laneOp;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[4 * kk], pb[4 * kk], acc[s], 0, 0, 0);
    }
    // fix-up: block row / column k and the pivot tile are replaced (nb of the nTiles tiles per step)
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
      if (tI[s] != k && tJ[s] != k) continue;
      if (tI[s] == k && tJ[s] == k) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = piv[(r0 + 4 * r) * kInvLd + c];
      } else if (tJ[s] == k) {   // i > k: G_ik <- T_i
        int o = tI[s] * kInvTile;
        asm volatile("" : "+s"(o));
        const double* src = tneg + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -src[(r0 + 4 * r) * kInvLd + c];
      } else {                   // j < k: G_kj <- T_j^T
        int o = tJ[s] * kInvTile;
        asm volatile("" : "+s"(o));
        const double* src = tneg + o;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = -src[c * kInvLd + r0 + 4 * r];
      }
    }
    CVD_INV_T(6);
    __syncthreads();
    CVD_INV_T(5);
  }

  // A^-1 = -G in f32: the tile as it lies, and its mirror image transposed through a private LDS tile so that both
  // stores run along rows (the panels are free now: the loop ended on a barrier)
  float* Mf = minv + static_cast<size_t>(f) * B * B;
  double* scratch = invSmem + w * kInvTile;
#pragma unroll
  for (int s = 0; s < TPW; ++s) {
    if (tI[s] < 0) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = kInvTS * tI[s] + r0 + 4 * r, j = kInvTS * tJ[s] + c;
      if (i < B && j < B) Mf[static_cast<size_t>(i) * B + j] = static_cast<float>(-acc[s][r]);
    }
    if (tI[s] != tJ[s]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) scratch[(r0 + 4 * r) * kInvLd + c] = acc[s][r];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = kInvTS * tI[s] + c, j = kInvTS * tJ[s] + r0 + 4 * r;  // element (i, j) of the tile -> M[j][i]
        const double v = scratch[c * kInvLd + r0 + 4 * r];
        if (i < B && j < B) Mf[static_cast<size_t>(j) * B + i] = static_cast<float>(-v);
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
#ifdef CVD_INV_PROFILE
  CVD_INV_T(7);
  if (f == 0 && lane == 0)
    for (int q = 0; q < 8; ++q) g_invProf[w * 8 + q] = prof[q];
#endif
}

// ---------------------------------------------------------------------------------------------------
// Block-Jacobi preconditioner: Minv_f = (H_ff + diag(lam_f))^-1.  One workgroup per frame, Cholesky of
// the packed lower triangle in LDS, L^-1 by column-parallel forward substitution (global scratch,
// L2-resident), Minv = L^-T L^-1.
// ---------------------------------------------------------------------------------------------------
inline __global__ __launch_bounds__(1024) void k_block_inverse(Layout L, const double* __restrict__ hBlocks,
                                                        const double* __restrict__ lam, float* __restrict__ minv,
                                                        double* __restrict__ work, int* __restrict__ fail) {
  // Everything stays in LDS (packed lower triangle A, B(B+1)/2 doubles + one column buffer):
  //   1. left-looking Cholesky: 4 lanes per row split the dot of two packed rows (quad shuffle reduce)
  //   2. X = L^-1 in place, right-to-left by columns: X[i][j] = -(sum_{k=j+1..i} X[i][k] L[k][j]) / L[j][j]
  //   3. Minv = X^T X written once to global as f32 (dense, symmetric).
  // blockDim = 4 * B rounded up to a wave multiple (<= 1024).
  extern __shared__ __attribute__((aligned(16))) double sm[];
  (void)work;
  const int B = L.B;
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const int nT = blockDim.x;
  const double* hf = hBlocks + static_cast<size_t>(f) * B * B;
  const int npk = B * (B + 1) / 2;
  double* A = sm;         // packed lower, row-major: (i, j) at i(i+1)/2 + j
  double* col = A + npk;  // B
  for (int idx = tid; idx < B * B; idx += nT) {
    const int i = idx / B, j = idx - i * B;
    if (j <= i) A[i * (i + 1) / 2 + j] = hf[idx] + (i == j ? lam[static_cast<size_t>(f) * B + i] : 0.0);
  }
  __syncthreads();
  const int row = tid >> 2, ln = tid & 3;
  // 1. Cholesky (column j finalised per step)
  for (int j = 0; j < B; ++j) {
    const int rj = j * (j + 1) / 2;
    const int i = j + row;
    double s = 0.0;
    if (i < B) {
      const int ri = i * (i + 1) / 2;
      double s0 = 0.0, s1 = 0.0;
      int k = ln;
      for (; k + 4 < j; k += 8) {
        s0 += A[ri + k] * A[rj + k];
        s1 += A[ri + k + 4] * A[rj + k + 4];
      }
      if (k < j) s0 += A[ri + k] * A[rj + k];
      s = s0 + s1;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (i < B && ln == 0) col[i] = A[i * (i + 1) / 2 + j] - s;  // un-normalised column j (col[j] = pivot^2)
    __syncthreads();
    double d = col[j];
    if (!(d > 0.0)) {
      if (tid == 0) atomicAdd(fail, 1);
      d = 1.0;
    }
    d = sqrt(d);
    const double id = 1.0 / d;
    for (int ii = j + tid; ii < B; ii += nT) A[ii * (ii + 1) / 2 + j] = (ii == j) ? d : col[ii] * id;
    __syncthreads();
  }
  // 2. in-place inverse of L
  for (int j = B - 1; j >= 0; --j) {
    for (int k = j + tid; k < B; k += nT) col[k] = A[k * (k + 1) / 2 + j];
    __syncthreads();
    const double ijj = 1.0 / col[j];
    const int i = j + row;
    double s = 0.0;
    if (i < B && i > j) {
      const int ri = i * (i + 1) / 2;
      double s0 = 0.0, s1 = 0.0;
      int k = j + 1 + ln;
      for (; k + 4 <= i; k += 8) {
        s0 += A[ri + k] * col[k];
        s1 += A[ri + k + 4] * col[k + 4];
      }
      if (k <= i) s0 += A[ri + k] * col[k];
      s = s0 + s1;
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    __syncthreads();  // every read of column j (through col) and of row entries is done before the overwrite
    if (i < B && ln == 0) A[i * (i + 1) / 2 + j] = (i == j) ? ijj : -s * ijj;
    __syncthreads();
  }
  // 3. Minv = X^T X
  float* Mf = minv + static_cast<size_t>(f) * B * B;
  for (int idx = tid; idx < npk; idx += nT) {
    int hi = static_cast<int>((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while ((hi + 1) * (hi + 2) / 2 <= idx) ++hi;
    while (hi * (hi + 1) / 2 > idx) --hi;
    const int lo = idx - hi * (hi + 1) / 2;
    double s0 = 0.0, s1 = 0.0;
    int k = hi;
    for (; k + 1 < B; k += 2) {
      const int rk = k * (k + 1) / 2, rk1 = rk + k + 1;
      s0 += A[rk + lo] * A[rk + hi];
      s1 += A[rk1 + lo] * A[rk1 + hi];
    }
    if (k < B) {
      const int rk = k * (k + 1) / 2;
      s0 += A[rk + lo] * A[rk + hi];
    }
    const float sv = static_cast<float>(s0 + s1);
    Mf[static_cast<size_t>(hi) * B + lo] = sv;
    Mf[static_cast<size_t>(lo) * B + hi] = sv;
  }
}

// ---------------------------------------------------------------------------------------------------
// Matrix-free product, pair-major (THE hot kernel): for the work item's constraints
//   t = rho' (J_a p_a + J_b p_b),  q_a += J_a^T t,  q_b += J_b^T t
// with J re-derived analytically from 24 B per constraint.  p is formed on the fly as z + beta p_old
// (beta lives on the device), masked for constant parameters.  Partials go to a per-item slot; the
// per-frame reduction (k_matvec_finish) is a gather, so there are no global atomics.
// ---------------------------------------------------------------------------------------------------
template <int KD, int KS>
inline __global__ __launch_bounds__(256) void k_matvec_pairs(Layout L, Table T, Items it, const double* __restrict__ x,
                                                      const FrameConst* __restrict__ fc,
                                                      const double* __restrict__ mask, const double* __restrict__ z,
                                                      const double* __restrict__ pOld,
                                                      const double* __restrict__ scal, int useBeta,
                                                      double* __restrict__ qPart, CoarseView V) {
  if (scal[S_DONE] != 0.0) return;  // PCG already converged (iterations enqueued ahead)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int B = L.B;
  double* xa = sm;
  double* xb = xa + B;
  double* pa = xb + B;
  double* pb = pa + B;
  double* qa = pb + B;
  double* qb = qa + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(qb + B);
  double* cl = reinterpret_cast<double*>(fcs + 2) + 18 + 4 * 24 + 8;  // 2 x kCB coarse corrections
  const int item = blockIdx.x;
  const int tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  // Prologue in ONE global round trip (the workgroup lives ~5 us, a dependent load costs ~1 us of it): every thread
  // issues its loads of x / z / p_old / mask for both frames, the coarse correction c_f (see CoarseView) and the
  // frame constants go to LDS in the same phase, and the search direction is formed after the barrier.  B <= 256:
  // one element per thread.
  constexpr int FCW = sizeof(FrameConst) / 8;
  if (V.Wb != nullptr) {  // (fused coarse variant: the correction is gathered here, one wave per frame)
    if (tid < 64) coarseFrameCorrection(V, fa, tid, cl);
    else if (tid < 128) coarseFrameCorrection(V, fb, tid - 64, cl + kCB);
  } else if (tid < 2 * kCB) {
    cl[tid] = (V.cF != nullptr) ? V.cF[(tid < kCB ? fa : fb) * kCB + (tid & (kCB - 1))] : 0.0;
  }
  if (tid >= 256 - 2 * FCW) {  // (the last waves: the first ones carry the coarse loads)
    const int t = tid - (256 - 2 * FCW);
    const int which = t / FCW, k = t % FCW;
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  // (two elements per thread: B <= 512 -- the generic kernel serves the blocks beyond the fast kernels' 256)
  double vza[2] = {0.0, 0.0}, vzb[2] = {0.0, 0.0}, vpa[2] = {0.0, 0.0}, vpb[2] = {0.0, 0.0}, vma[2] = {0.0, 0.0}, vmb[2] = {0.0, 0.0};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = tid + e * 256;
    if (i < B) {
      const size_t ia = static_cast<size_t>(fa) * B + i, ib = static_cast<size_t>(fb) * B + i;
      xa[i] = x[ia];
      xb[i] = x[ib];
      vza[e] = z[ia];
      vzb[e] = z[ib];
      if (useBeta) { vpa[e] = pOld[ia]; vpb[e] = pOld[ib]; }
      vma[e] = mask[ia];
      vmb[e] = mask[ib];
      qa[i] = 0.0;
      qb[i] = 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int i = tid + e * 256;
    if (i < B) {
      pa[i] = (vza[e] + coarseAtLds(cl, L, i) + (useBeta ? beta * vpa[e] : 0.0)) * vma[e];
      pb[i] = (vzb[e] + coarseAtLds(cl + kCB, L, i) + (useBeta ? beta * vpb[e] : 0.0)) * vmb[e];
    }
  }
  if (L.intrOpt == kIntrShared) {
    // every constraint's focal column is frame 0's slot (reference lib/PoseOptimizer.cpp:1226)
    __syncthreads();
    if (tid == 0) {
      const double p06 = (z[6] + (useBeta ? beta * pOld[6] : 0.0)) * mask[6];
      pa[6] = p06;
      pb[6] = p06;
    }
  }
  __syncthreads();
  for (int dir = 0; dir < 2; ++dir) {
  const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
  if (cb >= ce) continue;
  // role swap for the reverse pair: source = fb, target = fa
  const FrameConst& Fs = fcs[dir];
  const FrameConst& Ft = fcs[dir ^ 1];
  const double* xs = dir ? xb : xa;
  const double* xt = dir ? xa : xb;
  const double* ps = dir ? pb : pa;
  const double* pt = dir ? pa : pb;
  double* qs = dir ? qb : qa;
  double* qt = dir ? qa : qb;
  double qpa[7], qpb[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) { qpa[i] = 0.0; qpb[i] = 0.0; }
  for (long long c = cb + tid; c < ce; c += 256) {
    const float2 d = T.dsrc[c];
    if (!(d.x > 0.f)) continue;
    Sample<KD, KS> s;
    evalSample<KD, KS, true>(L, Fs, Ft, xs, xt, T.ndc[c], d, s);
    double t[3] = {0.0, 0.0, 0.0};
    sideJp(L, s.a, ps, t);
    sideJp(L, s.b, pt, t);
    t[0] *= s.rho1; t[1] *= s.rho1; t[2] *= s.rho1;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      qpa[i] += s.a.Jp[0][i] * t[0] + s.a.Jp[1][i] * t[1] + s.a.Jp[2][i] * t[2];
      qpb[i] += s.b.Jp[0][i] * t[0] + s.b.Jp[1][i] * t[1] + s.b.Jp[2][i] * t[2];
    }
    {
      const int nt = sideNumTapCols(L, s.a);
      for (int k = 0; k < nt; ++k) {
        int col;
        double J[3];
        sideTapCol(L, s.a, k, col, J);
        atomicAdd(&qs[col], J[0] * t[0] + J[1] * t[1] + J[2] * t[2]);
      }
    }
    {
      const int nt = sideNumTapCols(L, s.b);
      for (int k = 0; k < nt; ++k) {
        int col;
        double J[3];
        sideTapCol(L, s.b, k, col, J);
        atomicAdd(&qt[col], J[0] * t[0] + J[1] * t[1] + J[2] * t[2]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    qpa[i] = waveSum(qpa[i]);
    qpb[i] = waveSum(qpb[i]);
  }
  if ((tid & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      atomicAdd(&qs[i], qpa[i]);
      atomicAdd(&qt[i], qpb[i]);
    }
  }
  }  // dir
  __syncthreads();
  // rows are grouped by frame (slot = position in the frame's item list) so that k_matvec_finish streams them
  double* outA = qPart + static_cast<size_t>(it.slot[item * 2]) * B;
  double* outB = qPart + static_cast<size_t>(it.slot[item * 2 + 1]) * B;
  for (int i = tid; i < B; i += 256) {
    outA[i] = qa[i];
    outB[i] = qb[i];
  }
}

// Per frame: p_f = z + beta p_old (stored for the next iteration), q_f = mask * (sum of partials +
// regulariser J^T J p) + lam * p_f, and the frame's share of p.q.
// Regulariser Jacobian cache: the rows of the per-frame regularisers depend on x only, so they are evaluated once per
// linearisation point (RegCache, filled by k_reg_cache) instead of once per product.  Entry a of residual i of
// frame f lives at (f * stride + a) * nr + i (consecutive threads = consecutive residuals).
struct RegCache {
  double* jac;
  unsigned short* col;
  unsigned char* cnt;
  int nr;      // residuals per frame
  int stride;  // entries per residual
};

template <int KD>
inline __global__ __launch_bounds__(256) void k_reg_cache(Layout L, const double* __restrict__ x,
                                                   const float* __restrict__ median,
                                                   const unsigned char* __restrict__ owner, RegCache rc) {
  extern __shared__ __attribute__((aligned(16))) double sm[];  // the frame's parameters (see k_cost_frames)
  const int f = blockIdx.x;
  if (!owner[f]) return;
  for (int i = threadIdx.x; i < L.B; i += 256) sm[i] = x[static_cast<size_t>(f) * L.B + i];
  __syncthreads();
  const float med = median[f];
  for (int i = threadIdx.x; i < rc.nr; i += 256) {
    double r;
    int n;
    int cols[2 * KD + 2];
    double jac[2 * KD + 2];
    regResidual<KD>(L, f, i, sm, med, r, n, cols, jac);
    rc.cnt[static_cast<size_t>(f) * rc.nr + i] = static_cast<unsigned char>(n);
    for (int a = 0; a < n; ++a) {
      const size_t e = (static_cast<size_t>(f) * rc.stride + a) * rc.nr + i;
      rc.jac[e] = jac[a];
      rc.col[e] = static_cast<unsigned short>(cols[a]);
    }
  }
}

// Direction and product element `tid` of the frame (B <= 256), handed from the finish half to the update half of the fused
// tail kernel (k_pcg_tail) in registers.
struct TailCarry {
  double pv, qv;
};

// Body of k_matvec_finish.  FUSED = false: the kernel of that name (256 threads per frame).  FUSED = true: the first half of
// k_pcg_tail -- the workgroup may be larger than 256 threads (threads beyond B idle through the barriers), nothing is handed to
// a last workgroup (the caller's grid barrier follows), Z^T q is PUBLISHED (agent-scope stores: other workgroups of the same
// launch read it), and the direction / product of element tid stay in registers (carry).  Returns false when the PCG has
// converged already (uniform over the launch; nothing was written).
template <int KD, bool FUSED>
__device__ __forceinline__ bool matvecFinishBody(const Layout& L, const double* __restrict__ x,
                                                 const double* __restrict__ mask,
                                                 const double* __restrict__ lam, const float* __restrict__ median,
                                                 const unsigned char* __restrict__ inRange,
                                                 const unsigned char* __restrict__ rangeFlags,
                                                 const int* __restrict__ fiOff, const int* __restrict__ fiList,
                                                 const double* __restrict__ qPart, const double* __restrict__ z,
                                                 const double* __restrict__ pOld, double* __restrict__ pNew,
                                                 double* __restrict__ scal, unsigned int* __restrict__ counter,
                                                 int useBeta, double* __restrict__ q, double* __restrict__ fdot,
                                                 int distMode, int nRows, const RegCache& rc, const CoarseView& V,
                                                 double* __restrict__ qc, const CoarseColumns& cc,
                                                 const double* __restrict__ Hdiag, double* __restrict__ pqOut,
                                                 int ownFirst, int ownCount, double* __restrict__ sm, TailCarry& carry,
                                                 const TlStep* __restrict__ tsp, double* __restrict__ tlPart) {
  // Hdiag != nullptr (explicit cross blocks, cvd_cross.h): the partial rows hold the OFF-diagonal blocks' products only;
  // the frame-diagonal part, regularisers included, is H_ff p_f with the assembled H_ff.
  const double sDone = scal[S_DONE];  // PCG already converged (iterations enqueued ahead): tested after the input loads
  const int nT = FUSED ? static_cast<int>(blockDim.x) : 256;  // stride of the loops that are not bounded by B
  const int B = L.B;
  double* xf = sm;
  double* pf = xf + B;   // masked direction
  double* qf = pf + B;
  double* red = qf + B;
  double* cl = red + 8;  // kCB coarse corrections of this frame
  const int f = blockIdx.x;
  const int tid = threadIdx.x;
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  const size_t base = static_cast<size_t>(f) * B;
  // One global round trip for the inputs (B <= 256: one element per thread): the loads of z / mask / p_old / x / lam
  // are issued together with the coarse correction's, the direction is formed after the barrier and stays in a
  // register for the damping term and p.q at the end.
  if (V.Wb != nullptr) {
    if (tid < 64) coarseFrameCorrection(V, f, tid, cl);
  } else if (tid < kCB) {
    cl[tid] = (V.cF != nullptr) ? V.cF[f * kCB + tid] : 0.0;
  }
  // (two elements per thread: B <= 512; the fused kernel's scope ends at B = 256: one)
  constexpr int EPT = FUSED ? 1 : 2;
  double vz[2] = {0.0, 0.0}, vm[2] = {0.0, 0.0}, vp[2] = {0.0, 0.0}, vlam[2] = {0.0, 0.0}, pvReg[2] = {0.0, 0.0};
  // third level (CoarseView::tl): the frame's coefficients to LDS (behind the row walk's partial sums), the vertex taps to registers
  const bool tlOn = V.tl != nullptr;
  double* tls = cl + kCB + (FUSED ? ((nT >> 8) - 1) * 256 : 0);
  TlTaps tt[EPT];
  // (... and the thread's first entry of the transposed table for the restriction at the end: no load left behind the product)
  float tlW0 = 0.f;
  int tlV0 = 0, tlEntries = 0;
  if (tlOn) {
    if (tid < V.tlS) tls[tid] = V.tl[static_cast<size_t>(f) * V.tlS + tid];
#pragma unroll
    for (int e = 0; e < EPT; ++e) tlLoadTaps(V, tid + e * 256, L.nD, tt[e]);
  }
  if (tsp != nullptr) {
    tlEntries = tsp->S * tsp->width;
    if (tid < tlEntries) {
      tlW0 = tsp->elW[tid];
      tlV0 = tsp->elV[tid];
    }
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * 256;
    if (i < B) {
      vz[e] = z[base + i];
      vm[e] = mask[base + i];
      if (useBeta) vp[e] = pOld[base + i];
      vlam[e] = lam[base + i];
      xf[i] = x[base + i];
    }
  }
  const int e0 = fiOff[f], e1 = fiOff[f + 1];
  // The frame's partial rows (contiguous; independent streaming loads, four in flight per thread) are summed BEFORE the
  // barrier: they depend on nothing but the row range, so their way from L2 overlaps the vector loads above instead of
  // following them (this kernel is a chain of dependent round trips, not a bandwidth problem).
  double rowSum[2] = {0.0, 0.0};
  double* psum = cl + kCB;  // FUSED: (blockDim / 256 - 1) x 256 partial sums of the row walk
  if (L.includeStatic) {
    if constexpr (FUSED) {
      // every 256 threads of the (larger) workgroup walk their own residue class of the rows: the walk is a chain of dependent
      // round trips -- 40 rows of a hub frame of the hierarchical flow list = 10 trips of 4 loads -- and that chain, not the
      // bytes, was the fused kernel's critical path (the slowest frame reached the grid barrier after 25 us)
      const int nParts = nT >> 8, part = tid >> 8, i = tid & 255;
      if (i < B && part < nParts) {
        const size_t step = static_cast<size_t>(nParts) * B;
        const double* rowp = qPart + static_cast<size_t>(e0 + part) * B + i;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = e0 + part;
        for (; r + 3 * nParts < e1; r += 4 * nParts, rowp += 4 * step) {
          a0 += rowp[0];
          a1 += rowp[step];
          a2 += rowp[2 * step];
          a3 += rowp[3 * step];
        }
        for (; r < e1; r += nParts, rowp += step) a0 += rowp[0];
        const double t = (a0 + a1) + (a2 + a3);
        if (part == 0) rowSum[0] = t;
        else psum[(part - 1) * 256 + i] = t;
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * 256;
        if (i >= B) continue;
        const double* rowp = qPart + static_cast<size_t>(e0) * B + i;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = e0;
        for (; r + 3 < e1; r += 4, rowp += 4 * B) {
          a0 += rowp[0];
          a1 += rowp[B];
          a2 += rowp[2 * B];
          a3 += rowp[3 * B];
        }
        for (; r < e1; ++r, rowp += B) a0 += rowp[0];
        rowSum[e] = (a0 + a1) + (a2 + a3);
      }
    }
  }
  __syncthreads();
  if (sDone != 0.0) return false;  // uniform; nothing has been written to global memory yet
  if constexpr (FUSED) {
    if (L.includeStatic && tid < B)
      for (int k = 0; k + 1 < (nT >> 8); ++k) rowSum[0] += psum[k * 256 + tid];
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * 256;
    if (i >= B) continue;
    // search direction from the two-level preconditioned residual z + Z c (coarse part only on active unknowns)
    const double pv = vz[e] + (coarseAtLds(cl, L, i) + (tlOn ? tlAt(tls, tt[e]) : 0.0)) * vm[e] + (useBeta ? beta * vp[e] : 0.0);
    pNew[base + i] = pv;
    pvReg[e] = pv;
    pf[i] = pv * vm[e];
    // (no pair kernel ran: the partial buffer is stale; shared focal: frame 0's slot collects every row's entry below)
    qf[i] = (L.includeStatic && !(L.intrOpt == kIntrShared && i == 6)) ? rowSum[e] : 0.0;
  }
  __syncthreads();
  if (L.intrOpt == kIntrShared && f == 0 && L.includeStatic) {
    // shared focal: frame 0's slot collects the focal adjoint of EVERY partial row (both sides of every pair item, the
    // three rows of every triplet group)
    double a = 0.0;
    for (int k = tid; k < nRows; k += nT) a += qPart[static_cast<size_t>(k) * B + 6];
    a = waveSum(a);
    if ((tid & 63) == 0) atomicAdd(&qf[6], a);
    __syncthreads();
  }
  // (pair-sharded run: the reduced H_ff lives on the frame's OWNER rank only -- the others contribute the cross blocks of
  // their pairs and nothing else for this frame)
  if (Hdiag != nullptr && f >= ownFirst && f < ownFirst + ownCount) {
    // symmetric block: column access, coalesced over the row index; four independent loads in flight
    const double* Hf = Hdiag + static_cast<size_t>(f) * B * B;
    for (int i = tid; i < B; i += 256) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      int j = 0;
      for (; j + 3 < B; j += 4) {
        a0 += Hf[static_cast<size_t>(j) * B + i] * pf[j];
        a1 += Hf[static_cast<size_t>(j + 1) * B + i] * pf[j + 1];
        a2 += Hf[static_cast<size_t>(j + 2) * B + i] * pf[j + 2];
        a3 += Hf[static_cast<size_t>(j + 3) * B + i] * pf[j + 3];
      }
      for (; j < B; ++j) a0 += Hf[static_cast<size_t>(j) * B + i] * pf[j];
      qf[i] += (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
  } else if (Hdiag == nullptr && inRange[f]) {
    // J_reg^T (J_reg p) from the cached rows (k_reg_cache)
    // (deterministic build: the first wave alone -- the rows' LDS atomics then land in program order)
    const int regStride = CVD_DETERMINISTIC ? 64 : nT;
    for (int i = tid; i < rc.nr && tid < regStride; i += regStride) {
      const int n = rc.cnt[static_cast<size_t>(f) * rc.nr + i];
      const size_t e0 = static_cast<size_t>(f) * rc.stride * rc.nr + i;
      double t = 0.0;
      for (int a = 0; a < n; ++a) t += rc.jac[e0 + static_cast<size_t>(a) * rc.nr] * pf[rc.col[e0 + static_cast<size_t>(a) * rc.nr]];
      for (int a = 0; a < n; ++a)
        atomicAdd(&qf[rc.col[e0 + static_cast<size_t>(a) * rc.nr]], rc.jac[e0 + static_cast<size_t>(a) * rc.nr] * t);
    }
  }
  if (L.positionRegSqrt > 0.0 && tid < 3) {
    // neighbours' directions are re-formed from z / p_old (their pNew rows are being written concurrently)
    const double w = L.positionRegSqrt * L.positionRegSqrt;
    const double cf[3] = {1.0, -2.0, 1.0};
    double acc = 0.0;
    for (int o = 0; o < 3; ++o) {
      const int k = f - o;
      if (k < 0 || !posRegValid(L, rangeFlags, k)) continue;
      double a = 0.0;
      for (int j = 0; j < 3; ++j) {
        const size_t idx = static_cast<size_t>(k + j) * B + tid;
        a += cf[j] * (z[idx] + coarseAt(V.cF, L, k + j, tid) + (useBeta ? beta * pOld[idx] : 0.0)) * mask[idx];
      }
      acc += w * cf[o] * a;
    }
    atomicAdd(&qf[tid], acc);
  }
  __syncthreads();
  // distMode (pair-sharded multi-GPU): 1 = this rank adds the damping term, 2 = it does not; q is all-reduced afterwards.
  // pqOut == nullptr: p.q / alpha are formed by k_dot_pq on the reduced vector.  pqOut != nullptr (FUSED exchange): the
  // product, its restriction Z^T q and p.q are all linear in q, so this rank's shares of the three travel in ONE
  // all-reduce ([q | Z^T q | p.q] contiguous) and k_cg_update forms alpha from the reduced p.q itself.
  if (!FUSED && distMode && pqOut == nullptr) {
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int i = tid + e * 256;
      if (i < B) q[base + i] = qf[i] * vm[e] + (distMode == 1 ? vlam[e] * pvReg[e] : 0.0);
    }
    return true;
  }
  double dot = 0.0;
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * 256;
    if (i >= B) continue;
    const double pv = pvReg[e];
    const double qv = qf[i] * vm[e] + (distMode == 2 ? 0.0 : vlam[e] * pv);
    q[base + i] = qv;
    qf[i] = qv;
    dot += pv * qv;
    if (e == 0) { carry.pv = pv; carry.qv = qv; }
  }
  dot = waveSum(dot);
  if ((tid & 63) == 0 && tid < 256) red[tid >> 6] = dot;  // (waves beyond the first four hold no element: B <= 256 when FUSED)
  __syncthreads();
  if (tid == 0) publishPartial(fdot + f, red[0] + red[1] + red[2] + red[3]);
  if (qc != nullptr) {  // Z^T q and this frame's column of W (Z^T q) for the fused y update (CoarseStep)
    coarseRestrict<FUSED>(L, qf, f, tid, V.modeActive, qc);
    if constexpr (!FUSED) {
      __syncthreads();
      coarseColumnProducts(cc, qc + f * kCB, f, tid, 256);
    }
  }
  if (tsp != nullptr) {
    // third level: spatial restriction of this frame's product, sq[f][s] = sum_v Hs[v][s] q_f[7 + v], by the transposed vertex
    // table (entry k of hat s; fixed summation order).  (The level's descriptor is read where it is used: as a kernel argument
    // its 24 scalar registers stayed live through both halves of k_pcg_tail and cost the update half spilled operands.)
    const TlStep ts = *tsp;
    if (tid < tlEntries) tlPart[tid] = static_cast<double>(tlW0) * qf[7 + tlV0];
    for (int e = tid + nT; e < tlEntries; e += nT) tlPart[e] = static_cast<double>(ts.elW[e]) * qf[7 + ts.elV[e]];
    __syncthreads();
    if (tid < ts.S) {
      double a = 0.0;
      for (int k = 0; k < ts.width; ++k) a += tlPart[k * ts.S + tid];
      storeMaybePublished(ts.sq + static_cast<size_t>(f) * ts.S + tid, a, FUSED);
    }
  }
  if constexpr (!FUSED) {
    // the last workgroup to arrive reduces p.q over the frames and publishes alpha for k_cg_update
    if (lastBlockArrivesLite(counter, L.F, reinterpret_cast<int*>(red + 6))) {
      const double pq = blockSumPartials(fdot, L.F, red);
      if (tid == 0) {
        if (pqOut != nullptr) {
          *pqOut = pq;  // (this rank's share: reduced with q)
        } else {
          scal[S_PQ] = pq;
          scal[S_ALPHA] = scal[S_RZ] / pq;
        }
      }
    }
  }
  return true;
}

template <int KD>
inline __global__ __launch_bounds__(256) void k_matvec_finish(Layout L, const double* __restrict__ x,
                                                       const double* __restrict__ mask,
                                                       const double* __restrict__ lam, const float* __restrict__ median,
                                                       const unsigned char* __restrict__ inRange,
                                                       const unsigned char* __restrict__ rangeFlags,
                                                       const int* __restrict__ fiOff, const int* __restrict__ fiList,
                                                       const double* __restrict__ qPart, const double* __restrict__ z,
                                                       const double* __restrict__ pOld, double* __restrict__ pNew,
                                                       double* __restrict__ scal, unsigned int* __restrict__ counter,
                                                       int useBeta, double* __restrict__ q, double* __restrict__ fdot,
                                                       int distMode, int nRows, RegCache rc, CoarseView V,
                                                       double* __restrict__ qc, CoarseColumns cc,
                                                       const double* __restrict__ Hdiag, double* __restrict__ pqOut,
                                                       int ownFirst, int ownCount, const TlStep* __restrict__ tsp) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  TailCarry carry;
  // (third level: its coefficients behind the coarse corrections, the restriction's products behind those)
  (void)matvecFinishBody<KD, false>(L, x, mask, lam, median, inRange, rangeFlags, fiOff, fiList, qPart, z, pOld, pNew, scal, counter,
                                    useBeta, q, fdot, distMode, nRows, rc, V, qc, cc, Hdiag, pqOut, ownFirst, ownCount, sm, carry, tsp,
                                    sm + 3 * L.B + 8 + kCB + kTlMaxS);
}

// p.q of the all-reduced product (multi-GPU only) + alpha, same last-workgroup pattern as k_matvec_finish.
inline __global__ __launch_bounds__(256) void k_dot_pq(Layout L, const double* __restrict__ p, const double* __restrict__ q,
                                                double* __restrict__ scal, unsigned int* __restrict__ counter,
                                                double* __restrict__ fdot, double* __restrict__ qc,
                                                const unsigned char* __restrict__ modeActive, CoarseColumns cc,
                                                const TlStep* __restrict__ tsp) {
  if (scal[S_DONE] != 0.0) return;  // PCG already converged (iterations enqueued ahead)
  __shared__ double red[8];
  __shared__ double prod[kTlMaxS * kTlMaxWidth];
  const int f = blockIdx.x, tid = threadIdx.x;
  const size_t base = static_cast<size_t>(f) * L.B;
  if (tsp != nullptr) {  // third level: spatial restriction of the all-reduced product (matvecFinishBody does it on the other paths)
    const TlStep ts = *tsp;
    const int nE = ts.S * ts.width;
    for (int e = tid; e < nE; e += 256) prod[e] = static_cast<double>(ts.elW[e]) * q[base + 7 + ts.elV[e]];
    __syncthreads();
    if (tid < ts.S) {
      double a = 0.0;
      for (int k = 0; k < ts.width; ++k) a += prod[k * ts.S + tid];
      ts.sq[static_cast<size_t>(f) * ts.S + tid] = a;
    }
  }
  if (qc != nullptr) {  // on the all-reduced q
    coarseRestrict(L, q + base, f, tid, modeActive, qc);
    __syncthreads();
    coarseColumnProducts(cc, qc + f * kCB, f, tid, 256);
  }
  double dot = 0.0;
  for (int i = tid; i < L.B; i += 256) dot += p[base + i] * q[base + i];
  dot = waveSum(dot);
  if ((tid & 63) == 0) red[tid >> 6] = dot;
  __syncthreads();
  if (tid == 0) publishPartial(fdot + f, red[0] + red[1] + red[2] + red[3]);
  if (lastBlockArrivesLite(counter, L.F, reinterpret_cast<int*>(red + 6))) {
    const double pq = blockSumPartials(fdot, L.F, red);
    if (tid == 0) {
      scal[S_PQ] = pq;
      scal[S_ALPHA] = scal[S_RZ] / pq;
    }
  }
}

// PCG scalars after the preconditioner has been applied (shared by k_cg_update and, with the coarse level, k_coarse_apply).
__device__ __forceinline__ void pcgFinishScalars(double* __restrict__ scal, int init, double rzs, double rrs, double tol2,
                                                 double* __restrict__ hostMirror) {
  double done, iters;
  if (init) {
    scal[S_RZ0] = rzs;
    scal[S_RZOLD] = rzs;
    scal[S_BETA] = 0.0;
    scal[S_TARGET] = tol2 * rzs;
    iters = 0.0;
    done = (rzs == rzs) ? ((rzs > 0.0) ? 0.0 : 1.0) : 2.0;
  } else {
    const double old = scal[S_RZ];
    scal[S_RZOLD] = old;
    scal[S_BETA] = (old != 0.0) ? rzs / old : 0.0;
    iters = scal[S_ITERS] + 1.0;
    done = scal[S_DONE];
    if (!(rzs == rzs)) done = 2.0;
    else if (rzs <= scal[S_TARGET]) done = 1.0;
  }
  scal[S_ITERS] = iters;
  scal[S_DONE] = done;
  scal[S_RZ] = rzs;
  scal[S_RR] = rrs;
  // Progress for the host in pinned, coherent memory, so that it can bound its run-ahead without any copy or event
  // in the stream: slot 1 + (iterations applied & 7) = 4 (iterations applied + 1) + the done flag after exactly that many
  // iterations (a ring: the host's decisions depend on the iteration count only, never on timing, which keeps the ranks
  // of a multi-GPU run enqueuing the same collectives).  ONE relaxed store carries both numbers: a release store after a
  // separate flag store was a system-scope write-back of the XCD's L2 (buffer_wbl2 sc0 sc1) at the end of every iteration.
  if (hostMirror != nullptr)
    __hip_atomic_store(hostMirror + 1 + (static_cast<int>(iters) & 7), 4.0 * (iters + 1.0) + done, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}

// LDS doubles of k_cg_update's partial row sums (host and device agree on the layout)
__host__ __device__ inline int cgUpdatePartDoubles(int B, int nThreads) {
  return B > 256 ? nThreads : max(nThreads, 16 * ((B + 3) & ~3));
}

// Workgroup of the third level (TlStep) for coarse hat s: q_T = R_t sq over the frames of every temporal node (all NT entries: the
// rows need them all), the hat's nn rows of the inverse (one wave per row), t <- t - alpha A_T^-1 q_T, r_T <- r_T - alpha q_T, the
// hat's share of r^T P t, and the temporal interpolation of the new t for every frame (tl).  init: t = A_T^-1 q_T, r_T = q_T with
// sq = the restriction of the first residual.  LDS: NT + 2 nn doubles at sm.  Returns false when nothing was written.
#ifdef CVD_TAIL_PROFILE  // tools/tail_profile.py: wall-clock (100 MHz) stamps of k_pcg_tail's workgroups
__device__ unsigned long long g_tailProf[1024 * 8];
#define TAIL_STAMP(slot) do { if (threadIdx.x == 0) g_tailProf[(blockIdx.x & 1023) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define TAIL_STAMP(slot) do {} while (0)
#endif

struct NoMid {  // (the kernels that are not k_pcg_tail: nothing happens between the halves)
  __device__ bool first(double&, double&) { return true; }
  __device__ bool second(double&) { return true; }
  __device__ unsigned int generation() { return 0u; }
};
typedef unsigned int cvd_u32x4 __attribute__((ext_vector_type(4)));
template <bool FUSED, typename Mid>
__device__ __forceinline__ bool tlLevelRows(const TlStep* __restrict__ tsp, int wg, int F, double& alpha, int init, double sDone,
                                            const double* __restrict__ scal, double* __restrict__ sm, Mid& mid) {
  const TlStep ts = *tsp;
  const int tid = threadIdx.x, nThreads = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, nW = nThreads >> 6;
  // Workgroup = (hat s, node range): nodes [aLo, aHi] with aHi shared with the next range (both walk its row -- the frames between
  // two nodes need both coefficients -- the next range OWNS it: writes it, counts it).  One row per wave: span + 1 <= waves.
  const int s = wg / ts.parts, part = wg - s * ts.parts;
  const int aLo = part * ts.span, aHi = min(ts.nn - 1, aLo + ts.span), nA = aHi - aLo + 1;
  const bool ownsLast = part == ts.parts - 1;
  // t is double-buffered (a shared row's old coefficient is read by two workgroups while its owner writes the new one): iteration
  // k (S_ITERS = k - 1 applied) reads buffer (k - 1) & 1 and writes the other; the first residual writes buffer 0
  const int par = init ? 1 : (static_cast<int>(scal[S_ITERS]) & 1);
  const double* tin = ts.t + static_cast<size_t>(par) * ts.NT;
  double* tout = ts.t + static_cast<size_t>(par ^ 1) * ts.NT;
  double* qT = sm;                 // [S][nn]
  double* tn = qT + ts.NT;         // span + 1 new coefficients of this hat
  double* red = tn + ts.span + 1;  // span + 1 products t r_T, [span + 1 .. ]: flag
  // (the coefficients the wave's first row updates depend on nothing this launch computes: requested before the grid barrier.
  // The row itself as well -- 8 doubles per lane -- spills at the 80 registers of k_pcg_tail.)
  double rOldPre = 0.0, tOldPre = 0.0;
  if (lane == 0 && !init) {
    const int e = s * ts.nn + aLo + (wv < nA ? wv : 0);
    rOldPre = ts.rT[e];
    tOldPre = tin[e];
  }
  if constexpr (FUSED) {
    if (!mid.second(alpha)) return false;
  }
  const double inv = 1.0 / static_cast<double>(ts.step);
  bool direct = true;
  if constexpr (FUSED) {
    if (ts.rec != nullptr) {
      // Node sums through the other workgroups of the level: every hat's workgroup sums ITS hat over the frames of each node (one
      // load per lane, one wave per node), the first range's publishes the nn sums as 16-byte records {generation, value,
      // generation}, and everybody collects the NT records -- 1 + 1 round trips instead of (2 step - 1) / 8 batches of loads
      // of the whole sq.  The generation is the grid barrier's (unique per launch; records zeroed with the barrier's words).
      direct = false;
      const unsigned int gen = mid.generation() + 1u;
      cvd_u32x4* rec = reinterpret_cast<cvd_u32x4*>(ts.rec);
      for (int a = wv; a < ts.nn; a += nW) {
        const int f = (a - 1) * ts.step + 1 + lane;
        const bool in = lane < 2 * ts.step - 1 && f >= 0 && f < F;
        double v = in ? readPartial(ts.sq + static_cast<size_t>(f) * ts.S + s) : 0.0;
        v *= 1.0 - fabs(static_cast<double>(f - a * ts.step)) * inv;
        v = waveSum(in ? v : 0.0);
        if (lane == 0 && part == 0) {
          cvd_u32x4 r4;
          r4.x = gen;
          r4.y = static_cast<unsigned int>(__double2loint(v));
          r4.z = static_cast<unsigned int>(__double2hiint(v));
          r4.w = gen;
          *reinterpret_cast<volatile cvd_u32x4*>(rec + s * ts.nn + a) = r4;
        }
      }
      int* bad = reinterpret_cast<int*>(red + ts.span + 1);
      if (tid == 0) *bad = 0;
      __syncthreads();
      for (int e = tid; e < ts.NT; e += nThreads) {
        const volatile cvd_u32x4* src = reinterpret_cast<const volatile cvd_u32x4*>(rec + e);
        cvd_u32x4 r4;
        unsigned int spins = 0;
        for (;;) {
          r4 = *src;
          if (r4.x == gen && r4.w == gen) break;
          __builtin_amdgcn_s_sleep(4);
          if (++spins > (1u << 20)) {  // (a workgroup of the level is not resident: give up, the host reports the stalled PCG)
            *bad = 1;
            break;
          }
        }
        qT[e] = __hiloint2double(static_cast<int>(r4.z), static_cast<int>(r4.y));
      }
      __syncthreads();
      if (*bad) return false;
    }
  }
  if (direct) {
    for (int e = tid; e < ts.NT; e += nThreads) {  // e = a * S + s': neighbouring lanes read neighbouring words of a frame's row
      const int a = e / ts.S, sp = e - a * ts.S;
      const int fLo = max(0, (a - 1) * ts.step + 1), fHi = min(F - 1, (a + 1) * ts.step - 1);
      // (a node has up to 2 step - 1 frames: eight independent loads per batch, the walk is latency, not bytes)
      constexpr int kTlBatch = 8;
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int f = fLo; f <= fHi; f += kTlBatch) {
        double v[kTlBatch];
#pragma unroll
        for (int u = 0; u < kTlBatch; ++u) {
          const double* src = ts.sq + static_cast<size_t>(min(f + u, fHi)) * ts.S + sp;
          v[u] = FUSED ? readPartial(src) : *src;
        }
#pragma unroll
        for (int u = 0; u < kTlBatch; ++u) {
          const double w = 1.0 - fabs(static_cast<double>(f + u - a * ts.step)) * inv;
          acc[u & 3] += (f + u <= fHi) ? w * v[u] : 0.0;
        }
      }
      qT[sp * ts.nn + a] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    __syncthreads();
  }
  if (sDone != 0.0) return false;  // uniform; nothing written yet
  if constexpr (FUSED) TAIL_STAMP(5);
  const bool on = *ts.fail == 0;
  for (int k = wv; k < nA; k += nW) {
    const int a = aLo + k, e = s * ts.nn + a;
    const double* row = ts.Ainv + static_cast<size_t>(e) * ts.ld;
    const bool owned = ownsLast || a < aHi;
    // (what the row's update needs besides the product: requested with the row)
    double rOld = rOldPre, tOld = tOldPre;
    double acc0 = 0.0, acc1 = 0.0;
    int c = lane;
    if (k != wv && lane == 0 && !init) {
      rOld = ts.rT[e];
      tOld = tin[e];
    }
    for (; c + 64 < ts.NT; c += 128) {
      acc0 += row[c] * qT[c];
      acc1 += row[c + 64] * qT[c + 64];
    }
    if (c < ts.NT) acc0 += row[c] * qT[c];
    const double d = ts.weight * waveSum(acc0 + acc1);
    if (lane == 0) {
      const double rn = init ? qT[e] : rOld - alpha * qT[e];
      const double tv = on ? (init ? d : tOld - alpha * d) : 0.0;
      if (owned) {
        ts.rT[e] = rn;
        tout[e] = tv;
      }
      tn[k] = tv;
      red[k] = owned ? tv * rn : 0.0;
    }
  }
  __syncthreads();
  if constexpr (FUSED) TAIL_STAMP(6);
  const int fLo = aLo * ts.step, fHi = ownsLast ? F : min(F, aHi * ts.step);
  for (int f = fLo + tid; f < fHi; f += nThreads) {
    const int a0 = f / ts.step, k0 = a0 - aLo;
    const double tau = static_cast<double>(f - a0 * ts.step) * inv;
    ts.tl[static_cast<size_t>(f) * ts.S + s] = (1.0 - tau) * tn[k0] + tau * tn[min(k0 + 1, nA - 1)];
  }
  if (tid == 0) {
    double d = 0.0;
    for (int k = 0; k < nA; ++k) d += red[k];
    publishPartial(ts.dotPart + wg, d);
  }
  if constexpr (FUSED) TAIL_STAMP(7);
  return true;
}

// alpha = rz / sum(p.q) (published by k_matvec_finish); dx += alpha p; r -= alpha q; z = Minv_f r;
// partial r.z and r.r.  One workgroup per frame with 256 threads per 64-row chunk (blockDim = 256 * ceil(B/64),
// B <= 256): thread = (row, j-segment); the 4 segments of a row split the block mat-vec and are combined in LDS.
// 256 < B <= 512: 128 threads per chunk (two segments per row), same layout otherwise.
// init != 0: dx = 0, r = -g (already masked), z = Minv r.
// Body of k_cg_update.  FUSED = false: the kernel of that name.  FUSED = true: the second half of k_pcg_tail -- the caller's
// `mid.first(pv, qv)` runs the finish half of the frame first (direction / product element handed over in registers; false =
// converged already), then the loads that depend on nothing this launch computes are requested (the thread's share of M_f^-1
// resp. of A_c^-1, r, dx) and `mid.second(alpha)` -- the grid barrier and alpha = r^T z / sum p.q -- runs while they are in
// flight (held across the finish half as well they cost 32 spilled registers at the 80 a 768-thread workgroup pair allows).
template <bool FUSED, typename Mid>
__device__ __forceinline__ void cgUpdateBody(const Layout& L, int init, const double* __restrict__ g,
                                             const float* __restrict__ minv, const double* __restrict__ p,
                                             const double* __restrict__ q, double* __restrict__ scal,
                                             unsigned int* __restrict__ counter, double* __restrict__ dx,
                                             double* __restrict__ r, double* __restrict__ z,
                                             double* __restrict__ fdotRZ, double* __restrict__ fdotRR,
                                             double tol2, double* __restrict__ rc,
                                             const unsigned char* __restrict__ modeActive,
                                             double* __restrict__ hostMirror, const CoarseStep& cs, const DenseStep& ds,
                                             const double* __restrict__ pqReduced, double* __restrict__ sm, Mid& mid,
                                             int f0, int nF, double* __restrict__ partOut, const TlStep* __restrict__ tsp,
                                             const TlStep* __restrict__ tpp) {
  // [f0, f0 + nF): the frames this launch updates -- all of them, or (owner-sharded multi-GPU iteration) the calling rank's own
  // chunk; workgroups beyond nF are the dense coarse level's, two frames of the window each.  partOut != nullptr: the last
  // workgroup leaves THIS RANK's shares {r^T z, r^T r} there instead of finishing the PCG scalars (k_pcg_scalars_dist does).
  const double sDone = (init || FUSED) ? 0.0 : scal[S_DONE];  // converged earlier: the iterations enqueued ahead are no-ops (tested below)
  const int B = L.B;
  const int nThreads = blockDim.x;
  double* rf = sm;                 // B
  double* part = rf + B;           // partial row sums: nThreads (B > 256) or 16 segments x 4 ceil(B / 4) rows
  double* red = part + cgUpdatePartDoubles(B, nThreads);   // 2 * 16 wave partials + 10
  double* ypart = red + 48;        // 16 waves x kCB partial sums of the fused y update + kCB squares
  const bool fusedY = !init && cs.Wb != nullptr;
  const bool fusedDense = !init && ds.Ainv != nullptr;  // (the grid then has F extra workgroups)
  const int nWaves = nThreads >> 6;
  // Logical workgroup index: [0, nF) the frames, then the levels' workgroups.  A launch of its own (not the fused tail, whose
  // workgroups are all resident) dispatches the LEVELS' workgroups first: theirs is the longest dependent chain of the launch, and
  // behind a thousand frame workgroups -- four rounds of the device at configs[4] -- it started when everything else was done.
  const int nExtraWg = static_cast<int>(gridDim.x) - nF;
  const int bid = FUSED ? static_cast<int>(blockIdx.x)
                        : (static_cast<int>(blockIdx.x) < nExtraWg ? nF + static_cast<int>(blockIdx.x) : static_cast<int>(blockIdx.x) - nExtraWg);
  const bool denseWg = bid >= nF;
  const int f = denseWg ? L.F + (bid - nF) : f0 + bid;
  // (third level: its S workgroups follow the dense level's)
  const int nDenseWg = (fusedDense && ds.rowSplit > 0) ? (nF + kDenseFramesPerGroup - 1) / kDenseFramesPerGroup : 0;
  // (... and the kCB of the temporal pose level, coarse_level 3, follow those: same rows routine, its own descriptor)
  const bool tlOn = !init && tsp != nullptr, tpOn = !init && tpp != nullptr;
  const bool tlWg = (tlOn || tpOn) && bid >= nF + nDenseWg;
  const int tid = threadIdx.x;
  const size_t base = static_cast<size_t>(f) * B;
  // (pqReduced: the fused exchange of the pair-sharded mode left the all-reduced p.q there; S_RZ is rewritten only by the
  // last workgroup to arrive, i.e. after every workgroup has read it here)
  double alpha = (init || FUSED) ? 0.0 : (pqReduced != nullptr ? scal[S_RZ] / *pqReduced : scal[S_ALPHA]);
  double pvCarry = 0.0, qvCarry = 0.0;
  if constexpr (FUSED) {
    // the finish half of this frame (dense-level workgroups: only the convergence flag); its registers are dead before the
    // update half requests its operands below
    if (!mid.first(pvCarry, qvCarry)) return;
  }
#ifndef CVD_DENSE_LOADS
#define CVD_DENSE_LOADS 4
#endif
  // (7 in the fused kernel -- two round trips per row instead of four -- spills at its 80-register budget: 12.9 -> 16.9 us)
  constexpr int kDenseLoads = CVD_DENSE_LOADS;
  const size_t nC = static_cast<size_t>(L.F) * kCB;   // coarse unknowns
  const int n2 = static_cast<int>(nC / 2);
  // ---- dense coarse level, rows [r0, r0 + nR) of frame g: d = (A_c^-1 Z^T q) (nThreads / nR threads per row; every thread's
  // <= kDenseLoads 16-byte loads of the (f64) inverse are issued at once, qc comes from LDS, the row sums are folded in LDS),
  // c <- c - alpha d, rc <- rc - alpha qc, and the rows' share of the coarse part of r^T z into *dotSlot.  All threads of the
  // workgroup call it.  w: the first batch of the first row walk when preloaded.
  auto denseRows = [&](int g, int r0, int nR, const double* qcs, double* psum, bool on0, double2 (&w)[kDenseLoads], bool preloaded,
                       double* dotSlot) {
    const int per = nThreads / nR, m = tid / per, part = tid - m * per;
    const bool active = m < nR;
    const double2* row = reinterpret_cast<const double2*>(ds.Ainv + (static_cast<size_t>(g) * kCB + r0 + (active ? m : 0)) * nC);
    const int e = g * kCB + r0 + (tid < nR ? tid : 0);
    const double qcv = qcs[e], rcOld = ds.rc[e], cOld = ds.c[e];
    const bool on = on0 && ds.modeActive[e];
    double acc0 = 0.0, acc1 = 0.0;
    // the row in BATCHES of kDenseLoads loads per thread, every batch's loads issued together
    for (int base = 0; base < n2; base += kDenseLoads * per) {
      if (!preloaded || base > 0) {
#pragma unroll
        for (int u = 0; u < kDenseLoads; ++u) {
          const int j = base + part + u * per;
          w[u] = row[j < n2 ? j : 0];
          if (j >= n2 || !active) w[u] = make_double2(0.0, 0.0);
        }
      }
#pragma unroll
      for (int u = 0; u < kDenseLoads; ++u) {
        const int j = base + part + u * per;
        const double2 qq = *reinterpret_cast<const double2*>(qcs + 2 * (j < n2 ? j : 0));
        acc0 += w[u].x * qq.x;
        acc1 += w[u].y * qq.y;
      }
    }
    psum[tid] = active ? acc0 + acc1 : 0.0;
    __syncthreads();
    if (tid < nR * 8) {  // 8 lanes per row fold its `per` partials, then three shuffle steps
      const int r = tid >> 3, l = tid & 7;
      double t = 0.0;
      for (int k = l; k < per; k += 8) t += psum[r * per + k];
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if (l == 0) psum[nThreads + r] = t;
    }
    __syncthreads();
    if (tid < 8) {  // (8 lanes shuffle together; rows beyond nR contribute nothing)
      double t = 0.0;
      if (tid < nR) {
        const double d = psum[nThreads + tid];
        const double rcn = rcOld - alpha * qcv;
        const double cn = on ? cOld - alpha * d : 0.0;
        ds.rc[e] = rcn;
        ds.c[e] = cn;
        t = cn * rcn;
      }
      t += __shfl_xor(t, 1, 64);
      t += __shfl_xor(t, 2, 64);
      t += __shfl_xor(t, 4, 64);
      if (tid == 0) publishPartial(dotSlot, t);
    }
    __syncthreads();  // (psum is reused)
  };
  if (f >= L.F) {
    if (tlWg) {  // third level's workgroup (one coarse hat) or the temporal pose level's (one mode)
      const int k = bid - nF - nDenseWg, nTl = tlOn ? tsp->S * tsp->parts : 0;
      if (!tlLevelRows<FUSED>(k < nTl ? tsp : tpp, k < nTl ? k : k - nTl, L.F, alpha, 0, sDone, scal, sm, mid)) return;
    } else {
    // ---- dense-level workgroup: rows [0, rowSplit) of kDenseFramesPerGroup frames, one frame after the other (F + F / 2
    // workgroups of 768 threads still fit the device in ONE round, two per CU; F + F do not)
    const int nR = ds.rowSplit;
    double* qcs = sm;                       // nC doubles (the frame workgroups' layout is not used here)
    double* psum = sm + nC;                 // nThreads partial sums + 8 row sums
    const int g0 = f0 + (f - L.F) * kDenseFramesPerGroup;
    double2 w[kDenseLoads];
    {
      const int per = nThreads / nR, m = tid / per, part = tid - m * per;
      const double2* row = reinterpret_cast<const double2*>(ds.Ainv + (static_cast<size_t>(g0) * kCB + (m < nR ? m : 0)) * nC);
#pragma unroll
      for (int u = 0; u < kDenseLoads; ++u) {
        const int j = part + u * per;
        w[u] = row[j < n2 ? j : 0];
        if (j >= n2 || m >= nR) w[u] = make_double2(0.0, 0.0);
      }
    }
    if constexpr (FUSED) {  // (the first batch of the inverse's row is in flight across the grid barrier)
      if (!mid.second(alpha)) return;
      for (int i = tid; i < static_cast<int>(nC); i += nThreads) qcs[i] = readPartial(ds.qc + i);  // (published by the finish halves)
    } else {
      for (int i = tid; i < static_cast<int>(nC); i += nThreads) qcs[i] = ds.qc[i];
    }
    const bool on0 = *ds.fail == 0;
    __syncthreads();
    if (sDone != 0.0) return;  // uniform; nothing written yet
    for (int rep = 0; rep < kDenseFramesPerGroup; ++rep) {
      const int g = g0 + rep;
      if (g >= f0 + nF) break;
      denseRows(g, 0, nR, qcs, psum, on0, w, rep == 0, ds.dotPart + g);
    }
    }  // dense-level workgroup
  } else {
  // B <= 256: the thread's share of the f32 block M_f^-1 (see the mat-vec below) is requested FIRST -- it depends on nothing
  // this launch computes, so its way from L2 / MALL overlaps the vector loads, the update and the barrier
  struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
  constexpr int kMinvLoads = 8;   // rows j = sg, sg + nSeg, ...: the first 8 of up to ceil(256 / 16) = 16 (16 spill at the
                                  // 128 registers a 1024-thread workgroup leaves a wave; the rest follows after the barrier)
  const bool wide = B > 256;  // (two segments per row: blockDim = 128 * ceil(B / 64) <= 1024)
  const float* Mf = minv + static_cast<size_t>(f) * B * B;
  const int nQ = (B + 3) >> 2;
  const int nSeg = min(16, nThreads / nQ);
  const int q4 = tid % nQ, sg = tid / nQ;
  F4u mreg[kMinvLoads];
  if (!wide && sg < nSeg) {
    const float* col = Mf + 4 * q4;
#pragma unroll
    for (int u = 0; u < kMinvLoads; ++u) {
      const int j = sg + u * nSeg;
      if (u * nSeg < B) mreg[u] = *reinterpret_cast<const F4u*>(col + static_cast<size_t>(min(j, B - 1)) * B);  // (uniform test)
    }
  }
  {
    // one element per thread (blockDim = 256 * ceil(B / 64) >= B).  The vector loads are issued together with the
    // scalars' and the convergence flag is tested once they are back: one dependent global round trip instead of two
    const int j = tid;
    double pv = 0.0, qv = 0.0, rv = 0.0, dv = 0.0;
    if (j < B) {
      if (init) {
        rv = -g[base + j];
      } else {
        if constexpr (!FUSED) {
          pv = p[base + j];
          qv = q[base + j];
        }
        rv = r[base + j];
        dv = dx[base + j];
      }
    }
    if constexpr (!FUSED) asm volatile("" : "+v"(pv), "+v"(qv), "+v"(rv), "+v"(dv));  // (keeps the loads above the early exit)
    if constexpr (FUSED) {
      pv = pvCarry;
      qv = qvCarry;
      if (!mid.second(alpha)) return;  // grid barrier + alpha (the loads above are in flight meanwhile)
    } else {
      if (sDone != 0.0) return;  // uniform; nothing written yet
    }
    if (j < B) {
      if (!init) {
        dv += alpha * pv;
        rv -= alpha * qv;
      }
      dx[base + j] = dv;
      r[base + j] = rv;
      rf[j] = rv;
    }
  }
  __syncthreads();
  if (rc != nullptr) coarseRestrict(L, rf, f, tid, modeActive, rc);  // Z_f^T r_f (first residual: k_coarse_apply_w)
  if (fusedY) {
    // row f of W (Z^T q): the 8-vectors W_block (Z^T q)_column were left in row order by coarseColumnProducts (in the
    // kernel that completed q), so the row is a contiguous run: a wave sums eight entries per load, lane = (entry, mode)
    const int wv = tid >> 6, lane = tid & 63, nWv = nThreads >> 6;
    const size_t e0 = static_cast<size_t>(cs.wtPtr[f]) * kCB, e1 = static_cast<size_t>(cs.wtPtr[f + 1]) * kCB;
    double a0 = 0.0, a1 = 0.0;
    size_t e = e0 + static_cast<size_t>(wv) * 64 + lane;
    const size_t stride = static_cast<size_t>(nWv) * 64;
    for (; e + stride < e1; e += 2 * stride) {
      a0 += cs.wq[e];
      a1 += cs.wq[e + stride];
    }
    if (e < e1) a0 += cs.wq[e];
    double ya = a0 + a1;
    ya += __shfl_xor(ya, 8, 64);
    ya += __shfl_xor(ya, 16, 64);
    ya += __shfl_xor(ya, 32, 64);
    if (lane < kCB) ypart[wv * kCB + lane] = ya;
  }
  // preconditioner blocks are stored in f32 (an SPD approximation is all PCG needs; halves the traffic),
  // applied with f64 accumulation.  Symmetric block: column access, coalesced over the row index.
  const int chunk = wide ? tid >> 7 : tid >> 8, row = tid & 63, seg = wide ? (tid >> 6) & 1 : (tid >> 6) & 3;
  const int i = chunk * 64 + row;
  double acc = 0.0;
  if (wide) {
    if (i < B) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
      const float* col = Mf + i;
      int j = seg;
      for (; j + 6 < B; j += 8) {
        a0 += static_cast<double>(col[static_cast<size_t>(j) * B]) * rf[j];
        a1 += static_cast<double>(col[static_cast<size_t>(j + 2) * B]) * rf[j + 2];
        a2 += static_cast<double>(col[static_cast<size_t>(j + 4) * B]) * rf[j + 4];
        a3 += static_cast<double>(col[static_cast<size_t>(j + 6) * B]) * rf[j + 6];
      }
      for (; j < B; j += 2) a0 += static_cast<double>(col[static_cast<size_t>(j) * B]) * rf[j];
      acc = (a0 + a1) + (a2 + a3);
    }
  } else {
    // Thread = (QUAD of rows 4 q .. 4 q + 3, j-segment): one 16-byte load fetches the quad's entries of block row j (the
    // block is symmetric: row j = column j; consecutive lanes = consecutive quads, so a wave reads contiguous bytes of a
    // row), up to 16 independent loads per thread all in flight.  4 B per lane and load (thread = one row) made this half of
    // the launch latency-bound at 2.4 TB/s of L2/MALL-resident data.  The rows are only 4-byte aligned (B is odd in every
    // level of the default schedule): global dwordx4 loads need no more.  A quad that reaches past the end of a row reads
    // the head of the next one (past the last frame: the allocation's slack) into lanes whose sums are never used.
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (sg < nSeg) {
#pragma unroll
      for (int u = 0; u < kMinvLoads; ++u) {
        const int j = sg + u * nSeg;
        if (u * nSeg < B) {
          const double rj = j < B ? rf[j] : 0.0;
          a0 += static_cast<double>(mreg[u].x) * rj;
          a1 += static_cast<double>(mreg[u].y) * rj;
          a2 += static_cast<double>(mreg[u].z) * rj;
          a3 += static_cast<double>(mreg[u].w) * rj;
        }
      }
      if (kMinvLoads * nSeg < B) {  // (uniform: B > 128 at 16 segments) second batch, again all loads first
        const float* col = Mf + 4 * q4;
#pragma unroll
        for (int u = 0; u < kMinvLoads; ++u) {
          const int j = sg + (kMinvLoads + u) * nSeg;
          if ((kMinvLoads + u) * nSeg < B) mreg[u] = *reinterpret_cast<const F4u*>(col + static_cast<size_t>(min(j, B - 1)) * B);
        }
#pragma unroll
        for (int u = 0; u < kMinvLoads; ++u) {
          const int j = sg + (kMinvLoads + u) * nSeg;
          if ((kMinvLoads + u) * nSeg < B) {
            const double rj = j < B ? rf[j] : 0.0;
            a0 += static_cast<double>(mreg[u].x) * rj;
            a1 += static_cast<double>(mreg[u].y) * rj;
            a2 += static_cast<double>(mreg[u].z) * rj;
            a3 += static_cast<double>(mreg[u].w) * rj;
          }
        }
      }
      double* dst = part + static_cast<size_t>(sg) * (4 * nQ) + 4 * q4;
      dst[0] = a0; dst[1] = a1; dst[2] = a2; dst[3] = a3;
    }
    __syncthreads();
    if (tid < B) {
      double zv = 0.0;
      for (int k2 = 0; k2 < nSeg; ++k2) zv += part[static_cast<size_t>(k2) * (4 * nQ) + tid];
      acc = zv;
    }
  }
  if (wide) part[tid] = acc;
  __syncthreads();
  double rz = 0.0, rr = 0.0;
  if (wide) {
    if (seg == 0 && i < B) {
      const int b0 = chunk * 128 + row;
      const double zv = part[b0] + part[b0 + 64];
      z[base + i] = zv;
      rz = rf[i] * zv;
      rr = rf[i] * rf[i];
    }
  } else if (tid < B) {
    z[base + tid] = acc;
    rz = rf[tid] * acc;
    rr = rf[tid] * rf[tid];
  }
  rz = waveSum(rz);
  rr = waveSum(rr);
  if ((tid & 63) == 0) { red[tid >> 6] = rz; red[16 + (tid >> 6)] = rr; }
  __syncthreads();
  if (tid == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < nWaves; ++w) { a += red[w]; b += red[16 + w]; }
    publishPartial(fdotRZ + f, a);
    publishPartial(fdotRR + f, b);
  }
  if (fusedY) {
    // y_f <- y_f - alpha (W Z^T q)_f (ypart is complete: two barriers since it was written); |y_f|^2 is this
    // frame's share of the coarse part of r^T z = |W Z^T r|^2
    if (tid < kCB) {
      double sy = 0.0;
      for (int w = 0; w < nWaves; ++w) sy += ypart[w * kCB + tid];
      const double yn = cs.y[f * kCB + tid] - alpha * sy;
      cs.y[f * kCB + tid] = yn;
      ypart[16 * kCB + tid] = yn * yn;
    }
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < kCB; ++k) t += ypart[16 * kCB + k];
      publishPartial(cs.fdotY + f, t);
    }
  }
  if (fusedDense && ds.rowSplit < kCB) {
    // ---- this frame's rows [rowSplit, 8) of the dense coarse level (DenseStep::rowSplit), in an LDS region of their own
    double* qcs = sm + ds.ldsPsum;
    double* psum = qcs + nC;
    if constexpr (FUSED) {
      for (int i = tid; i < static_cast<int>(nC); i += nThreads) qcs[i] = readPartial(ds.qc + i);  // (published by the finish halves)
    } else {
      for (int i = tid; i < static_cast<int>(nC); i += nThreads) qcs[i] = ds.qc[i];
    }
    const bool on0 = *ds.fail == 0;
    __syncthreads();
    double2 w[kDenseLoads];
    denseRows(f, ds.rowSplit, kCB - ds.rowSplit, qcs, psum, on0, w, false, ds.dotPart2 + f);
  }
  }  // frame workgroups
  // last workgroup: rz_new = sum, beta = rz_new / rz_old (device-side scalars, no host round trip)
  if (lastBlockArrivesLite(counter, gridDim.x, reinterpret_cast<int*>(red + 40))) {
    double a = 0.0, b = 0.0, cY = 0.0;
    for (int k = f0 + tid; k < f0 + nF; k += nThreads) {
      a += readPartial(fdotRZ + k);
      b += readPartial(fdotRR + k);
      if (fusedY) cY += readPartial(cs.fdotY + k);
      if (fusedDense && ds.rowSplit > 0) cY += readPartial(ds.dotPart + k);
      if (fusedDense && ds.rowSplit < kCB) cY += readPartial(ds.dotPart2 + k);
    }
    double cT = 0.0;  // third level's part of r^T z (kept apart: a broken-down sparse factor switches ITS level off below, not this one)
    if (tlOn)
      for (int k = tid; k < tsp->S * tsp->parts; k += nThreads) cT += readPartial(tsp->dotPart + k);
    if (tpOn)
      for (int k = tid; k < tpp->S * tpp->parts; k += nThreads) cT += readPartial(tpp->dotPart + k);
    a = waveSum(a);
    b = waveSum(b);
    cY = waveSum(cY);
    cT = waveSum(cT);
    __syncthreads();
    if ((tid & 63) == 0) { red[tid >> 6] = a; red[16 + (tid >> 6)] = b; ypart[tid >> 6] = cY; ypart[16 + (tid >> 6)] = cT; }
    __syncthreads();
    if (tid == 0) {
      double rzs = 0.0, rrs = 0.0, ys = 0.0, tls = 0.0;
      for (int w = 0; w < nWaves; ++w) { rzs += red[w]; rrs += red[16 + w]; ys += ypart[w]; tls += ypart[16 + w]; }
      // (owner-sharded iteration: every rank walks ALL rows of the third level -- they are few -- so its part of r^T z counts once,
      // on the first rank)
      rzs += (partOut == nullptr || f0 == 0) ? tls : 0.0;
      if (partOut != nullptr) {  // this rank's shares, summed over the ranks by k_pcg_scalars_dist
        partOut[0] = rzs + ((fusedY && *cs.fail != 0) ? 0.0 : ys);
        partOut[1] = rrs;
      } else if (fusedY) {  // two-level r^T z; a broken-down coarse factorisation switches the level off (consumers use c = 0)
        pcgFinishScalars(scal, 0, rzs + (*cs.fail == 0 ? ys : 0.0), rrs, tol2, hostMirror);
      } else if (fusedDense) {  // (a failed inverse left c = 0 and zero shares)
        pcgFinishScalars(scal, 0, rzs + ys, rrs, tol2, hostMirror);
      } else if (rc != nullptr || (init && (tsp != nullptr || tpp != nullptr))) {  // first residual: the coarse levels' kernels (k_coarse_apply_w /
                                                               // k_coarse_dense_apply / k_tl_rows_init) add their parts of r^T z and finish the scalars
        scal[S_RZPART] = rzs;
        scal[S_RR] = rrs;
      } else {
        pcgFinishScalars(scal, init, rzs, rrs, tol2, hostMirror);
      }
    }
  }
}

inline __global__ __launch_bounds__(1024) void k_cg_update(Layout L, int init, const double* __restrict__ g,
                                                    const float* __restrict__ minv, const double* __restrict__ p,
                                                    const double* __restrict__ q, double* __restrict__ scal,
                                                    unsigned int* __restrict__ counter, double* __restrict__ dx,
                                                    double* __restrict__ r, double* __restrict__ z,
                                                    double* __restrict__ fdotRZ, double* __restrict__ fdotRR,
                                                    double tol2, double* __restrict__ rc,
                                                    const unsigned char* __restrict__ modeActive,
                                                    double* __restrict__ hostMirror, CoarseStep cs, DenseStep ds,
                                                    const double* __restrict__ pqReduced, int f0, int nF,
                                                    double* __restrict__ partOut, const TlStep* __restrict__ tsp,
                                                    const TlStep* __restrict__ tpp) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  NoMid mid;
  cgUpdateBody<false>(L, init, g, minv, p, q, scal, counter, dx, r, z, fdotRZ, fdotRR, tol2, rc, modeActive, hostMirror, cs, ds,
                      pqReduced, sm, mid, f0, nF, partOut, tsp, tpp);
}

// Owner-sharded PCG iteration (multi-GPU): every rank updated x, r, z of ITS frames and left {r^T z, r^T r} of those frames in
// parts[2 rank]; after the all-gather of the shares one workgroup finishes the iteration's scalars (beta, convergence flag,
// iteration count, host mirror) on every rank -- identically: the ranks sum the same numbers in the same order.
inline __global__ void k_pcg_scalars_dist(const double* __restrict__ parts, int world, double* __restrict__ scal, double tol2,
                                          double* __restrict__ hostMirror) {
  if (threadIdx.x != 0 || scal[S_DONE] != 0.0) return;  // (iterations enqueued past convergence are no-ops)
  double rz = 0.0, rr = 0.0;
  for (int r = 0; r < world; ++r) {
    rz += parts[2 * r];
    rr += parts[2 * r + 1];
  }
  pcgFinishScalars(scal, 0, rz, rr, tol2, hostMirror);
}

// ---- one kernel for the tail of a PCG iteration ------------------------------------------------------
// k_matvec_finish and k_cg_update were two launches, two prologues and two last-workgroup hand-offs per iteration for
// F (+ F / 2 dense-level) workgroups that are all co-resident, each a chain of dependent round trips.  k_pcg_tail runs both
// halves in ONE launch: every workgroup first requests what its update half needs and nothing of this launch produces (its
// share of the f32 block M_f^-1 resp. of the dense coarse inverse, r, dx), the frame workgroups then run the finish half
// (partial rows -> q_f, p.q share, Z^T q published), ONE grid barrier, every workgroup sums the F shares of p.q itself
// (alpha), and the update half follows with its operands already in registers.  The last workgroup to leave publishes beta /
// the convergence flag exactly as k_cg_update does.  Single GPU, B <= 256, dense coarse level or none (the sparse level's
// column products cross workgroups through plain stores; a sharded run has a collective between the halves).
//
// Grid barrier (tailArrive / tailWait): self-resetting (the last arriver zeroes the count and bumps the generation), one use per launch.  Payload crossing it is published with agent-scope stores and read with
// agent-scope loads (publishPartial / readPartial, cdna_hip_programming.md G16 R1), every wave drains its stores first.
// The spin is bounded: a launch whose workgroups are not all resident abandons (the host checks the occupancy and serialises
// gated kernels of different handles, so this is a safety net) and the host's progress check reports the stalled PCG.
// The barrier carries the one reduction the halves need: the LAST workgroup to arrive sums the nParts published partials
// (p.q shares of the frames), publishes the sum and only then releases the others, which read that one double.
// Split in two: tailArrive right after the finish half (the workgroup's payload is published; nothing it requests afterwards
// delays its arrival), tailWait after the update half's operand requests.  Waiting costs MEMORY TRAFFIC: every poll is an
// L2-bypassing load, and hundreds of workgroups polling one address saturate the memory channel that holds it -- the finish
// halves still running slowed down 2x (measured with stamps, tools/tail_profile.py: 12.3 -> 7.6 us median when the poll
// interval went from 4 to 32 sleep units).  The release word is therefore replicated over kTailBarCopies pages (different
// channels), each workgroup polls the copy blockIdx % kTailBarCopies at a long interval.
// bar: [0] arrivals, [1] abandon flag, generation copies at bar[kTailBarStride * (1 + k)].
// scratch: 24 doubles of LDS ([0..15] wave partials, [16] role, [17] generation, [18] the sum).
constexpr int kTailBarCopies = 16;
constexpr int kTailBarStride = 1024;  // unsigned ints: 4 KB
// (thread 0 reads its generation copy with tailGeneration at the START of the kernel and keeps the arrival ticket in a register
// until tailWait: neither round trip sits between the finish half and the operand requests)
__device__ __forceinline__ unsigned int tailGeneration(const unsigned int* bar) {
  if (threadIdx.x != 0) return 0u;
  return *reinterpret_cast<const volatile unsigned int*>(bar + kTailBarStride * (1 + (blockIdx.x & (kTailBarCopies - 1))));
}
__device__ __forceinline__ unsigned int tailArrive(unsigned int* bar) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its published payload has landed
  __syncthreads();
  if (threadIdx.x != 0) return 0u;
  return __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The release word of a copy is a 16-byte RECORD {generation, sum (2 words), generation}, written with ONE 16-byte store and read
// with one 16-byte load: the sum travels with the flag, so the last arriver neither publishes it separately nor drains that store
// before releasing (2.5 us of the 7 between the last arrival and the release).  A reader accepts a record whose two generation
// words agree (a 16-byte access of one lane is one transaction; the second word guards against a torn one anyway).
__device__ __forceinline__ bool tailWait(unsigned int* bar, unsigned int nGroups, unsigned int gen, unsigned int ticket,
                                         const double* __restrict__ parts, int nParts, double* __restrict__ scratch, double& sum) {
  int* role = reinterpret_cast<int*>(scratch + 16);
  if (threadIdx.x == 0) {
    *role = (ticket == nGroups - 1) ? 2 : 1;
    *reinterpret_cast<unsigned int*>(scratch + 17) = gen;
  }
  __syncthreads();
  gen = *reinterpret_cast<unsigned int*>(scratch + 17);
  if (*role == 2) {
    const double v = blockSumPartials(parts, nParts, scratch);
    if (threadIdx.x == 0) __hip_atomic_store(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < kTailBarCopies) {
      cvd_u32x4 rec;
      rec.x = gen + 1u;
      rec.y = static_cast<unsigned int>(__double2loint(v));
      rec.z = static_cast<unsigned int>(__double2hiint(v));
      rec.w = gen + 1u;
      *reinterpret_cast<volatile cvd_u32x4*>(bar + kTailBarStride * (1 + threadIdx.x)) = rec;
    }
    sum = v;
    return true;
  }
  if (threadIdx.x == 0) {
    const volatile cvd_u32x4* mine = reinterpret_cast<const volatile cvd_u32x4*>(bar + kTailBarStride * (1 + (blockIdx.x & (kTailBarCopies - 1))));
    int ok = 1;
    unsigned int spins = 0;
    cvd_u32x4 rec;
    for (;;) {
      rec = *mine;
      if (rec.x == gen + 1u && rec.w == gen + 1u) break;
      __builtin_amdgcn_s_sleep(16);
      ++spins;
      if ((spins & 63u) == 0 && __hip_atomic_load(bar + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
      if (spins > (1u << 18)) {  // ~1 s
        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        //  ... and poison the arrival count: no later arriver of this solve can draw the LAST ticket any more, so an
        // abandoned barrier is never released behind the workgroups that left it (x / r stay consistent; the host's progress
        // check reports the stall and falls back to the two-launch tail)
        __hip_atomic_fetch_add(bar, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = 0;
        break;
      }
    }
    scratch[18] = ok ? __hiloint2double(static_cast<int>(rec.z), static_cast<int>(rec.y)) : 0.0;
    *role = ok;
  }
  __syncthreads();
  sum = scratch[18];
  return *role != 0;
}

// what the update half needs beside the finish half's arguments
struct TailUpdate {
  const float* minv;
  double* dx;
  double* r;
  double* z;
  double* fdotRZ;
  double* fdotRR;
  double tol2;
  const unsigned char* modeActive;
  double* hostMirror;
  unsigned int* counter;   // last-workgroup ticket of the update half
  unsigned int* gridBar;   // kTailBarStride * (1 + kTailBarCopies) words, zeroed by the host before a PCG solve
  double* pqSlot;          // where the barrier's last arriver publishes sum p.q (a line of its own)
  int ldsFinish;           // offset (doubles) of the finish half's LDS region
  int ldsScratch;          // offset of 24 doubles for the barrier flag and the p.q sum (beyond both kinds of workgroups' regions)
  DenseStep ds;
  const TlStep* ts;        // third level (device copy of its descriptor), nullptr: off
  const TlStep* tp;        // temporal pose level (coarse_level 3), nullptr: off
  int diagInProduct;       // explicit cross blocks: H_ff p_f is one of the frame's partial rows (k_cross_matvec), not the finish half's work
};

template <int KD>
inline __global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void k_pcg_tail(Layout L, const double* __restrict__ x,
                                                    const double* __restrict__ mask,
                                                    const double* __restrict__ lam, const float* __restrict__ median,
                                                    const unsigned char* __restrict__ inRange,
                                                    const unsigned char* __restrict__ rangeFlags,
                                                    const int* __restrict__ fiOff, const int* __restrict__ fiList,
                                                    const double* __restrict__ qPart, const double* __restrict__ pOld,
                                                    double* __restrict__ pNew, double* __restrict__ scal, int useBeta,
                                                    double* __restrict__ q, double* __restrict__ fdot, int nRows, RegCache rc,
                                                    CoarseView V, double* __restrict__ qc, const double* __restrict__ Hdiag,
                                                    TailUpdate U) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int f = blockIdx.x;
  const CoarseColumns ccOff{nullptr, nullptr, nullptr, nullptr, nullptr};
  const CoarseStep csOff{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  TAIL_STAMP(0);
  const unsigned int barGen = tailGeneration(U.gridBar);
  unsigned int barTicket = 0u;
  auto first = [&](double& pv, double& qv) -> bool {
    if (f < L.F) {
      TailCarry carry{0.0, 0.0};
      if (!matvecFinishBody<KD, true>(L, x, mask, lam, median, inRange, rangeFlags, fiOff, fiList, qPart, U.z, pOld, pNew, scal,
                                      nullptr, useBeta, q, fdot, 0, nRows, rc, V, qc, ccOff, Hdiag, nullptr, 0, U.diagInProduct ? 0 : L.F,
                                      sm + U.ldsFinish, carry, U.ts, sm + L.B))  // (restriction products: the update half's partial-sum region)
        return false;
      pv = carry.pv;
      qv = carry.qv;
    } else if (scal[S_DONE] != 0.0) {
      return false;  // (dense-level workgroup of a launch enqueued past convergence)
    }
    TAIL_STAMP(1);
    barTicket = tailArrive(U.gridBar);
    return true;
  };
  auto second = [&](double& alpha) -> bool {
    double pq;
    TAIL_STAMP(2);
    if (!tailWait(U.gridBar, gridDim.x, barGen, barTicket, fdot, L.F, sm + U.ldsScratch, pq)) return false;
    TAIL_STAMP(3);
    // (S_RZ is rewritten only by the last workgroup to take the update half's ticket, i.e. after every workgroup has read it)
    alpha = scal[S_RZ] / pq;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      scal[S_PQ] = pq;
      scal[S_ALPHA] = alpha;
    }
    return true;
  };
  struct Mid {
    decltype(first)& f1;
    decltype(second)& f2;
    double* scratch;
    __device__ bool first(double& pv, double& qv) { return f1(pv, qv); }
    __device__ bool second(double& alpha) { return f2(alpha); }
    // (the barrier's generation at the start of this launch: tailWait left it in the scratch words for every thread)
    __device__ unsigned int generation() { return *reinterpret_cast<unsigned int*>(scratch + 17); }
  } mid{first, second, sm + U.ldsScratch};
  cgUpdateBody<true>(L, 0, nullptr, U.minv, nullptr, nullptr, scal, U.counter, U.dx, U.r, U.z, U.fdotRZ, U.fdotRR, U.tol2, nullptr,
                     U.modeActive, U.hostMirror, csOff, U.ds, nullptr, sm, mid, 0, L.F, nullptr, U.ts, U.tp);
  TAIL_STAMP(4);
}

// Step statistics (one block): d.g, d.r, d.(lam d), |d|^2, |x|^2 (active unknowns), max |g|, and the number of active unknowns
// (entries of diag(H) that are not zero: what the host used to count from a downloaded copy at the start of every solve).
inline __global__ __launch_bounds__(256) void k_step_stats(size_t n, const double* __restrict__ dx,
                                                    const double* __restrict__ g, const double* __restrict__ r,
                                                    const double* __restrict__ lam, const double* __restrict__ x,
                                                    const double* __restrict__ hdiagActive, double* __restrict__ scal,
                                                    double* __restrict__ part, unsigned int* __restrict__ counter) {
  // gridDim.x workgroups stride over the vector; the last one to arrive folds the per-workgroup partials
  constexpr int NQ = 7;   // sums 0..4 and 6, maximum 5
  __shared__ double red[NQ][4];
  __shared__ int flag;
  const int G = gridDim.x;
  double a[NQ] = {0, 0, 0, 0, 0, 0, 0};
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<size_t>(G) * 256) {
    const double d = dx[i];
    a[0] += d * g[i];
    a[1] += d * r[i];
    a[2] += d * lam[i] * d;
    a[3] += d * d;
    if (hdiagActive[i] != 0.0) { a[4] += x[i] * x[i]; a[6] += 1.0; }
    a[5] = fmax(a[5], fabs(g[i]));
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k)
    if (k != 5) a[k] = waveSum(a[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a[5] = fmax(a[5], __shfl_xor(a[5], off, 64));
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < NQ; ++k) red[k][threadIdx.x >> 6] = a[k];
  __syncthreads();
  if (threadIdx.x < NQ) {
    const int k = threadIdx.x;
    part[k * G + blockIdx.x] = (k != 5) ? (red[k][0] + red[k][1]) + (red[k][2] + red[k][3])
                                        : fmax(fmax(red[5][0], red[5][1]), fmax(red[5][2], red[5][3]));
  }
  if (!lastBlockArrives(counter, G, &flag)) return;
  double t[NQ] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < G; b += 256) {
#pragma unroll
    for (int k = 0; k < NQ; ++k)
      if (k != 5) t[k] += part[k * G + b];
    t[5] = fmax(t[5], part[5 * G + b]);
  }
#pragma unroll
  for (int k = 0; k < NQ; ++k)
    if (k != 5) t[k] = waveSum(t[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t[5] = fmax(t[5], __shfl_xor(t[5], off, 64));
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < NQ; ++k) red[k][threadIdx.x >> 6] = t[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    scal[S_DG] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    scal[S_DR] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    scal[S_DLD] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    scal[S_DD] = red[3][0] + red[3][1] + red[3][2] + red[3][3];
    scal[S_XX] = red[4][0] + red[4][1] + red[4][2] + red[4][3];
    scal[S_GMAX] = fmax(fmax(red[5][0], red[5][1]), fmax(red[5][2], red[5][3]));
    scal[S_NACTIVE] = red[6][0] + red[6][1] + red[6][2] + red[6][3];
  }
}

// xcand = x + dx with the lower bound 0 on theta_k[0] when requested (ParameterBlock::Plus projection).
inline __global__ void k_apply_step(Layout L, int boundDepth0, const double* __restrict__ x, const double* __restrict__ dx,
                             double* __restrict__ xc) {
  const size_t n = static_cast<size_t>(L.F) * L.B;
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i] + dx[i];
  if (boundDepth0) {
    const int c = static_cast<int>(i % L.B);
    if (c >= 7 && c < 7 + L.nD && ((c - 7) % L.N) == 0) v = fmax(v, 0.0);
  }
  xc[i] = v;
}

inline __global__ void k_extract_diag(Layout L, const double* __restrict__ hBlocks, double* __restrict__ out) {
  const size_t n = static_cast<size_t>(L.F) * L.B;
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t f = i / L.B, c = i - f * L.B;
  out[i] = hBlocks[(f * L.B + c) * L.B + c];
}


// =====================================================================================================
// Fast path of the hot kernel (identity spatial transform, reprojection losses, Identity / Global / bilinear
// depth transform = everything the reference's default pipeline uses).
//
// Same operator as k_matvec_pairs, restructured so that no 3x7 Jacobian is ever materialised:
//   forward  : dX = directional derivative of the world point along p_a, dq = R_b^T (dX - p_tb) + E_b^T v,
//              t = rho' * (d r / d q . dq + direct terms)            with E_f = sum_i p_w,i dR_f,i per frame
//   backward : y_q = (d r / d q)^T t, y_X = R_b y_q; translations get +-y_X, the rotation / focal / depth
//              columns are contracted per WORKGROUP from 3x3 outer-product accumulators
//              O_a = sum D_a y_X c_a^T, O_b = sum v y_q^T  (q_w,i = <dR_i, O>), so dR never enters the loop.
// Per-lane state: 23 accumulators; taps are fully unrolled (no scratch).
// =====================================================================================================
template <int KD>
struct FastTaps {
  int idx[KD];
  double w[KD];
  __device__ __forceinline__ bool ok(int) const { return true; }
  __device__ __forceinline__ int I(int k) const { return idx[k]; }
  __device__ __forceinline__ double Wt(int k) const { return w[k]; }
};

// bicubic: the 16 taps stay in separable form (8 weights + 3 ints instead of 16 + 16 registers); tap k = (k & 3, k >> 2)
// exists when it lies inside the folded xs x ys footprint, in the same row-major order as bicubicTaps enumerates.
template <>
struct FastTaps<16> {
  CubicSep s;
  int gx;
  __device__ __forceinline__ bool ok(int k) const { return (k & 3) < s.xs && (k >> 2) < s.ys; }
  __device__ __forceinline__ int I(int k) const { return s.base + (k & 3) + (k >> 2) * gx; }
  __device__ __forceinline__ double Wt(int k) const { return s.fx[k & 3] * s.fy[k >> 2]; }
};

// gridCell (cvd_device.h) in 6 instructions instead of 9, bit-identical: (loc + 1) is exact in double, so one FMA rounds the same
// real number as the product (loc + 1) * ((g - 1) / 2) does ((g - 1) / 2 is a half-integer: exact), and for the clamped s >= 0
// v_fract_f64 is s - trunc(s).  For the hot product's loop (four cells per constraint).
__device__ __forceinline__ void gridCellFast(float loc, double hg, double maxc, int& i, double& r) {
  double s = __builtin_fma(static_cast<double>(loc), hg, hg);
  s = fmin(fmax(s, 0.0), maxc);
  i = static_cast<int>(s);
  r = __builtin_amdgcn_fract(s);
}
template <int KD>
__device__ __forceinline__ void fastGather(const Layout& L, float lx, float ly, FastTaps<KD>& t) {
  if constexpr (KD == 16) {
    bicubicSeparable(lx, ly, L.gx, L.gy, L.maxcx, L.maxcy, t.s);
    t.gx = L.gx;
  } else if constexpr (KD == 4) {
    int ix, iy;
    double rx, ry;
    gridCellFast(lx, 0.5 * static_cast<double>(L.gx - 1), L.maxcx, ix, rx);
    gridCellFast(ly, 0.5 * static_cast<double>(L.gy - 1), L.maxcy, iy, ry);
    const int i0 = ix + iy * L.gx;
    t.idx[0] = i0;            t.w[0] = (1.0 - rx) * (1.0 - ry);
    t.idx[1] = i0 + 1;        t.w[1] = rx * (1.0 - ry);
    t.idx[2] = i0 + L.gx;     t.w[2] = (1.0 - rx) * ry;
    t.idx[3] = i0 + L.gx + 1; t.w[3] = rx * ry;
  } else {
    t.idx[0] = 0;
    t.w[0] = 1.0;
  }
}

// Candidate-point cost on the fast path (same scope as k_matvec_pairs_fast: identity spatial transform, reprojection
// losses): the residual chain of the fast kernels with register-resident taps.  The generic k_cost_items keeps the taps
// of Sample<KD, KS> in dynamically indexed arrays, i.e. in scratch memory (672 B per lane, stores and dependent reloads
// per constraint): 53 us for 1.09 M constraints where the arithmetic needs ~10.
template <int KD, bool DENSE = false>
inline __global__ __launch_bounds__(256) void k_cost_items_fast(Layout L, Table T, Items it, const double* __restrict__ x,
                                                         const FrameConst* __restrict__ fc, double* __restrict__ costItem) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr double eps = 1e-6;
  const int B = L.B;
  double* xa = sm;
  double* xb = sm + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(sm + 2 * B);
  double* red = reinterpret_cast<double*>(fcs + 2);
  const int item = blockIdx.x;
  const int tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  for (int i = tid; i < B; i += 256) {
    xa[i] = x[static_cast<size_t>(fa) * B + i];
    xb[i] = x[static_cast<size_t>(fb) * B + i];
  }
  constexpr int FCW = sizeof(FrameConst) / 8;
  if (tid < 2 * FCW) {
    const int which = tid / FCW, k = tid % FCW;
    reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
  }
  __syncthreads();
  const int N = L.N;
  const double A = L.aspect;
  double acc = 0.0;
  for (int dir = 0; dir < 2; ++dir) {
    const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
    const FrameConst& Fa = fcs[dir];
    const FrameConst& Fb = fcs[dir ^ 1];
    const double* xs = dir ? xb : xa;
    const double* xt = dir ? xa : xb;
    const double fya = Fa.fy, fxa = Fa.fy * A;
    const double fyb = Fb.fy;
    const double ifyb = 1.0 / fyb, ifxb = 1.0 / (fyb * A);
    const int fsrc = dir ? fb : fa, ftgt = dir ? fa : fb;
    const long long pixBase = DENSE ? (cb / (static_cast<long long>(T.W) * T.H)) * (static_cast<long long>(T.W) * T.H) : 0;
    std::conditional_t<DENSE, DenseStreamAhead, RecordStream<false>> rs;
    const int nDir = static_cast<int>(ce - cb);
    if constexpr (DENSE) rs.prime(T, cb, tid, 256, nDir, pixBase, fsrc, ftgt);
    else rs.prime(T, cb, tid, nDir);
    for (int ci = tid; ci < nDir; ci += 256) {
      float4 nd;
      float2 d;
      if (!rs.take(T, cb, ci, 256, nDir, pixBase, fsrc, ftgt, nd, d)) continue;
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
            const double wa = ta.Wt(k);
            Da += (N == 2 ? da * xs[7 + ia * 2] + xs[7 + ia * 2 + 1] : da * xs[7 + ia]) * wa;
          }
          if (tb.ok(k)) {
            const int ib = tb.I(k);
            const double wb = tb.Wt(k);
            Db += (N == 2 ? db * xt[7 + ib * 2] + xt[7 + ib * 2 + 1] : db * xt[7 + ib]) * wb;
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
      const double r0 = (q0 * iz * ifxb - pbx) * L.ws;
      const double r1 = (q1 * iz * ifyb - pby) * L.ws;
      double r2;
      if (L.lossType == kLossDisparity) {
        const double zc = !(zz < eps) ? zz : eps, bc = !(Db < eps) ? Db : eps;
        r2 = (1.0 / zc - 1.0 / bc) * L.wd;
      } else {
        const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
        const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
        r2 = (L.lossType == kLossRatio ? (mx / mn - 1.0) : log(mn / mx)) * L.wd;
      }
      double rho0, rho1;
      robustRho(L, r0 * r0 + r1 * r1 + r2 * r2, rho0, rho1);
      acc += rho0;
    }
  }
  acc = waveSum(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) costItem[item] = 0.5 * ((red[0] + red[1]) + (red[2] + red[3]));
}

// 1 / x to f64 accuracy (1-2 ulp; not correctly rounded): hardware estimate + two Newton steps, 5 instructions instead of the
// ~10 of the IEEE division sequence (div_scale x2, rcp, fma chain, div_fmas, div_fixup).  For the hot product's loop only.
__device__ __forceinline__ double rcpFast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}

// 1 / sqrt(x) the same way (hardware estimate + two Newton steps): the Huber weight of the specialised product.
__device__ __forceinline__ double rsqrtFast(double x) {
  double r = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = __builtin_fma(-h * r, r, 0.5);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-h * r, r, 0.5);
  return __builtin_fma(r, e, r);
}

// A wave-uniform double as a scalar (SGPR pair): the compiler cannot prove that an LDS load is uniform.
__device__ __forceinline__ double uniformValue(double v) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

#ifdef CVD_MV_PROFILE  // tools/mv_profile.py: wall-clock (100 MHz) stamps of the hot product's workgroups
__device__ unsigned long long g_mvProf[4096 * 8];
#define MV_STAMP(slot) do { if ((threadIdx.x & 63) == 0) g_mvProf[(blockIdx.x & 4095) * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define MV_STAMP(slot) do {} while (0)
#endif
constexpr int kRedVals = 15;               // accumulators of k_matvec_pairs_fast reduced per workgroup (11 + the 4 depth-block sums of KD = 1)
constexpr int kRedStride = 4 * 33 + 1;     // 128 columns (lane pairs pre-summed) in 33-padded segments of 32, +1 skew

// SPEC = 2: the same with the Huber robustifier (round 5: configs[4]'s Huber variant ran the generic-loss kernel at 253 us
// per product against 149 us for Cauchy).
// SPEC = 1: the default pipeline's variant fixed at compile time (one value parameter per vertex, ReproDisparity loss,
// Cauchy robustifier): the branches on the runtime Layout fields drop out of the constraint loop.
#ifndef CVD_MV_WAVES
#define CVD_MV_WAVES 4   // waves per SIMD the specialised list-mode product (Global / bilinear) is compiled for (round 5: 129 -> 128
                         // VGPRs; 3 = rounds 2-4 and still the bicubic and dense variants, which spill at 128)
#endif
#ifndef CVD_MV_SLOAD
#define CVD_MV_SLOAD 1   // 1: frame constants by scalar loads from global memory; 0: staged in LDS, v_readfirstlane per direction
                         // (same box, 4140-pair set: 39.6 us against 41.4 / 41.9; without the early request below 40.8)
#endif
#ifndef CVD_MV_EARLY
#define CVD_MV_EARLY 1   // 1: direction 0's first table record is requested at the top of the kernel
#endif
template <int KD, int NT, int SPEC = 0, bool DENSE = false>
inline __global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(SPEC ? ((KD <= 4 && !DENSE) ? CVD_MV_WAVES : 3) : 2))) void k_matvec_pairs_fast(Layout L, Table T, Items it, const double* __restrict__ x,
                                                           const FrameConst* __restrict__ fc,
                                                           const double* __restrict__ mask,
                                                           const double* __restrict__ z, const double* __restrict__ pOld,
                                                           const double* __restrict__ scal, int useBeta,
                                                           double* __restrict__ qPart, CoarseView V) {
  // (PCG already converged: iterations enqueued ahead return -- tested after the prologue's loads have been issued, so
  // that the flag does not cost a dependent global round trip of its own in every working launch)
  const double sDone = scal[S_DONE];
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr int NV = KD == 1 ? kRedVals : 11;  // the depth-block sums exist only with one tap per side
  constexpr double eps = 1e-6;
  const int B = L.B;
  if (threadIdx.x == 0) MV_STAMP(0);
  double* xa = sm;
  double* xb = xa + B;
  double* pa = xb + B;
  double* pb = pa + B;
  double* qa = pb + B;
  double* qb = qa + B;
  FrameConst* fcs = reinterpret_cast<FrameConst*>(qb + B);   // (CVD_MV_SLOAD = 0: both frames' constants)
  double* red = reinterpret_cast<double*>(fcs + 2);          // the reduced accumulators
  double* W = red + 32;                            // transposed reduction scratch: kRedVals rows x kRedStride
  // Dense mode: neighbouring lanes are neighbouring pixels and hit the SAME grid vertices -- 64-way same-address LDS
  // atomics.  The grid columns are therefore accumulated into kPriv lane-keyed private copies (after W) and folded
  // before the epilogue.
  constexpr int kPriv = DENSE ? 8 : 1;   // (list mode: 2 lane-keyed copies 52.6 -> 51.4 us in raster order, nothing once the table is cell-ordered)
  constexpr bool kUsePriv = kPriv > 1;
  double* qpriv = W + static_cast<size_t>(kRedVals) * kRedStride;  // DENSE: [kPriv][2][B]
  double* cl = W;                                  // prologue only: 2 x kCB coarse corrections
  const int item = blockIdx.x;
  const int tid = threadIdx.x;
  const int fa = it.fa[item], fb = it.fb[item];
  const double beta = useBeta ? scal[S_BETA] : 0.0;
  // The frame constants (R, t, fy, J_l of both frames) are wave-uniform READ-ONLY global data at an address that depends on
  // blockIdx only: the compiler fetches them with scalar loads straight into SGPRs (round 5; rounds 2-4 staged the two structs in
  // LDS and moved 26 doubles per direction into SGPRs through v_readfirstlane: ~200 instructions per wave and direction).
#if CVD_MV_SLOAD
  const FrameConst* __restrict__ fcF[2] = {fc + fa, fc + fb};
#else
  const FrameConst* fcF[2] = {fcs, fcs + 1};
  {
    constexpr int FCW = sizeof(FrameConst) / 8;
    if (tid >= NT - 2 * FCW) {  // (the last waves: the first ones carry the coarse loads)
      const int t = tid - (NT - 2 * FCW);
      const int which = t / FCW, k = t % FCW;
      reinterpret_cast<double*>(fcs + which)[k] = reinterpret_cast<const double*>(fc + (which ? fb : fa))[k];
    }
  }
#endif
  // The first table record of direction 0 is requested before anything else: it is back when the prologue's barriers are.
  const long long cb0 = it.range[item * 4], ce0 = it.range[item * 4 + 1];
  // (not the bicubic variant: six more live registers through the prologue spill at its budget)
  constexpr bool kEarlyPrime = KD <= 4 && CVD_MV_EARLY;
  RecordStream<DENSE> rs;   // (list mode: the next trip's table record is in flight during this trip's arithmetic)
  if constexpr (kEarlyPrime) rs.prime(T, cb0, (tid >> 6) * 64 + (tid & 63), static_cast<int>(ce0 - cb0));
  // Prologue in ONE global round trip (the workgroup lives ~5 us, a dependent load costs ~1 us of it): every thread
  // issues its loads of x / z / p_old / mask for both frames and the coarse correction c_f (see CoarseView) in the same phase,
  // and the search direction is formed after the barrier.  B <= 256: one element per thread.
  if (V.Wb != nullptr) {  // (fused coarse variant: the correction is gathered here, one wave per frame)
    if (tid < 64) coarseFrameCorrection(V, fa, tid, cl);
    if (NT == 64) coarseFrameCorrection(V, fb, tid, cl + kCB);   // (one-wave workgroups of the deterministic build)
    else if (tid >= 64 && tid < 128) coarseFrameCorrection(V, fb, tid - 64, cl + kCB);
  } else if (tid < 2 * kCB) {
    cl[tid] = (V.cF != nullptr) ? V.cF[(tid < kCB ? fa : fb) * kCB + (tid & (kCB - 1))] : 0.0;
  }
  constexpr int EPT = 256 / NT;  // elements of a frame block per thread (B <= 256)
  double vza[EPT], vzb[EPT], vpa[EPT], vpb[EPT], vma[EPT], vmb[EPT];
  // third level (CoarseView::tl): the two frames' coefficients go to LDS behind the coarse corrections, the thread's vertex
  // taps to registers -- same round trip as everything else here
  const bool tlOn = V.tl != nullptr;
  double* tls = cl + 2 * kCB;
  TlTaps tt[EPT];
  if (tlOn) {
    for (int i = tid; i < 2 * V.tlS; i += NT) {
      const int which = i >= V.tlS ? 1 : 0;
      tls[i] = V.tl[static_cast<size_t>(which ? fb : fa) * V.tlS + (i - which * V.tlS)];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) tlLoadTaps(V, tid + e * NT, L.nD, tt[e]);
  }
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * NT;
    vza[e] = vzb[e] = vpa[e] = vpb[e] = vma[e] = vmb[e] = 0.0;
    if (i < B) {
      const size_t ia = static_cast<size_t>(fa) * B + i, ib = static_cast<size_t>(fb) * B + i;
      xa[i] = x[ia];
      xb[i] = x[ib];
      vza[e] = z[ia];
      vzb[e] = z[ib];
      if (useBeta) { vpa[e] = pOld[ia]; vpb[e] = pOld[ib]; }
      vma[e] = mask[ia];
      vmb[e] = mask[ib];
      qa[i] = 0.0;
      qb[i] = 0.0;
    }
  }
  if constexpr (kUsePriv)
    for (int i = tid; i < kPriv * 2 * B; i += NT) qpriv[i] = 0.0;
  __syncthreads();
  if (sDone != 0.0) return;  // uniform; nothing has been written to global memory yet
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * NT;
    if (i < B) {
      double ca = coarseAtLds(cl, L, i), cb2 = coarseAtLds(cl + kCB, L, i);
      if (tlOn) {
        ca += tlAt(tls, tt[e]);
        cb2 += tlAt(tls + V.tlS, tt[e]);
      }
      pa[i] = (vza[e] + ca + (useBeta ? beta * vpa[e] : 0.0)) * vma[e];
      pb[i] = (vzb[e] + cb2 + (useBeta ? beta * vpb[e] : 0.0)) * vmb[e];
    }
  }
  if (L.intrOpt == kIntrShared) {
    // every constraint's focal column is frame 0's slot (reference lib/PoseOptimizer.cpp:1226)
    __syncthreads();
    if (tid == 0) {
      const double p06 = (z[6] + (useBeta ? beta * pOld[6] : 0.0)) * mask[6];
      pa[6] = p06;
      pb[6] = p06;
    }
  }
  __syncthreads();
  // Rotation derivatives in CROSS-PRODUCT form (round 5).  R(w) = exp([w]x)  =>  dR/dw_i = [a_i]x R with a_i the i-th column of
  // the left Jacobian of SO(3) (FrameConst::Jl, read off dR_i R^T by k_frame_consts).  With e = sum_i p_w,i a_i
  //   forward   source: sum_i p_w,i dR_i c = e x (R c);   target: sum_i p_w,i dR_i^T v = R^T (v x e)
  //   adjoint   source: y . dR_i (D c)   = a_i . ((D R c) x y);   target: y_q . dR_i^T v = -a_i . (v x R y_q)
  // so a constraint contributes ONE cross product v x y_X to the rotation rows of both frames (3 accumulators and 6
  // instructions instead of the two 3 x 3 outer products: 18 accumulators, 21 instructions), the 18 E entries of a trip's LDS
  // broadcast reads become 6 scalars, and the workgroup reduction carries 11 values instead of 23.
  if (threadIdx.x == 0) MV_STAMP(1);

  const int N = SPEC ? 1 : L.N;
  const int lossType = SPEC ? static_cast<int>(kLossDisparity) : L.lossType;
  const double A = L.aspect;
  // The pose-level accumulators serve both directions through ONE reduction: aT = sum y_X of the running direction (the
  // translation adjoint: q_t,src = +aT, q_t,tgt = -aT; direction 0's sum is parked in aT0, the rotation rows need both),
  // Cx = sum v x y_X (negated between the directions: source and target swap), the focal sums per ROLE (swapped).
  double aT[3] = {0, 0, 0}, aT0[3] = {0, 0, 0};
  double Cx[3] = {0, 0, 0};
  double aFa = 0.0, aFb = 0.0;  // (unnormalised: x 1 / fy of the frame in the epilogue)
  double gDa[2] = {0.0, 0.0}, gDb[2] = {0.0, 0.0};  // KD == 1: every sample hits the one depth block -> registers
  const long long units0 = (it.range[item * 4 + 1] - it.range[item * 4] + 63) >> 6;  // wave-units of direction 0
  for (int dir = 0; dir < 2; ++dir) {
  const long long cb = it.range[item * 4 + dir * 2], ce = it.range[item * 4 + dir * 2 + 1];
  // role swap for the reverse pair (source = fb, target = fa): swap every per-frame pointer
  const FrameConst& Fa = *fcF[dir];
  const FrameConst& Fb = *fcF[dir ^ 1];
  if (dir) {
    double* t;
    t = xa; xa = xb; xb = t;
    t = pa; pa = pb; pb = t;
    t = qa; qa = qb; qb = t;
#pragma unroll
    for (int i = 0; i < 3; ++i) { aT0[i] = aT[i]; aT[i] = 0.0; Cx[i] = -Cx[i]; }
    { const double o = aFa; aFa = aFb; aFb = o; }
#pragma unroll
    for (int n = 0; n < 2; ++n) { const double o = gDa[n]; gDa[n] = gDb[n]; gDb[n] = o; }
  }
  const int firstUnit = dir == 0 ? (tid >> 6) : static_cast<int>(((tid >> 6) - units0) & (NT / 64 - 1));
  const int nDir = static_cast<int>(ce - cb);
  const int iFirst = firstUnit * 64 + (tid & 63);
  if (dir || !kEarlyPrime) rs.prime(T, cb, iFirst, nDir);   // (direction 0: requested at the top of the kernel; here: before the scalar set-up)
  // The frame constants are the same for every lane and reach the loop as SCALAR values (scalar loads; a VALU instruction
  // takes one scalar operand).  Rotations and translations of both frames are 24 doubles = 48 VGPRs less per lane.
  double RaU[9], RbU[9], dTU[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { RaU[i] = uniformValue(Fa.R[i]); RbU[i] = uniformValue(Fb.R[i]); }
#pragma unroll
  for (int i = 0; i < 3; ++i) dTU[i] = uniformValue(Fa.t[i] - Fb.t[i]);
  // (round 5: the rotation parts of the search direction are 3 + 3 scalars -- e = J_l^T p_w of both frames -- and its
  // translation / focal parts another 5; as 18 + 14 values they overflowed the SGPR file and were LDS broadcast reads inside
  // the loop.  Every wave forms them itself from the LDS copy of p: 18 multiply-adds instead of a barrier)
  double eaU[3], ebU[3], dptU[3];
  {
    const double pwa[3] = {uniformValue(pa[3]), uniformValue(pa[4]), uniformValue(pa[5])};
    const double pwb[3] = {uniformValue(pb[3]), uniformValue(pb[4]), uniformValue(pb[5])};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      eaU[k] = uniformValue(pwa[0] * Fa.Jl[k] + pwa[1] * Fa.Jl[3 + k] + pwa[2] * Fa.Jl[6 + k]);
      ebU[k] = uniformValue(pwb[0] * Fb.Jl[k] + pwb[1] * Fb.Jl[3 + k] + pwb[2] * Fb.Jl[6 + k]);
      dptU[k] = uniformValue(pa[k] - pb[k]);
    }
  }
  const double fya = uniformValue(Fa.fy), fxa = fya * A;
  const double fyb = uniformValue(Fb.fy);
  const double ifyb = uniformValue(1.0 / fyb), ifxb = uniformValue(1.0 / (fyb * A));
  const double pfaU = uniformValue(pa[6] / fya);          // p_f,src / fy_src: the focal column of the source is D R (c + e_z) / fy
  const double pfrU = uniformValue(L.ws * pb[6] * ifyb);   // ws p_f,tgt / fy_tgt

  const int fsrc = dir ? fb : fa, ftgt = dir ? fa : fb;
  const long long pixBase = DENSE ? (cb / (static_cast<long long>(T.W) * T.H)) * (static_cast<long long>(T.W) * T.H) : 0;
  // (dense mode keeps lane = pixel here: the private accumulator copies take care of the same-vertex collisions, and
  // the run-per-lane mapping of the assembly kernel measured slower for this kernel: 1.00 vs 0.75 ms)
  // Wave-units of 64 constraints are dealt round-robin over the waves ACROSS the two directions: direction 1 starts with
  // the wave after the one that took direction 0's last unit.  With ~9 units per direction and 4 waves the slowest wave
  // walks 5 units instead of 3 + 3 (both directions' remainders used to land on waves 0, 1).
  for (int ci = iFirst; ci < nDir; ci += NT) {
    float4 nd;
    float2 d;
    if (!rs.take(T, cb, ci, NT, nDir, pixBase, fsrc, ftgt, nd, d)) continue;
    const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
    FastTaps<KD> ta, tb;
    fastGather<KD>(L, nd.x, nd.y, ta);
    fastGather<KD>(L, nd.z, nd.w, tb);
    // depths and their directional derivatives
    double Da, Db, sDa, sDb;
    if (N == 0) {
      Da = da; Db = db; sDa = 0.0; sDb = 0.0;
    } else {
      Da = 0.0; Db = 0.0; sDa = 0.0; sDb = 0.0;
#pragma unroll
      for (int k = 0; k < KD; ++k) {
        if (ta.ok(k)) {
          const int ia = ta.I(k);
          const double wa = ta.Wt(k);
          if (N == 2) {
            Da += (da * xa[7 + ia * 2] + xa[7 + ia * 2 + 1]) * wa;
            sDa += (da * pa[7 + ia * 2] + pa[7 + ia * 2 + 1]) * wa;
          } else {  // (one value parameter: the source depth multiplies the interpolated scale once, below)
            Da += xa[7 + ia] * wa;
            sDa += pa[7 + ia] * wa;
          }
        }
        if (tb.ok(k)) {
          const int ib = tb.I(k);
          const double wb = tb.Wt(k);
          if (N == 2) {
            Db += (db * xb[7 + ib * 2] + xb[7 + ib * 2 + 1]) * wb;
            sDb += (db * pb[7 + ib * 2] + pb[7 + ib * 2 + 1]) * wb;
          } else {
            Db += xb[7 + ib] * wb;
            sDb += pb[7 + ib] * wb;
          }
        }
      }
      if (N != 2) { Da *= da; sDa *= da; Db *= db; sDb *= db; }
    }
    const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
    const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
    const double ca[3] = {pax * fxa, pay * fya, -1.0};
    const double Rca[3] = {dot3(RaU, ca), dot3(RaU + 3, ca), dot3(RaU + 6, ca)};
    const double v[3] = {dTU[0] + Rca[0] * Da, dTU[1] + Rca[1] * Da, dTU[2] + Rca[2] * Da};
    const double q0 = RbU[0] * v[0] + RbU[3] * v[1] + RbU[6] * v[2];
    const double q1 = RbU[1] * v[0] + RbU[4] * v[1] + RbU[7] * v[2];
    const double q2 = RbU[2] * v[0] + RbU[5] * v[1] + RbU[8] * v[2];
    const double zz = -q2;
    const double iz = SPEC ? rcpFast(zz) : 1.0 / zz;
    const double u = q0 * iz * ifxb;
    const double vv = q1 * iz * ifyb;
    const double r0 = (u - pbx) * L.ws;
    const double r1 = (vv - pby) * L.ws;
    double r2, dr2dA, dr2dDb;
    if (lossType == kLossDisparity) {
      const bool zo = !(zz < eps), bo = !(Db < eps);
      const double zc = zo ? zz : eps, bc = bo ? Db : eps;
      const double izc = zo ? iz : 1.0 / eps, ibc = SPEC ? rcpFast(bc) : 1.0 / bc;
      r2 = (izc - ibc) * L.wd;
      dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
      dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
    } else {
      const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
      const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
      if (lossType == kLossRatio) {
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
    const double sq = r0 * r0 + r1 * r1 + r2 * r2;
    // (SPEC 1: Cauchy, what the reference hard-wires; SPEC 2: Huber, BASELINE configs[4]'s stress variant -- rho' = a / sqrt(s)
    // beyond s = a^2, clamped like ceres::HuberLoss)
    const double rho1 = SPEC == 1 ? rcpFast(1.0 + sq * L.cauchyC)
                                  : (SPEC == 2 ? (sq > L.cauchyB ? fmax(2.2250738585072014e-308, L.robustA * rsqrtFast(sq)) : 1.0)
                                               : robustRho1(L, sq));

    // ---- forward: dX, dq, t
    // R c_f, c_f = d c_a / d fy = (c_a + e_z) / fy: the rotated ray plus R's third column (the 1 / fy sits in pfaU and, for the
    // adjoint, in the epilogue)
    const double Rcf[3] = {Rca[0] + RaU[2], Rca[1] + RaU[5], Rca[2] + RaU[8]};
    const double Eca[3] = {eaU[1] * Rca[2] - eaU[2] * Rca[1], eaU[2] * Rca[0] - eaU[0] * Rca[2], eaU[0] * Rca[1] - eaU[1] * Rca[0]};
    const double vxe[3] = {v[1] * ebU[2] - v[2] * ebU[1], v[2] * ebU[0] - v[0] * ebU[2], v[0] * ebU[1] - v[1] * ebU[0]};
    double w3[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w3[i] = dptU[i] + Da * (Eca[i] + pfaU * Rcf[i]) + sDa * Rca[i] + vxe[i];
    const double dq0 = RbU[0] * w3[0] + RbU[3] * w3[1] + RbU[6] * w3[2];
    const double dq1 = RbU[1] * w3[0] + RbU[4] * w3[1] + RbU[7] * w3[2];
    const double dq2 = RbU[2] * w3[0] + RbU[5] * w3[1] + RbU[8] * w3[2];
    const double wiz = L.ws * iz;
    const double m00 = wiz * ifxb, m11 = wiz * ifyb, m02 = wiz * u, m12 = wiz * vv;
    double t0 = (m00 * dq0 + m02 * dq2 - u * pfrU) * rho1;
    double t1 = (m11 * dq1 + m12 * dq2 - vv * pfrU) * rho1;
    double t2 = (-dr2dA * dq2 + dr2dDb * sDb) * rho1;

    // ---- backward
    const double yq0 = m00 * t0, yq1 = m11 * t1, yq2 = m02 * t0 + m12 * t1 - dr2dA * t2;
    aFb -= L.ws * (u * t0 + vv * t1);
    const double yX[3] = {RbU[0] * yq0 + RbU[1] * yq1 + RbU[2] * yq2, RbU[3] * yq0 + RbU[4] * yq1 + RbU[5] * yq2,
                          RbU[6] * yq0 + RbU[7] * yq1 + RbU[8] * yq2};
#pragma unroll
    for (int i = 0; i < 3; ++i) aT[i] += yX[i];
    Cx[0] += v[1] * yX[2] - v[2] * yX[1];
    Cx[1] += v[2] * yX[0] - v[0] * yX[2];
    Cx[2] += v[0] * yX[1] - v[1] * yX[0];
    aFa += Da * (Rcf[0] * yX[0] + Rcf[1] * yX[1] + Rcf[2] * yX[2]);
    if (N > 0) {
      const double ga = Rca[0] * yX[0] + Rca[1] * yX[1] + Rca[2] * yX[2];
      const double gb = dr2dDb * t2;
      // (the source depths are converted again here instead of being kept as doubles across the trip: with the next record
      // in flight the loop needs every register of its three-waves budget; the asm keeps the two conversions apart)
      float dxLate = d.x, dyLate = d.y;
      asm volatile("" : "+v"(dxLate), "+v"(dyLate));
      const double da = static_cast<double>(dxLate), db = static_cast<double>(dyLate);
      if constexpr (KD == 1) {
        gDa[0] += ga * da; gDa[1] += ga;
        gDb[0] += gb * db; gDb[1] += gb;
      } else {
      // (dense: this lane's private copy; qa / qb are swapped per direction, the private copies are indexed by role)
      const int pkey = (DENSE ? tid : (tid ^ (tid >> 5))) & (kPriv - 1);
      double* qaw = kUsePriv ? qpriv + (static_cast<size_t>(pkey) * 2 + dir) * B : qa;
      double* qbw = kUsePriv ? qpriv + (static_cast<size_t>(pkey) * 2 + (dir ^ 1)) * B : qb;
      const double gad = ga * da, gbd = gb * db;
#pragma unroll
      for (int k = 0; k < KD; ++k) {
        if (ta.ok(k)) {
          const int ia = ta.I(k);
          const double wa = ta.Wt(k);
          if (N == 2) {
            atomicAdd(&qaw[7 + ia * 2], gad * wa);
            atomicAdd(&qaw[7 + ia * 2 + 1], ga * wa);
          } else {
            atomicAdd(&qaw[7 + ia], gad * wa);
          }
        }
        if (tb.ok(k)) {
          const int ib = tb.I(k);
          const double wb = tb.Wt(k);
          if (N == 2) {
            atomicAdd(&qbw[7 + ib * 2], gbd * wb);
            atomicAdd(&qbw[7 + ib * 2 + 1], gb * wb);
          } else {
            atomicAdd(&qbw[7 + ib], gbd * wb);
          }
        }
      }
      }
    }
  }
  }  // dir
  MV_STAMP(4 + ((threadIdx.x >> 6) & 3));  // each wave's end of the constraint loop
  // ---- one workgroup reduction of the 11 (15) accumulators (roles of direction 1: source = fb, target = fa).
  // Lane pairs store their values transposed into LDS (row = accumulator, column = lane pair, 33-padded 32-column
  // segments), then 4 threads per accumulator sum one segment each: ~25 LDS ops per thread instead of 11 x 6
  // cross-lane butterfly steps.
  const FrameConst& Fa = *fcF[1];
  const FrameConst& Fb = *fcF[0];
  {
    double vals[NV];
#pragma unroll
    for (int i = 0; i < 3; ++i) { vals[i] = aT[i]; vals[3 + i] = aT0[i]; vals[6 + i] = Cx[i]; }
    vals[9] = aFa;
    vals[10] = aFb;
    if constexpr (KD == 1) { vals[11] = gDa[0]; vals[12] = gDa[1]; vals[13] = gDb[0]; vals[14] = gDb[1]; }
    // neighbouring lanes are summed in registers first (one DPP swap), so only the even lanes store: half the LDS
    const int half = tid >> 1;
    const int colw = (half >> 5) * 33 + (half & 31);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const double v = vals[i] + dppMove<0xB1>(vals[i]);
      if ((tid & 1) == 0) W[i * kRedStride + colw] = v;
    }
  }
  __syncthreads();
  constexpr int SEG = NT / 64;  // 32-column segments per accumulator row (lane pairs): 4 at 256 threads, 2 at 128
  if (tid < NV * SEG) {
    const double* row = W + (tid / SEG) * kRedStride + (tid % SEG) * 33;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k += 4) {
      s0 += row[k];
      s1 += row[k + 1];
      s2 += row[k + 2];
      s3 += row[k + 3];
    }
    double sacc = (s0 + s1) + (s2 + s3);
    if (SEG >= 2) sacc += dppMove<0xB1>(sacc);   // SEG consecutive lanes hold one accumulator's segments
    if (SEG == 4) sacc += dppMove<0x4E>(sacc);
    if (tid % SEG == 0) red[tid / SEG] = sacc;
  }
  __syncthreads();
  // red: [0..2] aT of direction 1 (source fb), [3..5] aT of direction 0 (source fa), [6..8] C = C_1 - C_0, [9] / [10] focal sums of
  // the source / target ROLE of direction 1, [11..14] the depth-block sums (KD == 1).  With dT = t_src - t_tgt of direction 1:
  //   q_t,src = aT_1 - aT_0 = -q_t,tgt;   q_w,src,i = a_src,i . (C - dT x aT_1);   q_w,tgt,i = a_tgt,i . (dT x aT_0 - C)
  if (tid < 3) {
    const double d = red[tid] - red[3 + tid];
    qa[tid] += d;
    qb[tid] -= d;
  } else if (tid < 9) {
    const bool tgt = tid >= 6;
    const int i = tgt ? tid - 6 : tid - 3;
    const double dT[3] = {Fa.t[0] - Fb.t[0], Fa.t[1] - Fb.t[1], Fa.t[2] - Fb.t[2]};
    const double* y = red + (tgt ? 3 : 0);
    const double m[3] = {red[6] - (dT[1] * y[2] - dT[2] * y[1]), red[7] - (dT[2] * y[0] - dT[0] * y[2]),
                         red[8] - (dT[0] * y[1] - dT[1] * y[0])};
    const double* a = (tgt ? Fb.Jl : Fa.Jl) + 3 * i;   // (frame fb is the source of direction 1)
    const double sdot = a[0] * m[0] + a[1] * m[1] + a[2] * m[2];
    if (tgt) qb[3 + i] -= sdot;
    else qa[3 + i] += sdot;
  } else if (tid == 9) {
    qa[6] += red[9] / Fa.fy;
  } else if (tid == 10) {
    qb[6] += red[10] / Fb.fy;
  } else if (KD == 1 && tid < 15) {
    const int n = (tid - 11) & 1;
    if (n < N) {
      if (tid < 13) qa[7 + n] += red[11 + n];
      else qb[7 + n] += red[13 + n];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) MV_STAMP(2);
  { double* t = qa; qa = qb; qb = t; }  // undo the role swap
  if constexpr (kUsePriv) {
    // private copy [k][0] collected frame fa's grid columns (source of direction 0, target of direction 1), [k][1] fb's
    for (int i = tid; i < B; i += NT) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int k = 0; k < kPriv; ++k) {
        sa += qpriv[(static_cast<size_t>(k) * 2 + 0) * B + i];
        sb += qpriv[(static_cast<size_t>(k) * 2 + 1) * B + i];
      }
      qa[i] += sa;
      qb[i] += sb;
    }
    __syncthreads();
  }
  // rows are grouped by frame (slot = position in the frame's item list) so that k_matvec_finish streams them
  double* outA = qPart + static_cast<size_t>(it.slot[item * 2]) * B;
  double* outB = qPart + static_cast<size_t>(it.slot[item * 2 + 1]) * B;
  for (int i = tid; i < B; i += NT) {
    outA[i] = qa[i];
    outB[i] = qb[i];
  }
  if (threadIdx.x == 0) MV_STAMP(3);
}

}  // namespace cvd

namespace cvd {

// =====================================================================================================
// Fast path of the frame-major assembly (same scope as k_matvec_pairs_fast: identity spatial transform,
// reprojection losses, Identity / Global / bilinear depth transform).  Only the OWN side's Jacobian is formed
// (3x7 pose-like columns + the 3-vector d r / d D); taps are unrolled, nothing is indexed dynamically, so the
// kernel needs neither scratch nor 256 VGPRs.  Accumulation: 7x7 + gradient in registers (wave-reduced at
// the end), pose x grid and grid x grid through LDS f64 atomics into the packed lower triangle.
// =====================================================================================================
constexpr int kAsmThreads = CVD_DETERMINISTIC ? 64 : 512;  // 8 waves per frame: two per SIMD at 256 VGPRs (deterministic build: one)

#ifdef CVD_ASM_PROFILE
__device__ unsigned long long g_asmProf[2048 * 16];
#define ASM_STAMP(slot) do { if (lane == 0) g_asmProf[(blockIdx.x & 2047) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define ASM_STAMP(slot) do {} while (0)
#endif
// STAGE (round 5): the OTHER frame's parameters of the unit a wave walks are copied into a per-wave LDS buffer first.  Without it
// the taps of the other side are 4-tap gathers from global memory -- a dependent round trip in every trip of a kernel that runs
// two waves per SIMD -- and, since `side` selects between an LDS and a global pointer at run time, every parameter read of the
// loop is a FLAT load.  With it both sides are LDS reads.  Needs 8 B more doubles of LDS: on whenever that fits (cvd_eval.hip).
// FOLD (round 6, dense mode inside the explicit-block scope): the constraints were walked by k_dense_walk (cvd_dense_walk.h); a
// frame's workgroup sums its records -- a gather over (pair, side) entries, no atomics -- in place of the walk, and continues with
// the regularisers and the write-out as ever.  One workgroup per frame (blockIdx.x = frame; the work list is not used).
template <int KD, bool DENSE = false, bool STAGE = false, bool FOLD = false>
inline __global__ __launch_bounds__(kAsmThreads) void k_assemble_fast(Layout L, Table T, const double* __restrict__ x,
                                                       const FrameConst* __restrict__ fc,
                                                       const double* __restrict__ mask, const float* __restrict__ median,
                                                       const unsigned char* __restrict__ regOwner,
                                                       const unsigned char* __restrict__ rangeFlags,
                                                       AsmWork work,
                                                       double* __restrict__ gOut, double* __restrict__ hOut,
                                                       double* __restrict__ costFrame, double* __restrict__ focalG,
                                                       double* __restrict__ focalH, DenseRecords dr = DenseRecords{}) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  constexpr double eps = 1e-6;
  const int B = L.B;
  const int npk = B * (B + 1) / 2;
  // LDS holds only what is accumulated into: the packed block, the gradient, this frame's parameters and the
  // reduction scratch ((B(B+1)/2 + 2B + 144) doubles: B = 199, the 16x12 grid, still fits 160 KiB).  The other
  // frame's parameters and both frames' FrameConst are read through L1/L2 (wave-uniform or 4-tap gathers).
  double* Hs = sm;
  double* gs = Hs + npk;
  double* xf = gs + B;
  double* red = xf + B;  // 36 workgroup sums (LDS atomics, one set per wave) + scratch
  double* xstage = red + 4 * 36;  // STAGE: kAsmThreads / 64 x B doubles
  const AsmPart me = FOLD ? AsmPart{static_cast<int>(blockIdx.x), 0, 0, 0, 1, 0} : work.parts[blockIdx.x];
  const int f = me.frame;
  const int tid = threadIdx.x;
  constexpr int NT = kAsmThreads;
  // wave-uniform wave index (scalar register: the per-entry frame constants below become scalar loads)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  if (tid == 0) ASM_STAMP(0);
  for (int i = tid; i < npk; i += NT) Hs[i] = 0.0;
  if (tid < 48) red[tid] = 0.0;
  for (int i = tid; i < B; i += NT) {
    gs[i] = 0.0;
    xf[i] = x[static_cast<size_t>(f) * B + i];
  }
  __syncthreads();
  if (tid == 0) ASM_STAMP(1);

  double PP[28], gp[7];
  // KD == 1 (Global / Identity): the single depth block is hit by every sample -> register accumulators
  // [0..13] pose x theta (7 x N), [14..16] theta x theta (lower), [17..18] gradient
  double GD[19];
#pragma unroll
  for (int i = 0; i < 19; ++i) GD[i] = 0.0;
  double cost = 0.0;
  double shG = 0.0, shH = 0.0;  // IntrinsicsOptimization::Shared: focal gradient / squared column norm
#pragma unroll
  for (int i = 0; i < 28; ++i) PP[i] = 0.0;
#pragma unroll
  for (int i = 0; i < 7; ++i) gp[i] = 0.0;
  const int N = L.N;
  const double A = L.aspect;

  if constexpr (FOLD) {
    if (L.includeStatic) {
      // output o: [0, 28) pose x pose (lower), [28, 35) pose gradient, 35 cost, then 36 + c G + v: c < 7 theta[v] x pose[c],
      // c = 7 gradient of theta[v], c = 8 + d band d of theta x theta.  Side 0 (source) / 1 (target) read different parts of a record.
      const int G = L.nD;
      const int recN = dwRecordDoubles(G);
      const int e0 = dr.fpOff[f], e1 = dr.fpOff[f + 1];
      const int nOut = 36 + 13 * G;
      for (int o = tid; o < nOut; o += NT) {
        int off0, off1, c = -1, v = 0;
        if (o < 28) {
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= o) ++i;
          const int j = o - i * (i + 1) / 2;
          off0 = i * 16 + j;
          off1 = (7 + i) * 16 + 7 + j;
        } else if (o < 35) {
          off0 = (o - 28) * 16 + 14;
          off1 = (7 + o - 28) * 16 + 14;
        } else if (o == 35) {
          off0 = 256 + 40 * G;
          off1 = -1;
        } else {
          const int e = o - 36;
          c = e / G;
          v = e - c * G;
          if (c < 7) { off0 = 256 + c * G + v; off1 = 256 + 15 * G + (7 + c) * G + v; }
          else if (c == 7) { off0 = 256 + 14 * G + v; off1 = 256 + 29 * G + v; }
          else { off0 = 256 + 30 * G + (c - 8) * G + v; off1 = 256 + 35 * G + (c - 8) * G + v; }
        }
        double s = 0.0;
        for (int e = e0; e < e1; ++e) {
          const int code = dr.fpList[e];
          const int off = (code & 1) ? off1 : off0;
          if (off < 0) continue;
          for (int q = dr.recOff[code >> 1]; q < dr.recOff[(code >> 1) + 1]; ++q) s += dr.records[static_cast<size_t>(q) * recN + off];
        }
        if (o < 36) atomicAdd(&red[o], s);
        else if (c < 7) Hs[packedIdx(7 + v, c)] = s;
        else if (c == 7) gs[7 + v] = s;
        else {
          const int d = c - 8;
          const int v2 = v + (d == 0 ? 0 : (d == 1 ? 1 : L.gx + d - 3));
          if (v2 < G) atomicAdd(&Hs[packedIdx(7 + v2, 7 + v)], s);  // (gx = 2: bands 1 and 2 are the same vertex pair)
        }
      }
    }
  } else
  if (L.includeStatic) {
    // one unit (slice of a (pair, side) entry) per WAVE at a time: the waves run through their units independently
    // (no barrier until the end), 64 lanes over the slice's constraints
    for (int u = me.u0 + wv; u < me.u1; u += NT / 64) {
      const int2 unit = work.units[u];
      const int code = __builtin_amdgcn_readfirstlane(unit.x);
      const int p = code >> 1;
      const int side = code & 1;  // 0: f is the source of pair p, 1: f is the target
      const int o = side ? T.pairA[p] : T.pairB[p];
      const double* __restrict__ xo = x + static_cast<size_t>(o) * B;
      const FrameConst& Fa = side ? fc[o] : fc[f];
      const FrameConst& Fb = side ? fc[f] : fc[o];
      double* xw = xstage + wv * B;
      if constexpr (STAGE) {
        // (a wave's LDS operations execute in order: its own earlier reads of the buffer are done before these writes land, and
        // the reads below see them -- no workgroup barrier, the waves walk their units independently)
        for (int i = lane; i < B; i += 64) xw[i] = xo[i];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      const double* xa = STAGE ? (side ? xw : xf) : (side ? xo : xf);
      const double* xb = STAGE ? (side ? xf : xw) : (side ? xf : xo);
      const double fya = Fa.fy, fxa = Fa.fy * A;
      const double fyb = Fb.fy;
      const double ifyb = 1.0 / fyb, ifxb = 1.0 / (fyb * A);
      const long long cBegin = T.pairOff[p] + __builtin_amdgcn_readfirstlane(unit.y);
      constexpr int kUnit = DENSE ? kAsmUnitDense : kAsmUnit;
      const long long cEnd = cBegin + kUnit < T.pairOff[p + 1] ? cBegin + kUnit : T.pairOff[p + 1];
      const int fsrc = side ? o : f, ftgt = side ? f : o;
      // list mode: lane = constraint, stride 64; dense mode: every lane walks its own run of kDenseRun pixels
      const long long cFirst = DENSE ? cBegin + static_cast<long long>(lane) * kDenseRun : cBegin + lane;
      const long long cStop = DENSE ? (cFirst + kDenseRun < cEnd ? cFirst + kDenseRun : cEnd) : cEnd;
      constexpr long long cStep = DENSE ? 1 : 64;
      // (list mode: no RecordStream -- a unit is two trips per lane and the kernel sits at its 256-register budget, the six
      // registers of a record in flight spill: 0.33 -> 0.35 ms.  Dense mode: mask and flow of the lane's next pixel in flight)
      RecordStream<true> rs;
      const int iFirst = static_cast<int>(cFirst - cBegin), iStop = static_cast<int>(cStop - cBegin);
      if constexpr (DENSE) rs.prime(T, cBegin, iFirst, iStop);
      for (long long c = cFirst; c < cStop; c += cStep) {
        float4 nd;
        float2 d;
        if constexpr (DENSE) {
          if (!rs.take(T, cBegin, static_cast<int>(c - cBegin), 1, iStop, T.pairOff[p], fsrc, ftgt, nd, d)) continue;
        } else {
          if (!loadConstraint<false>(T, c, T.pairOff[p], fsrc, ftgt, nd, d)) continue;
        }
        const double da = static_cast<double>(d.x), db = static_cast<double>(d.y);
        FastTaps<KD> ta, tb;
        fastGather<KD>(L, nd.x, nd.y, ta);
        fastGather<KD>(L, nd.z, nd.w, tb);
        double Da, Db;
        if (N == 0) {
          Da = da; Db = db;
        } else {
          Da = 0.0; Db = 0.0;
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            if (ta.ok(k)) {
              const int ia = ta.I(k);
              if (N == 2) Da += (da * xa[7 + ia * 2] + xa[7 + ia * 2 + 1]) * ta.Wt(k);
              else Da += da * xa[7 + ia] * ta.Wt(k);
            }
            if (tb.ok(k)) {
              const int ib = tb.I(k);
              if (N == 2) Db += (db * xb[7 + ib * 2] + xb[7 + ib * 2 + 1]) * tb.Wt(k);
              else Db += db * xb[7 + ib] * tb.Wt(k);
            }
          }
        }
        const double pax = static_cast<double>(nd.x), pay = static_cast<double>(nd.y);
        const double pbx = static_cast<double>(nd.z), pby = static_cast<double>(nd.w);
        const double ca[3] = {pax * fxa, pay * fya, -1.0};
        const double Rca[3] = {dot3(Fa.R, ca), dot3(Fa.R + 3, ca), dot3(Fa.R + 6, ca)};
        const double v[3] = {Fa.t[0] + Rca[0] * Da - Fb.t[0], Fa.t[1] + Rca[1] * Da - Fb.t[1],
                             Fa.t[2] + Rca[2] * Da - Fb.t[2]};
        const double q0 = Fb.R[0] * v[0] + Fb.R[3] * v[1] + Fb.R[6] * v[2];
        const double q1 = Fb.R[1] * v[0] + Fb.R[4] * v[1] + Fb.R[7] * v[2];
        const double q2 = Fb.R[2] * v[0] + Fb.R[5] * v[1] + Fb.R[8] * v[2];
        const double zz = -q2;
        const double iz = 1.0 / zz;
        const double u = q0 * iz * ifxb;
        const double vv = q1 * iz * ifyb;
        double r[3];
        r[0] = (u - pbx) * L.ws;
        r[1] = (vv - pby) * L.ws;
        double dr2dA, dr2dDb;
        if (L.lossType == kLossDisparity) {
          const bool zo = !(zz < eps), bo = !(Db < eps);
          const double izc = zo ? iz : 1.0 / eps, ibc = 1.0 / (bo ? Db : eps);
          r[2] = (izc - ibc) * L.wd;
          dr2dA = zo ? (-L.wd * izc * izc) : 0.0;
          dr2dDb = bo ? (L.wd * ibc * ibc) : 0.0;
        } else {
          const bool zIsMax = !(zz < Db), zIsMin = !(Db < zz);
          const double mx = zIsMax ? zz : Db, mn = zIsMin ? zz : Db;
          if (L.lossType == kLossRatio) {
            r[2] = (mx / mn - 1.0) * L.wd;
            const double dmx = 1.0 / mn, dmn = -mx / (mn * mn);
            dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
            dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
          } else {
            r[2] = log(mn / mx) * L.wd;
            const double dmn = 1.0 / mn, dmx = -1.0 / mx;
            dr2dA = ((zIsMax ? dmx : 0.0) + (zIsMin ? dmn : 0.0)) * L.wd;
            dr2dDb = ((zIsMax ? 0.0 : dmx) + (zIsMin ? 0.0 : dmn)) * L.wd;
          }
        }
        double rho0, w;  // rho, rho'
        robustRho(L, r[0] * r[0] + r[1] * r[1] + r[2] * r[2], rho0, w);
        if (!side) cost += rho0;

        // d r / d q (rows): M0 = (m00, 0, m02), M1 = (0, m11, m12), M2 = (0, 0, m22)
        const double wiz = L.ws * iz;
        const double m00 = wiz * ifxb, m11 = wiz * ifyb, m02 = wiz * u, m12 = wiz * vv, m22 = -dr2dA;
        double Jp[3][7];
        double JD[3];
        const FastTaps<KD>& tm = side ? tb : ta;
        const double dm = side ? db : da;
        if (!side) {
          // G = M R_b^T ; columns: t -> G, w_i -> G (D_a dR_a,i c_a), fy -> G (D_a R_a cf), D -> G R c_a
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
            Jp[rr][0] = G[rr][0]; Jp[rr][1] = G[rr][1]; Jp[rr][2] = G[rr][2];
            Jp[rr][6] = dot3(G[rr], dXdf);
            JD[rr] = dot3(G[rr], Rca);
          }
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double dX[3] = {Da * dot3(Fa.dR[i], ca), Da * dot3(Fa.dR[i] + 3, ca), Da * dot3(Fa.dR[i] + 6, ca)};
            Jp[0][3 + i] = dot3(G[0], dX);
            Jp[1][3 + i] = dot3(G[1], dX);
            Jp[2][3 + i] = dot3(G[2], dX);
          }
        } else {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            Jp[0][i] = -(m00 * Fb.R[i * 3 + 0] + m02 * Fb.R[i * 3 + 2]);
            Jp[1][i] = -(m11 * Fb.R[i * 3 + 1] + m12 * Fb.R[i * 3 + 2]);
            Jp[2][i] = -(m22 * Fb.R[i * 3 + 2]);
            const double* D = Fb.dR[i];  // d q / d w_b,i = dR_b,i^T v
            const double dq0 = D[0] * v[0] + D[3] * v[1] + D[6] * v[2];
            const double dq1 = D[1] * v[0] + D[4] * v[1] + D[7] * v[2];
            const double dq2 = D[2] * v[0] + D[5] * v[1] + D[8] * v[2];
            Jp[0][3 + i] = m00 * dq0 + m02 * dq2;
            Jp[1][3 + i] = m11 * dq1 + m12 * dq2;
            Jp[2][3 + i] = m22 * dq2;
          }
          Jp[0][6] = -L.ws * u * ifyb;
          Jp[1][6] = -L.ws * vv * ifyb;
          Jp[2][6] = 0.0;
          JD[0] = 0.0; JD[1] = 0.0; JD[2] = dr2dDb;
        }
        if (L.intrOpt == kIntrShared) {
          // one focal length: column = d r / d f_a + d r / d f_b (the other side's part is added here)
          if (!side) {
            Jp[0][6] += -L.ws * u * ifyb;
            Jp[1][6] += -L.ws * vv * ifyb;
          } else {
            const double cf[3] = {pax * A, pay, 0.0};
            const double dXdf[3] = {Da * (Fa.R[0] * cf[0] + Fa.R[1] * cf[1]), Da * (Fa.R[3] * cf[0] + Fa.R[4] * cf[1]),
                                    Da * (Fa.R[6] * cf[0] + Fa.R[7] * cf[1])};
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              Jp[0][6] += (m00 * Fb.R[i * 3 + 0] + m02 * Fb.R[i * 3 + 2]) * dXdf[i];
              Jp[1][6] += (m11 * Fb.R[i * 3 + 1] + m12 * Fb.R[i * 3 + 2]) * dXdf[i];
              Jp[2][6] += (m22 * Fb.R[i * 3 + 2]) * dXdf[i];
            }
          }
          if (!side) {
            shG += w * (Jp[0][6] * r[0] + Jp[1][6] * r[1] + Jp[2][6] * r[2]);
            shH += w * (Jp[0][6] * Jp[0][6] + Jp[1][6] * Jp[1][6] + Jp[2][6] * Jp[2][6]);
          }
        }
        // ---- accumulate
        int qi = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          gp[i] += w * (Jp[0][i] * r[0] + Jp[1][i] * r[1] + Jp[2][i] * r[2]);
#pragma unroll
          for (int j = 0; j <= i; ++j) {
            PP[qi] += w * (Jp[0][i] * Jp[0][j] + Jp[1][i] * Jp[1][j] + Jp[2][i] * Jp[2][j]);
            ++qi;
          }
        }
        if (N > 0) {
          double v7[7];
#pragma unroll
          for (int i = 0; i < 7; ++i) v7[i] = w * (Jp[0][i] * JD[0] + Jp[1][i] * JD[1] + Jp[2][i] * JD[2]);
          const double sDD = w * (JD[0] * JD[0] + JD[1] * JD[1] + JD[2] * JD[2]);
          const double sDr = w * (JD[0] * r[0] + JD[1] * r[1] + JD[2] * r[2]);
          if constexpr (KD == 16) {
            // bicubic: column factors come straight from the separable weights, taps outside the folded
            // footprint are skipped (d/d theta_k[0] = w_k d, d/d theta_k[1] = w_k)
#pragma unroll
            for (int ka = 0; ka < 16; ++ka) {
              if (!tm.ok(ka)) continue;
              const double wa = tm.Wt(ka);
              const int ia = tm.I(ka);
#pragma unroll
              for (int na = 0; na < 2; ++na) {
                if (na >= N) continue;
                const int ct = 7 + ia * N + na;
                const double fa = (N == 2 && na == 1) ? wa : wa * dm;
                const int rowBase = ct * (ct + 1) / 2;
                atomicAdd(&gs[ct], sDr * fa);
#pragma unroll
                for (int i = 0; i < 7; ++i) atomicAdd(&Hs[rowBase + i], v7[i] * fa);
#pragma unroll
                for (int kb = 0; kb <= ka; ++kb) {
                  if (!tm.ok(kb)) continue;
                  const double wb = tm.Wt(kb);
                  const int ib = tm.I(kb);
#pragma unroll
                  for (int nb = 0; nb < 2; ++nb) {
                    if (nb >= N || (kb == ka && nb > na)) continue;
                    const int c2 = 7 + ib * N + nb;
                    const double fb = (N == 2 && nb == 1) ? wb : wb * dm;
                    atomicAdd(&Hs[rowBase + c2], sDD * fa * fb);  // tap order is index order: c2 <= ct
                  }
                }
              }
            }
          } else {
          // tap column factors: value params (d/d theta_k[0] = w_k d, d/d theta_k[1] = w_k)
          double fac[KD * 2];
          int col[KD * 2];
#pragma unroll
          for (int k = 0; k < KD; ++k) {
            if (N == 2) {
              col[2 * k] = 7 + tm.I(k) * 2;       fac[2 * k] = tm.Wt(k) * dm;
              col[2 * k + 1] = col[2 * k] + 1;    fac[2 * k + 1] = tm.Wt(k);
            } else {
              col[k] = 7 + tm.I(k);               fac[k] = tm.Wt(k) * dm;
            }
          }
          const int nt = (N == 2) ? 2 * KD : KD;
          if constexpr (KD == 1) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
              if (a < nt) {
#pragma unroll
                for (int i = 0; i < 7; ++i) GD[a * 7 + i] += v7[i] * fac[a];
                GD[17 + a] += sDr * fac[a];
              }
            }
            GD[14] += sDD * fac[0] * fac[0];
            if (N == 2) { GD[15] += sDD * fac[1] * fac[0]; GD[16] += sDD * fac[1] * fac[1]; }
          } else {
#pragma unroll
          for (int a = 0; a < KD * 2; ++a) {
            if (a < nt) {
              const int ct = col[a];
              const int rowBase = ct * (ct + 1) / 2;
              atomicAdd(&gs[ct], sDr * fac[a]);
#pragma unroll
              for (int i = 0; i < 7; ++i) atomicAdd(&Hs[rowBase + i], v7[i] * fac[a]);
#pragma unroll
              for (int b = 0; b < KD * 2; ++b) {
                if (b <= a) {
                  const int c2 = col[b];
                  const int hi = ct > c2 ? ct : c2, lo = ct > c2 ? c2 : ct;
                  atomicAdd(&Hs[packedIdx(hi, lo)], sDD * fac[a] * fac[b]);
                }
              }
            }
          }
          }
          }
        }
      }
    }
  }
  ASM_STAMP(4 + wv);  // each wave's end of the constraint loop
  __syncthreads();
  if (tid == 0) ASM_STAMP(2);
  {
#pragma unroll
    for (int i = 0; i < 28; ++i) PP[i] = waveSum(PP[i]);
#pragma unroll
    for (int i = 0; i < 7; ++i) gp[i] = waveSum(gp[i]);
    cost = waveSum(cost);
    if constexpr (KD == 1) {
      if (N > 0) {
#pragma unroll
        for (int i = 0; i < 19; ++i) GD[i] = waveSum(GD[i]);
        if (lane == 0) {
          for (int a = 0; a < N; ++a) {
            const int ct = 7 + a;
            for (int i = 0; i < 7; ++i) atomicAdd(&Hs[ct * (ct + 1) / 2 + i], GD[a * 7 + i]);
            atomicAdd(&gs[ct], GD[17 + a]);
          }
          atomicAdd(&Hs[packedIdx(7, 7)], GD[14]);
          if (N == 2) { atomicAdd(&Hs[packedIdx(8, 7)], GD[15]); atomicAdd(&Hs[packedIdx(8, 8)], GD[16]); }
        }
      }
    }
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < 28; ++i) atomicAdd(&red[i], PP[i]);
#pragma unroll
      for (int i = 0; i < 7; ++i) atomicAdd(&red[28 + i], gp[i]);
      atomicAdd(&red[35], cost);
    }
  }
  __syncthreads();
  if (tid < 28) {
    int i = 0;
    while ((i + 1) * (i + 2) / 2 <= tid) ++i;
    const int j = tid - i * (i + 1) / 2;
    Hs[packedIdx(i, j)] += red[tid];
  } else if (tid < 35) {
    gs[tid - 28] += red[tid];
  }
  __syncthreads();
  const double staticCost = 0.5 * red[35];
  __syncthreads();
  if (tid == 0) ASM_STAMP(13);
  if (L.intrOpt == kIntrShared) {
    // The focal column of every constraint belongs to frame 0's slot: publish this frame's static focal
    // gradient / diagonal for k_shared_focal_fixup and drop the entries from the frame's own block (for f != 0
    // they are off-diagonal couplings with frame 0, which the block-Jacobi preconditioner does not hold).
    shG = waveSum(shG);
    shH = waveSum(shH);
    if (lane == 0) { red[wv] = shG; red[16 + wv] = shH; }
    __syncthreads();
    if (tid == 0) {
      double sg = 0.0, sh = 0.0;
      for (int w = 0; w < NT / 64; ++w) { sg += red[w]; sh += red[16 + w]; }
      focalG[f] = sg;  // (a split frame: overwritten with the sum over the parts below)
      focalH[f] = sh;
      red[42] = sg;
      red[43] = sh;
      gs[6] = 0.0;
      Hs[packedIdx(6, 6)] = 0.0;
    }
    if (f != 0) {
      for (int j = tid; j < B; j += NT)
        if (j != 6) Hs[j > 6 ? packedIdx(j, 6) : packedIdx(6, j)] = 0.0;
    }
    __syncthreads();
  }

  if (tid == 0) ASM_STAMP(14);
  double regCost = 0.0;
  if (regOwner[f] && me.part == 0) {
    const int nr = numRegResiduals<KD>(L);
    for (int i = tid; i < nr; i += NT) {
      double r;
      int n;
      int cols[2 * KD + 2];
      double jac[2 * KD + 2];
      regResidual<KD>(L, f, i, xf, median[f], r, n, cols, jac);
      regCost += r * r;
      for (int a = 0; a < n; ++a) {
        atomicAdd(&gs[cols[a]], jac[a] * r);
        for (int b = 0; b <= a; ++b) {
          const int hi = cols[a] > cols[b] ? cols[a] : cols[b];
          const int lo = cols[a] > cols[b] ? cols[b] : cols[a];
          atomicAdd(&Hs[packedIdx(hi, lo)], jac[a] * jac[b]);
        }
      }
    }
  }
  if (tid == 0 && L.positionRegSqrt > 0.0 && me.part == 0) {
    double o3[3] = {0, 0, 0}, dg = 0.0, cst = 0.0;
    posRegFrame(L, rangeFlags, f, x, nullptr, o3, dg, cst);
    for (int i = 0; i < 3; ++i) {
      atomicAdd(&gs[i], o3[i]);
      atomicAdd(&Hs[packedIdx(i, i)], dg);
    }
    regCost += cst;
  }
  regCost = waveSum(regCost);
  __syncthreads();
  if (tid == 0) ASM_STAMP(15);
  if (lane == 0) red[wv] = regCost;
  __syncthreads();
  if (tid == 0) {
    double rc = 0.0;
    for (int w = 0; w < NT / 64; ++w) rc += red[w];
    if (me.nParts == 1) costFrame[f] = staticCost + 0.5 * rc;
    red[40] = staticCost + 0.5 * rc;
  }
  if (me.nParts > 1) {
    // split frame: publish this part's packed block / gradient / cost; the last part to arrive folds the others in
    __syncthreads();
    const size_t stride = static_cast<size_t>(npk) + B + 4;
    double* mine = work.scratch + static_cast<size_t>(me.slot0 + me.part) * stride;
    for (int i = tid; i < npk; i += NT) mine[i] = Hs[i];
    for (int i = tid; i < B; i += NT) mine[npk + i] = gs[i];
    if (tid == 0) {
      mine[npk + B] = red[40];
      if (L.intrOpt == kIntrShared) { mine[npk + B + 1] = red[42]; mine[npk + B + 2] = red[43]; }
    }
    if (!lastBlockArrives(work.counters + f, static_cast<unsigned int>(me.nParts), reinterpret_cast<int*>(red + 41))) return;
#if CVD_DETERMINISTIC
    // (which part arrives last is a matter of timing: fold ALL parts in index order, this one's from its published copy)
    double costSum = 0.0, sgSum = 0.0, shSum = 0.0;
    for (int i = tid; i < npk; i += NT) Hs[i] = 0.0;
    for (int i = tid; i < B; i += NT) gs[i] = 0.0;
    for (int q = 0; q < me.nParts; ++q) {
#else
    double costSum = red[40];
    double sgSum = red[42], shSum = red[43];
    for (int q = 0; q < me.nParts; ++q) {
      if (q == me.part) continue;
#endif
      const double* other = work.scratch + static_cast<size_t>(me.slot0 + q) * stride;
      for (int i = tid; i < npk; i += NT) Hs[i] += other[i];
      for (int i = tid; i < B; i += NT) gs[i] += other[npk + i];
      costSum += other[npk + B];
      if (L.intrOpt == kIntrShared) { sgSum += other[npk + B + 1]; shSum += other[npk + B + 2]; }
    }
    if (tid == 0) {
      costFrame[f] = costSum;
      if (L.intrOpt == kIntrShared) { focalG[f] = sgSum; focalH[f] = shSum; }
    }
    __syncthreads();
  }

  if (tid == 0) ASM_STAMP(3);
  const double* mf = mask + static_cast<size_t>(f) * B;
  for (int i = tid; i < B; i += NT) gOut[static_cast<size_t>(f) * B + i] = gs[i] * mf[i];
  double* hf = hOut + static_cast<size_t>(f) * B * B;
  for (int idx = tid; idx < B * B; idx += NT) {
    const int i = idx / B, j = idx - i * B;
    const int hi = i > j ? i : j, lo = i > j ? j : i;
    hf[idx] = Hs[packedIdx(hi, lo)] * mf[i] * mf[j];
  }
  if (tid == 0) ASM_STAMP(12);
}

// AdaptiveDeformationCost constructor (reference lib/PoseOptimizer.cpp:560-618): every mask pixel is splatted bilinearly
// onto the four surrounding grid vertices, into the static (mask > 127) or the dynamic sums; vertex weight = dynamic /
// (dynamic + static).  One workgroup per frame, LDS accumulators (the sums are order-dependent only in the last bits).
inline __global__ __launch_bounds__(256) void k_adaptive_weights(const unsigned char* __restrict__ masks, int dw, int dh, int gw,
                                                          int gh, double* __restrict__ weights) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const int G = gw * gh;
  double* dyn = sm;
  double* sta = sm + G;
  const int f = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * G; i += blockDim.x) sm[i] = 0.0;
  __syncthreads();
  const unsigned char* m = masks + static_cast<size_t>(f) * dw * dh;
  for (int p = threadIdx.x; p < dw * dh; p += blockDim.x) {
    const int y = p / dw, x = p - y * dw;
    const double fy = static_cast<double>(y) * (gh - 1) / dh;
    const int iy = static_cast<int>(fy);
    const double ry = fy - iy;
    const double fx = static_cast<double>(x) * (gw - 1) / dw;
    const int ix = static_cast<int>(fx);
    const double rx = fx - ix;
    double* w = m[p] > 127 ? sta : dyn;
    atomicAdd(&w[iy * gw + ix], (1.0 - rx) * (1.0 - ry));
    atomicAdd(&w[iy * gw + ix + 1], rx * (1.0 - ry));
    atomicAdd(&w[(iy + 1) * gw + ix], (1.0 - rx) * ry);
    atomicAdd(&w[(iy + 1) * gw + ix + 1], rx * ry);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < G; i += blockDim.x) weights[static_cast<size_t>(f) * G + i] = dyn[i] / (dyn[i] + sta[i]);
}

}  // namespace cvd
