// Flow-constraint sampling: FlowConstraintsCollection::compute(PairKey) + sampleConstraints,
// reference lib/FlowConstraints.cpp:296-397 (Pixel ordering, disk mask, greedy suppression) and :400-465 (candidates).
//
// Per directed frame pair (a, b): every pixel of frame a whose flow mask is set, whose flow target rounds into the
// image and which is far enough from dynamic objects in both frames is a candidate, ranked by the corner response of
// frame a; candidates are accepted greedily in rank order unless an earlier accepted one lies within
// `matchSeparation` pixels (a disk).  Reference: one std::sort + a sequential sweep per pair on the CPU, O(P W H).
//
// Here: k_fc_candidates (thread / pixel, all pairs of a batch) writes one 64-bit key per pixel = (order-preserving
// bits of the corner strength, ~pixel index); one rocPRIM segmented radix sort, descending (ties resolve by ascending
// pixel index -- the reference's std::sort leaves tie order unspecified; unique keys, so the library's choice of a
// stable or unstable algorithm per segment size does not matter); k_fc_greedy runs the greedy sweep of one pair in ONE WAVE with the invalid-pixel
// bitmask in LDS: 64 candidates are tested per step, the survivors are accepted in rank order, and the 64 lanes stamp
// each accepted disk together.  The accepted constraints leave in rank order, scaled like scaleConstraint (:336-339).
#pragma once

#include "cvd_kernels.h"

namespace cvd {

struct SamplingArgs {
  int W, H;
  float invAspect;
  int matchSeparation;
  float minDynamicDistance;
  const float* corner;       // [F][H][W]
  const float* dyn;          // [F][dh][dw] distance to the nearest dynamic pixel, or nullptr (= FLT_MAX everywhere)
  int dw, dh;
};

// One candidate test, reference lib/FlowConstraints.cpp:427-459 (float arithmetic and int truncation as written there).
__device__ __forceinline__ bool fcCandidate(const SamplingArgs& A, int fa, int fb, int ix0, int iy0, float2 ff,
                                            unsigned char m, float& fx1, float& fy1) {
  const size_t dpl = static_cast<size_t>(A.dw) * A.dh;
  const float sx = __fdiv_rn(static_cast<float>(A.dw), static_cast<float>(A.W));
  const float sy = __fdiv_rn(static_cast<float>(A.dh), static_cast<float>(A.H));
  if (!m) return false;
  if (A.dyn != nullptr) {
    float t0 = static_cast<float>(iy0) * sy;
    float t1 = static_cast<float>(ix0) * sx;
    asm volatile("" : "+v"(t0), "+v"(t1));  // (rounded products: no FMA with the + 0.5f below)
    // (unchecked Mat access in the reference: a half-resolution mask can be indexed one past its last row; clamped)
    const int iy0s = min(static_cast<int>(t0 + 0.5f), A.dh - 1), ix0s = min(static_cast<int>(t1 + 0.5f), A.dw - 1);
    if (!(A.dyn[fa * dpl + static_cast<size_t>(iy0s) * A.dw + ix0s] > A.minDynamicDistance)) return false;
  }
  fx1 = __fadd_rn(static_cast<float>(ix0), ff.x);
  fy1 = __fadd_rn(static_cast<float>(iy0), ff.y);
  const int ix1 = static_cast<int>(__fadd_rn(fx1, 0.5f));
  const int iy1 = static_cast<int>(__fadd_rn(fy1, 0.5f));
  if (!(ix1 >= 0 && ix1 < A.W && iy1 >= 0 && iy1 < A.H)) return false;
  if (A.dyn != nullptr) {
    float t0 = fx1 * sx;
    float t1 = fy1 * sy;
    asm volatile("" : "+v"(t0), "+v"(t1));
    const int ix1s = min(max(static_cast<int>(t0 + 0.5f), 0), A.dw - 1), iy1s = min(max(static_cast<int>(t1 + 0.5f), 0), A.dh - 1);
    if (!(A.dyn[fb * dpl + static_cast<size_t>(iy1s) * A.dw + ix1s] > A.minDynamicDistance)) return false;
  }
  return true;
}

// keys: [pairs in batch][W*H]; invalid pixels get the smallest key and sink to the end of their segment.
__device__ __forceinline__ unsigned int orderedBits(float v) {  // ascending float order -> ascending unsigned order
  const unsigned int u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline __global__ __launch_bounds__(256) void k_fc_candidates(SamplingArgs A, int pair0, const int* __restrict__ pairFrames,
                                                       const float2* __restrict__ flow,
                                                       const unsigned char* __restrict__ mask,
                                                       unsigned long long* __restrict__ keys,
                                                       unsigned int* __restrict__ nValid) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int pb = blockIdx.y;  // pair inside the batch
  const int p = pair0 + pb;
  const int npx = A.W * A.H;
  bool ok = false;
  if (pix < npx) {
    const int iy0 = pix / A.W, ix0 = pix - iy0 * A.W;
    const int fa = pairFrames[2 * p], fb = pairFrames[2 * p + 1];
    const size_t gi = static_cast<size_t>(p) * npx + pix;
    float fx1, fy1;
    ok = fcCandidate(A, fa, fb, ix0, iy0, flow[gi], mask[gi], fx1, fy1);
    const size_t o = static_cast<size_t>(pb) * npx + pix;
    const unsigned int hi = ok ? orderedBits(A.corner[static_cast<size_t>(fa) * npx + pix]) : 0u;
    keys[o] = (static_cast<unsigned long long>(hi) << 32) | static_cast<unsigned int>(~static_cast<unsigned int>(pix));
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&nValid[pb], static_cast<unsigned int>(__popcll(b)));
}

// Triplet candidates, reference lib/FlowConstraints.cpp:467-545: centre pixel (ix1, iy1) of frame c with the flows
// c -> c-1 and c -> c+1.  Two index slips of the reference are kept: the corner response is read
// at column ix0 (the flow target in frame c-1) of row iy1, and the third dynamic-distance test reads frame c's map.
__device__ __forceinline__ bool fcTripletCandidate(const SamplingArgs& A, int fc, int ix1, int iy1, float2 f10, float2 f12,
                                                   unsigned char m10, unsigned char m12, float& fx0, float& fy0,
                                                   float& fx2, float& fy2, int& ix0) {
  const size_t dpl = static_cast<size_t>(A.dw) * A.dh;
  const float sx = __fdiv_rn(static_cast<float>(A.dw), static_cast<float>(A.W));
  const float sy = __fdiv_rn(static_cast<float>(A.dh), static_cast<float>(A.H));
  auto scaled = [&](float v, float sc, int hi) {
    float t = v * sc;
    asm volatile("" : "+v"(t));
    return min(max(static_cast<int>(t + 0.5f), 0), hi);
  };
  if (!(m10 && m12)) return false;
  if (A.dyn != nullptr) {
    const int iy1s = scaled(static_cast<float>(iy1), sy, A.dh - 1), ix1s = scaled(static_cast<float>(ix1), sx, A.dw - 1);
    if (!(A.dyn[fc * dpl + static_cast<size_t>(iy1s) * A.dw + ix1s] > A.minDynamicDistance)) return false;
  }
  fx0 = __fadd_rn(static_cast<float>(ix1), f10.x);
  fy0 = __fadd_rn(static_cast<float>(iy1), f10.y);
  ix0 = static_cast<int>(__fadd_rn(fx0, 0.5f));
  const int iy0 = static_cast<int>(__fadd_rn(fy0, 0.5f));
  fx2 = __fadd_rn(static_cast<float>(ix1), f12.x);
  fy2 = __fadd_rn(static_cast<float>(iy1), f12.y);
  const int ix2 = static_cast<int>(__fadd_rn(fx2, 0.5f));
  const int iy2 = static_cast<int>(__fadd_rn(fy2, 0.5f));
  if (!(ix0 >= 0 && ix0 < A.W && iy0 >= 0 && iy0 < A.H && ix2 >= 0 && ix2 < A.W && iy2 >= 0 && iy2 < A.H)) return false;
  if (A.dyn != nullptr) {
    const int ix0s = scaled(fx0, sx, A.dw - 1), iy0s = scaled(fy0, sy, A.dh - 1);
    const int ix2s = scaled(fx2, sx, A.dw - 1), iy2s = scaled(fy2, sy, A.dh - 1);
    if (!(A.dyn[(fc - 1) * dpl + static_cast<size_t>(iy0s) * A.dw + ix0s] > A.minDynamicDistance)) return false;
    if (!(A.dyn[fc * dpl + static_cast<size_t>(iy2s) * A.dw + ix2s] > A.minDynamicDistance)) return false;  // (sic: frame c)
  }
  return true;
}

inline __global__ __launch_bounds__(256) void k_fc_triplet_candidates(SamplingArgs A, int g0, const int* __restrict__ centers,
                                                               const float2* __restrict__ flow10,
                                                               const unsigned char* __restrict__ mask10,
                                                               const float2* __restrict__ flow12,
                                                               const unsigned char* __restrict__ mask12,
                                                               unsigned long long* __restrict__ keys,
                                                               unsigned int* __restrict__ nValid) {
  const int pix = blockIdx.x * blockDim.x + threadIdx.x;
  const int gb = blockIdx.y;
  const int g = g0 + gb;
  const int npx = A.W * A.H;
  bool ok = false;
  if (pix < npx) {
    const int iy1 = pix / A.W, ix1 = pix - iy1 * A.W;
    const int fc = centers[g];
    const size_t gi = static_cast<size_t>(g) * npx + pix;
    float fx0, fy0, fx2, fy2;
    int ix0 = 0;
    ok = fcTripletCandidate(A, fc, ix1, iy1, flow10[gi], flow12[gi], mask10[gi], mask12[gi], fx0, fy0, fx2, fy2, ix0);
    const unsigned int hi = ok ? orderedBits(A.corner[static_cast<size_t>(fc) * npx + static_cast<size_t>(iy1) * A.W + ix0]) : 0u;
    keys[static_cast<size_t>(gb) * npx + pix] =
        (static_cast<unsigned long long>(hi) << 32) | static_cast<unsigned int>(~static_cast<unsigned int>(pix));
  }
  const unsigned long long b = __ballot(ok);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&nValid[gb], static_cast<unsigned int>(__popcll(b)));
}

// One wave per pair: greedy disk suppression over the rank-ordered candidates.
// slab: [pairs in batch][W*H][4] accepted constraints in rank order; count[pb] = how many.
// TRIPLET: flow = c -> c-1, flow2 = c -> c+1 and the slab holds 6 floats per constraint (3 x float2).
template <bool TRIPLET>
inline __global__ __launch_bounds__(64) void k_fc_greedy(SamplingArgs A, int pair0, const unsigned long long* __restrict__ sortedKeys,
                                                  const unsigned int* __restrict__ nValid,
                                                  const float2* __restrict__ flow, const float2* __restrict__ flow2,
                                                  float2* __restrict__ slab, unsigned int* __restrict__ count) {
  extern __shared__ unsigned int invalid[];  // W*H bits
  const int pb = blockIdx.x, lane = threadIdx.x;
  const int p = pair0 + pb;
  const int npx = A.W * A.H;
  const int words = (npx + 31) / 32;
  for (int i = lane; i < words; i += 64) invalid[i] = 0u;
  __syncthreads();
  const unsigned int n = nValid[pb];
  const unsigned long long* cand = sortedKeys + static_cast<size_t>(pb) * npx;
  const float sxo = __fdiv_rn(1.f, static_cast<float>(A.W));
  const float syo = __fdiv_rn(A.invAspect, static_cast<float>(A.H));
  const int r = A.matchSeparation;
  const int side = 2 * r + 1;
  unsigned int nOut = 0;
  for (unsigned int base = 0; base < n; base += 64) {
    const unsigned int k = base + lane;
    unsigned int pix = 0;
    bool free_ = false;
    if (k < n) {
      pix = ~static_cast<unsigned int>(cand[k]);  // low word of the key
      free_ = !((invalid[pix >> 5] >> (pix & 31)) & 1u);
    }
    unsigned long long todo = __ballot(free_);
    while (todo) {
      const int l = __ffsll(static_cast<long long>(todo)) - 1;
      todo &= todo - 1;
      const unsigned int cp = __shfl(pix, l, 64);
      // an earlier survivor of this very step may have covered it
      if ((invalid[cp >> 5] >> (cp & 31)) & 1u) continue;  // uniform: every lane reads the same word
      const int cy = static_cast<int>(cp) / A.W, cx = static_cast<int>(cp) - cy * A.W;
      if (lane == 0) {
        const float2 ff = flow[static_cast<size_t>(p) * npx + cp];
        const float fxa = __fadd_rn(static_cast<float>(cx), ff.x), fya = __fadd_rn(static_cast<float>(cy), ff.y);
        const float2 here = make_float2(__fmul_rn(static_cast<float>(cx), sxo), __fmul_rn(static_cast<float>(cy), syo));
        const float2 there = make_float2(__fmul_rn(fxa, sxo), __fmul_rn(fya, syo));
        if constexpr (TRIPLET) {
          const float2 f2 = flow2[static_cast<size_t>(p) * npx + cp];
          const float fxb = __fadd_rn(static_cast<float>(cx), f2.x), fyb = __fadd_rn(static_cast<float>(cy), f2.y);
          float2* o = slab + (static_cast<size_t>(pb) * npx + nOut) * 3;
          o[0] = there;  // frame c-1
          o[1] = here;   // frame c (the reference pixel)
          o[2] = make_float2(__fmul_rn(fxb, sxo), __fmul_rn(fyb, syo));
        } else {
          float2* o = slab + (static_cast<size_t>(pb) * npx + nOut) * 2;
          o[0] = here;
          o[1] = there;
        }
      }
      ++nOut;
      // stamp the disk (reference buildDiskMask: dx^2 + dy^2 <= r^2), clipped to the image
      for (int c = lane; c < side * side; c += 64) {
        const int dy = c / side - r, dx = c - (c / side) * side - r;
        const int mx = cx + dx, my = cy + dy;
        if (dx * dx + dy * dy <= r * r && mx >= 0 && mx < A.W && my >= 0 && my < A.H) {
          const unsigned int q = static_cast<unsigned int>(my * A.W + mx);
          atomicOr(&invalid[q >> 5], 1u << (q & 31));
        }
      }
      __syncthreads();  // single-wave workgroup: LDS writes of the stamp are visible to the next test
    }
  }
  if (lane == 0) count[pb] = nOut;
}

// compaction of one batch: slab rows -> the pair's slice of the output
inline __global__ __launch_bounds__(256) void k_fc_compact(int npx, int width, int pair0, const long long* __restrict__ offsets,
                                                    const float2* __restrict__ slab, float2* __restrict__ out) {
  const int pb = blockIdx.y;
  const long long o0 = offsets[pair0 + pb] * width, n = (offsets[pair0 + pb + 1] - offsets[pair0 + pb]) * width;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n; i += static_cast<long long>(gridDim.x) * 256)
    out[o0 + i] = slab[static_cast<size_t>(pb) * npx * width + i];
}

}  // namespace cvd
