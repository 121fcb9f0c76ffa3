// Flow-guided depth filter: DepthVideoProcessor::flowGuidedFilter, reference lib/Processor.cpp:315-590
// with DepthVideo::project, reference lib/DepthVideo.cpp:637-681.
//
// For every output pixel the reference gathers depth samples from a spatial window of the frame itself and from the
// chains obtained by following the optical flow forwards / backwards through up to frameRadius neighbouring frames
// (a chain stops at a masked-out flow vector or when it leaves the image), expresses each sample as a depth along the
// reference camera's forward axis, weights it by exp(-3 max(d, d_ref) / min(d, d_ref)) and takes the weighted mean or
// the weighted median.  One thread per output pixel; all inputs resident: the kernel streams flow (8 B) + mask (1 B)
// per chain step and one depth texel per sample, everything else is per-frame constants.  f32 throughout, in the
// reference's operation order (no FMA contraction); the only non-reproducible operation is expf (the device
// function is not glibc's), hence a float tolerance in the parity tests.
#pragma once
#include <hip/hip_runtime.h>

namespace cvd {

struct FilterCam {   // per frame: DepthPhoto::Extrinsics / Intrinsics expanded as DepthVideo::project uses them
  float pos[3], right[3], up[3], front[3];
  float tanH, tanV;  // tan(hFov / 2), tan(vFov / 2)
};

struct FilterArgs {
  int n;                    // frames in the batch (consecutive); batch frame k <-> video frame firstFrame + k
  int first, count;         // output frames [first, first + count) of the batch; chains may visit all n frames
  int w, h;                 // flow / output raster
  int dw, dh;               // depth raster
  float invAspect;
  int frameRadius, spatialRadius, median;
  const float* depth;       // [n][dh][dw]
  const FilterCam* cams;    // [n]
  const float2* flowFwd;    // [n-1][h][w]  k -> k+1
  const unsigned char* maskFwd;
  const float2* flowBwd;    // [n-1][h][w]  entry k: k+1 -> k
  const unsigned char* maskBwd;
  float* out;               // [count][h][w]
};

// depth of the sample at pixel position loc (flow raster units) of batch frame k, along the reference camera's axis
__device__ __forceinline__ float filterSampleDepth(const FilterArgs& A, int k, float lx, float ly, const float* refPos,
                                                   const float* refFwd) {
#pragma clang fp contract(off)
  const float nx = lx / static_cast<float>(A.w);
  const float ny = ly / static_cast<float>(A.h) * A.invAspect;
  int x = static_cast<int>(nx * static_cast<float>(A.dw) + 0.5f);
  int y = static_cast<int>(ny / A.invAspect * static_cast<float>(A.dh) + 0.5f);
  x = x < 0 ? 0 : (x > A.dw - 1 ? A.dw - 1 : x);  // (the reference clamps the upper side only; lower side: no wild reads)
  y = y < 0 ? 0 : (y > A.dh - 1 ? A.dh - 1 : y);
  const float depth = A.depth[(static_cast<size_t>(k) * A.dh + y) * A.dw + x];
  const FilterCam& c = A.cams[k];
  const float rx = -1.f + 2.f * nx;
  const float ry = 1.f - 2.f * ny / A.invAspect;
  const float a = rx * c.tanH, b = ry * c.tanV;
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float ray = (c.front[i] + c.right[i] * a) + c.up[i] * b;
    const float p = c.pos[i] + ray * depth;
    const float t = (p - refPos[i]) * refFwd[i];
    d = i == 0 ? t : d + t;
  }
  return d;
}

// Visits the samples of output pixel (x, y) of batch frame kf in the reference's order and hands (depth) to `visit`.
template <typename Visit>
__device__ __forceinline__ void filterForEachSample(const FilterArgs& A, int kf, int x, int y, const float* refPos,
                                                    const float* refFwd, Visit&& visit) {
#pragma clang fp contract(off)
  const int w = A.w, h = A.h;
  const int k0 = max(0, kf - A.frameRadius), k1 = min(A.n - 1, kf + A.frameRadius);
  const int y0 = max(0, y - A.spatialRadius), y1 = min(h - 1, y + A.spatialRadius);
  const int x0 = max(0, x - A.spatialRadius), x1 = min(w - 1, x + A.spatialRadius);
  const size_t px = static_cast<size_t>(w) * h;
  for (int wy = y0; wy <= y1; ++wy) {
    for (int wx = x0; wx <= x1; ++wx) {
      visit(filterSampleDepth(A, kf, static_cast<float>(wx), static_cast<float>(wy), refPos, refFwd));
#pragma unroll
      for (int dir = 0; dir < 2; ++dir) {  // forward chain, then backward chain
        float lx = static_cast<float>(wx), ly = static_cast<float>(wy);
        for (int k = kf + (dir ? -1 : 1); dir ? k >= k0 : k <= k1; k += dir ? -1 : 1) {
          // flow from the previous frame of the chain: forward k-1 -> k is entry k-1, backward k+1 -> k is entry k
          const size_t e = static_cast<size_t>(dir ? k : k - 1) * px;
          int ix = min(static_cast<int>(lx + 0.5f), w - 1);
          int iy = min(static_cast<int>(ly + 0.5f), h - 1);
          const size_t at = e + static_cast<size_t>(iy) * w + ix;
          if (!(dir ? A.maskBwd : A.maskFwd)[at]) break;
          const float2 f = (dir ? A.flowBwd : A.flowFwd)[at];
          lx += f.x;
          ly += f.y;
          ix = static_cast<int>(lx + 0.5f);
          iy = static_cast<int>(ly + 0.5f);
          if (ix < 0 || ix >= w || iy < 0 || iy >= h) break;
          visit(filterSampleDepth(A, k, lx, ly, refPos, refFwd));
        }
      }
    }
  }
}

__device__ __forceinline__ float filterWeight(float d, float ref) {
#pragma clang fp contract(off)
  const float value = fmaxf(d, ref) / fminf(d, ref);
  return expf(-value * 3.f);
}

template <int CAP>  // CAP = capacity of the per-thread sample list (median only); 0: mean
inline __global__ __launch_bounds__(256) void k_flow_guided_filter(FilterArgs A) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= A.w * A.h) return;
  const int kf = A.first + blockIdx.z;
  const int x = p % A.w, y = p / A.w;
  const FilterCam& rc = A.cams[kf];
  const float refPos[3] = {rc.pos[0], rc.pos[1], rc.pos[2]};
  const float refFwd[3] = {rc.front[0], rc.front[1], rc.front[2]};
  const float ref = filterSampleDepth(A, kf, static_cast<float>(x), static_cast<float>(y), refPos, refFwd);
  float result;
  if constexpr (CAP == 0) {
    float depthSum = 0.f, weightSum = 0.f;
    filterForEachSample(A, kf, x, y, refPos, refFwd, [&](float d) {
      const float wgt = filterWeight(d, ref);
      depthSum += d * wgt;
      weightSum += wgt;
    });
    result = weightSum > 0.f ? depthSum / weightSum : 0.f;
  } else {
    float ds[CAP], ws[CAP];
    int n = 0;
    float weightSum = 0.f;
    filterForEachSample(A, kf, x, y, refPos, refFwd, [&](float d) {
      const float wgt = filterWeight(d, ref);
      weightSum += wgt;
      // insertion into the list kept sorted by depth (ties: later sample after the earlier one)
      int i = n++;
      while (i > 0 && ds[i - 1] > d) {
        ds[i] = ds[i - 1];
        ws[i] = ws[i - 1];
        --i;
      }
      ds[i] = d;
      ws[i] = wgt;
    });
    const float half = weightSum / 2.f;
    float cum = 0.f;
    result = 0.f;  // (the reference leaves the pixel unwritten if no sample reaches half the weight, e.g. NaN weights)
    for (int i = 0; i < n; ++i) {
      cum += ws[i];
      if (cum >= half) { result = ds[i]; break; }
    }
  }
  A.out[(static_cast<size_t>(blockIdx.z) * A.h + y) * A.w + x] = result;
}

}  // namespace cvd
