// Dense consumers of the optimizer's result: the per-pixel maps the reference's Python reads right
// after every optimisation (loaders/video_dataset.py:205,214) --
//   DepthXform::apply          reference lib/DepthMapTransform.cpp:394-415   transformed depth map of a frame
//   GridDepthXform::paramMap   reference lib/DepthMapTransform.cpp:950-994   interpolated value parameters per pixel
//   SpatialXform::warp         reference lib/DepthMapTransform.cpp:428-449   interpolated warp per pixel
// (one heap-allocated functor per pixel in the reference).  Here: one thread per pixel, all frames in one launch,
// the frame's control grid read through L1/L2 (<= 1.4 KB per frame), HBM-bound streaming kernels.
// NB these maps use the pixel-CENTRE convention loc = (-1 + x * 2/(w-1), 1 - y * 2/(h-1)) in f32 (the constraints
// use the pixel-edge one).
#pragma once

#include "cvd_kernels.h"

namespace cvd {

__device__ __forceinline__ void pixelLoc(int x, int y, int w, int h, float& lx, float& ly) {
  // The reference rounds x * xScale to f32 before the add.  HIP's __fmul_rn is a plain multiply that clang contracts
  // with the add into an FMA (fp-contract=fast is the HIP default, and `#pragma clang fp contract(off)` does not
  // survive the inlining here), which moves ~6 % of the columns by one ulp: pin the rounded product in a register.
  const float xs = __fdiv_rn(2.f, __fsub_rn(static_cast<float>(w), 1.f));
  const float ys = __fdiv_rn(2.f, __fsub_rn(static_cast<float>(h), 1.f));
  float px = static_cast<float>(x) * xs;
  float py = static_cast<float>(y) * ys;
  asm volatile("" : "+v"(px), "+v"(py));
  lx = -1.f + px;
  ly = 1.f - py;
}

// out[f][y][x] = D(depth[f][y][x]; theta_f) as f32
template <int KD>
inline __global__ __launch_bounds__(256) void k_apply_depth(Layout L, int W, int H, int frame0, const float* __restrict__ depth,
                                                     const double* __restrict__ x, float* __restrict__ out) {
  const int pidx = blockIdx.x * blockDim.x + threadIdx.x;  // pixels of one frame, flattened: no idle row tails
  if (pidx >= W * H) return;
  const int py = pidx / W, px = pidx - py * W;
  const int f = frame0 + blockIdx.z;
  const size_t pix = (static_cast<size_t>(f) * H + py) * W + px;
  const size_t opix = (static_cast<size_t>(blockIdx.z) * H + py) * W + px;
  const double d = static_cast<double>(depth[pix]);
  if (L.depthType == kDepthIdentity || L.N == 0) {
    out[opix] = static_cast<float>(d);
    return;
  }
  float lx, ly;
  pixelLoc(px, py, W, H, lx, ly);
  Taps<KD> t;
  depthGather<KD>(L, lx, ly, depth[pix], t);
  const double* th = x + static_cast<size_t>(f) * L.B + 7;
  double D = 0.0;
  for (int k = 0; k < t.n; ++k) {
    const double v = (L.N == 2) ? (d * th[t.idx[k] * 2] + th[t.idx[k] * 2 + 1]) : (d * th[t.idx[k]]);
    D += v * t.w[k];
  }
  out[opix] = static_cast<float>(D);
}

// out[f][y][x][n] = sum_k w_k theta_f[k][n] (f64, N channels)
template <int KD>
inline __global__ __launch_bounds__(256) void k_param_map(Layout L, int W, int H, int frame0, const float* __restrict__ depth,
                                                   const double* __restrict__ x, double* __restrict__ out) {
  const int pidx = blockIdx.x * blockDim.x + threadIdx.x;  // pixels of one frame, flattened: no idle row tails
  if (pidx >= W * H) return;
  const int py = pidx / W, px = pidx - py * W;
  const int f = frame0 + blockIdx.z;
  float lx, ly;
  pixelLoc(px, py, W, H, lx, ly);
  Taps<KD> t;  // (the source depth only matters for depth-wise grids: reference lib/DepthMapTransform.cpp:966-975)
  depthGather<KD>(L, lx, ly, depth[(static_cast<size_t>(f) * H + py) * W + px], t);
  const double* th = x + static_cast<size_t>(f) * L.B + 7;
  double a0 = 0.0, a1 = 0.0;
  for (int k = 0; k < t.n; ++k) {
    a0 += th[t.idx[k] * L.N] * t.w[k];
    if (L.N == 2) a1 += th[t.idx[k] * 2 + 1] * t.w[k];
  }
  double* o = out + ((static_cast<size_t>(blockIdx.z) * H + py) * W + px) * L.N;
  o[0] = a0;
  if (L.N == 2) o[1] = a1;
}

// out[f][y][x][2] = sum_k u_k phi_f[k][0..1] (f32), for a (h, w) raster
template <int KS>
inline __global__ __launch_bounds__(256) void k_warp_map(Layout L, int W, int H, int frame0, const double* __restrict__ x,
                                                  float2* __restrict__ out) {
  const int pidx = blockIdx.x * blockDim.x + threadIdx.x;  // pixels of one frame, flattened: no idle row tails
  if (pidx >= W * H) return;
  const int py = pidx / W, px = pidx - py * W;
  const int f = frame0 + blockIdx.z;
  float lx, ly;
  pixelLoc(px, py, W, H, lx, ly);
  double wx = 0.0, wy = 0.0;
  if constexpr (KS > 0) {
    Taps<KS> t;
    spatialGather<KS>(L, lx, ly, t);
    const double* ph = x + static_cast<size_t>(f) * L.B + 7 + L.nD;
    for (int k = 0; k < t.n; ++k) {
      wx += ph[t.idx[k] * 2] * t.w[k];
      wy += ph[t.idx[k] * 2 + 1] * t.w[k];
    }
  }
  out[(static_cast<size_t>(blockIdx.z) * H + py) * W + px] = make_float2(static_cast<float>(wx), static_cast<float>(wy));
}

}  // namespace cvd
