// Image operators in front of the constraint sampler: what FlowConstraintsCollection::compute asks
// OpenCV for before it ranks pixels (reference lib/FlowConstraints.cpp:257-286, 417-423):
//   cvtColor(BGR2GRAY) + cornerMinEigenVal(gray, blockSize 3)   -> k_bgr_to_gray, k_sobel_cov, k_box_min_eigenval
//   distanceTransform(binarized dynamic mask, DIST_L2, DIST_MASK_5) -> k_chamfer_5x5
// OpenCV is a third-party dependency that is not part of /root/reference (and is not installed here): these kernels
// restate its published algorithms (imgproc corner.cpp / distransform.cpp); the float summation order inside OpenCV's
// SIMD row/column filters is not a documented contract, so the orders fixed below are ours (the oracle uses the same
// ones and must agree bit for bit; agreement with an OpenCV build is unpinned).
//
// All three corner kernels are one-thread-per-pixel streaming kernels (HBM-bound: 12 + 4 B read, 4 + 12 B written,
// 12 B re-read per pixel with the 3x3 neighbourhoods served by L2).  The chamfer transform is the classic two raster
// passes; each pass is sequential over rows, and inside a row the recurrence v[j] = min(c[j], v[j-1] + a) is a
// min-plus prefix scan, done by one workgroup per image with the three live rows in LDS.  Integer arithmetic
// (OpenCV's 16-bit fixed point) makes the parallel scan bit-identical to the sequential sweep.
#pragma once
#include <hip/hip_runtime.h>

namespace cvd {

__device__ __forceinline__ int reflect101(int i, int n) {  // BORDER_DEFAULT: gfedcb|abcdefgh|gfedcba
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
  return i;
}

// cvtColor(COLOR_BGR2GRAY) on float images: 0.114 B + 0.587 G + 0.299 R, left to right, no contraction.
inline __global__ __launch_bounds__(256) void k_bgr_to_gray(const float* __restrict__ bgr, size_t pixels, float* __restrict__ gray) {
#pragma clang fp contract(off)  // every product is rounded on its own (the __f*_rn intrinsics alone do not stop FMA fusion)
  const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
  if (i >= pixels) return;
  const float b = bgr[i * 3], g = bgr[i * 3 + 1], r = bgr[i * 3 + 2];
  gray[i] = (b * 0.114f + g * 0.587f) + r * 0.299f;
}

// Sobel 3x3 derivatives scaled by 1 / (2^(aperture-1) * blockSize) = 1/12 (the scale sits on the smoothing taps, as
// in OpenCV's Sobel), then cov = (dx dx, dx dy, dy dy).
inline __global__ __launch_bounds__(256) void k_sobel_cov(const float* __restrict__ gray, int w, int h, float* __restrict__ cov) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= w * h) return;
  const float* g = gray + static_cast<size_t>(blockIdx.z) * w * h;
  const int x = p % w, y = p / w;
  const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
  const int ym = reflect101(y - 1, h), yp = reflect101(y + 1, h);
  const float k0 = static_cast<float>(1.0 / 12.0), k1 = static_cast<float>(2.0 / 12.0);
  const int rows[3] = {ym, y, yp};
  float diff[3], smooth[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float* row = g + static_cast<size_t>(rows[r]) * w;
    diff[r] = row[xp] - row[xm];                              // [-1 0 1]
    smooth[r] = row[x] * k1 + (row[xm] + row[xp]) * k0;       // [1 2 1] / 12
  }
  const float dx = diff[1] * k1 + (diff[0] + diff[2]) * k0;
  const float dy = smooth[2] - smooth[0];
  float* c = cov + (static_cast<size_t>(blockIdx.z) * w * h + p) * 3;
  c[0] = dx * dx;
  c[1] = dx * dy;
  c[2] = dy * dy;
}

// boxFilter(3x3, normalize = false) of cov, then calcMinEigenVal: a = A/2, c = C/2, (a + c) - sqrt((a - c)^2 + B^2).
inline __global__ __launch_bounds__(256) void k_box_min_eigenval(const float* __restrict__ cov, int w, int h,
                                                          float* __restrict__ out) {
#pragma clang fp contract(off)
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= w * h) return;
  const float* cv = cov + static_cast<size_t>(blockIdx.z) * w * h * 3;
  const int x = p % w, y = p / w;
  const int xs[3] = {reflect101(x - 1, w), x, reflect101(x + 1, w)};
  const int ys[3] = {reflect101(y - 1, h), y, reflect101(y + 1, h)};
  float s[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float rs[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const float* row = cv + static_cast<size_t>(ys[r]) * w * 3;
      rs[r] = (row[xs[0] * 3 + k] + row[xs[1] * 3 + k]) + row[xs[2] * 3 + k];
    }
    s[k] = (rs[0] + rs[1]) + rs[2];
  }
  const float a = s[0] * 0.5f, b = s[1], c = s[2] * 0.5f;
  const float d = a - c;
  // correctly rounded float square root: through double (53 >= 2 * 24 + 2 bits, so the double rounding is exact);
  // __fsqrt_rn maps to the native ~1 ulp instruction here
  const float root = static_cast<float>(sqrt(static_cast<double>(d * d + b * b)));
  out[static_cast<size_t>(blockIdx.z) * w * h + p] = (a + c) - root;
}

// ---- distanceTransform(DIST_L2, DIST_MASK_5): chamfer weights 1, 1.4, 2.1969 in 16-bit fixed point -----------------
constexpr unsigned int kDistShift = 16;
constexpr unsigned int kDistMax = 0x7fffffffu >> 2;
constexpr unsigned int kDistHV = 65536u;     // cvRound(1.0    * 2^16)
constexpr unsigned int kDistDiag = 91750u;   // cvRound(1.4    * 2^16)
constexpr unsigned int kDistLong = 143976u;  // cvRound(2.1969 * 2^16)
constexpr int kChamferThreads = 512;

// inclusive min-plus scan along a row held in LDS: v[j] = min(c[j], v[j-1] + a) (forward) or with j+1 (backward).
// Each thread owns `per` consecutive elements; the carries cross threads through a block-wide prefix minimum of
// (value - position * a), in 64-bit so that nothing wraps.
__device__ inline void chamferRowScan(unsigned int* row, int w, bool backward, long long* part) {
  const int tid = threadIdx.x, nt = blockDim.x;
  const int per = (w + nt - 1) / nt;
  const int b0 = tid * per, b1 = min(w, b0 + per);
  const long long a = kDistHV;
  // position along the scan direction: s = j (forward) or w - 1 - j (backward)
  long long best = (1ll << 62);
  for (int s = b0; s < b1; ++s) {
    const int j = backward ? w - 1 - s : s;
    const long long key = static_cast<long long>(row[j]) - s * a;
    best = key < best ? key : best;
  }
  part[tid] = best;
  __syncthreads();
  // Hillis-Steele inclusive prefix minimum over the per-thread summaries
  for (int off = 1; off < nt; off <<= 1) {
    const long long other = tid >= off ? part[tid - off] : (1ll << 62);
    __syncthreads();
    part[tid] = other < part[tid] ? other : part[tid];
    __syncthreads();
  }
  long long carry = tid > 0 ? part[tid - 1] : (1ll << 62);
  for (int s = b0; s < b1; ++s) {
    const int j = backward ? w - 1 - s : s;
    const long long key = static_cast<long long>(row[j]) - s * a;
    carry = key < carry ? key : carry;
    row[j] = static_cast<unsigned int>(carry + s * a);
  }
  __syncthreads();
}

// One workgroup per image. tmp: (h + 4) x (w + 4) unsigned ints per image (2-pixel frame of kDistMax).
inline __global__ __launch_bounds__(kChamferThreads) void k_chamfer_5x5(const unsigned char* __restrict__ mask, int w, int h,
                                                                 unsigned int* __restrict__ tmpAll,
                                                                 float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smRaw[];
  long long* part = reinterpret_cast<long long*>(smRaw);                          // kChamferThreads
  unsigned int* cur = reinterpret_cast<unsigned int*>(part + kChamferThreads);    // w
  const int tid = threadIdx.x, nt = blockDim.x;
  const int step = w + 4;
  const unsigned char* src = mask + static_cast<size_t>(blockIdx.x) * w * h;
  unsigned int* tmp = tmpAll + static_cast<size_t>(blockIdx.x) * step * (h + 4);
  float* dst = out + static_cast<size_t>(blockIdx.x) * w * h;
  for (int i = tid; i < step * (h + 4); i += nt) tmp[i] = kDistMax;
  __syncthreads();
  // forward pass: rows top to bottom; row i of the image is row i + 2 of tmp, column j is j + 2
  for (int i = 0; i < h; ++i) {
    const unsigned int* r1 = tmp + static_cast<size_t>(i + 1) * step + 2;  // row above
    const unsigned int* r2 = tmp + static_cast<size_t>(i) * step + 2;      // two rows above
    for (int j = tid; j < w; j += nt) {
      unsigned int t0 = 0;
      if (src[static_cast<size_t>(i) * w + j] >= 127) {  // binarisation of the reference (:271)
        t0 = r2[j - 1] + kDistLong;
        t0 = min(t0, r2[j + 1] + kDistLong);
        t0 = min(t0, r1[j - 2] + kDistLong);
        t0 = min(t0, r1[j - 1] + kDistDiag);
        t0 = min(t0, r1[j] + kDistHV);
        t0 = min(t0, r1[j + 1] + kDistDiag);
        t0 = min(t0, r1[j + 2] + kDistLong);
        if (j == 0) t0 = min(t0, kDistMax + kDistHV);  // left frame pixel
      }
      cur[j] = t0;
    }
    __syncthreads();
    chamferRowScan(cur, w, false, part);
    unsigned int* r0 = tmp + static_cast<size_t>(i + 2) * step + 2;
    for (int j = tid; j < w; j += nt) r0[j] = cur[j];
    __threadfence_block();
    __syncthreads();
  }
  // backward pass: rows bottom to top
  for (int i = h - 1; i >= 0; --i) {
    unsigned int* r0 = tmp + static_cast<size_t>(i + 2) * step + 2;
    const unsigned int* r1 = tmp + static_cast<size_t>(i + 3) * step + 2;  // row below
    const unsigned int* r2 = tmp + static_cast<size_t>(i + 4) * step + 2;  // two rows below
    for (int j = tid; j < w; j += nt) {
      unsigned int t0 = r0[j];
      if (t0 > kDistHV) {
        t0 = min(t0, r2[j + 1] + kDistLong);
        t0 = min(t0, r2[j - 1] + kDistLong);
        t0 = min(t0, r1[j + 2] + kDistLong);
        t0 = min(t0, r1[j + 1] + kDistDiag);
        t0 = min(t0, r1[j] + kDistHV);
        t0 = min(t0, r1[j - 1] + kDistDiag);
        t0 = min(t0, r1[j - 2] + kDistLong);
        if (j == w - 1) t0 = min(t0, kDistMax + kDistHV);  // right frame pixel
      }
      cur[j] = t0;
    }
    __syncthreads();
    chamferRowScan(cur, w, true, part);
    for (int j = tid; j < w; j += nt) {
      r0[j] = cur[j];
      dst[static_cast<size_t>(i) * w + j] = __fmul_rn(__uint2float_rn(cur[j]), 1.0f / 65536.0f);
    }
    __threadfence_block();
    __syncthreads();
  }
}

}  // namespace cvd
