// cvd_frontend.hip -- the steps either side of the solve: constraint sampling, image operators, dense consumers, flow-guided filter.
#include "cvd_host.h"
#include <rocprim/device/device_segmented_radix_sort.hpp>

namespace cvd {

// ---- constraint sampling (cvd_sampling.h) -----------------------------------------------------------
// triplet == false: keyFrames = 2 x n frames (a, b) of the directed pairs, flow / mask = a -> b.
// triplet == true : keyFrames = n centre frames c, flow / mask = c -> c-1, flow2 / mask2 = c -> c+1.
void sampleConstraints(cvd_handle* h, bool triplet, int num, const int32_t* keyFrames, const float* corner,
                              const float* flow, const uint8_t* mask, const float* flow2, const uint8_t* mask2,
                              const float* dyn, int dw, int dh, int matchSeparation, float minDynamicDistance,
                              int64_t* offsets) {
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (matchSeparation < 0) throw std::runtime_error("matchSeparation must be >= 0");
  const int W = h->W, H = h->H;
  const size_t npx = static_cast<size_t>(W) * H;
  const int width = triplet ? 3 : 2;  // float2 per constraint
  if ((npx + 31) / 32 * 4 > kMaxLds) throw std::runtime_error("image too large for the LDS-resident sampling mask");
  for (int i = 0; i < (triplet ? 1 : 2) * num; ++i) {
    const int f = keyFrames[i];
    if (f < 0 || f >= h->F || (triplet && (f < 1 || f + 1 >= h->F))) throw std::runtime_error("sampling frame out of range");
  }
  hipStream_t s = h->stream;
  DevBuf<float> dCorner, dDyn;
  DevBuf<float2> dFlow, dFlow2, dSlab;
  DevBuf<unsigned char> dMaskS, dMaskS2, dTmp;
  DevBuf<int> dKeysF;
  DevBuf<unsigned long long> dKeys, dKeysOut;
  DevBuf<unsigned int> dNValid, dCount, dSeg;
  DevBuf<long long> dOff;
  dCorner.upload(corner, static_cast<size_t>(h->F) * npx, s);
  if (dyn) dDyn.upload(dyn, static_cast<size_t>(h->F) * dw * dh, s);
  dKeysF.upload(keyFrames, static_cast<size_t>(num) * (triplet ? 1 : 2), s);
  dFlow.upload(reinterpret_cast<const float2*>(flow), static_cast<size_t>(num) * npx, s);
  dMaskS.upload(mask, static_cast<size_t>(num) * npx, s);
  if (triplet) {
    dFlow2.upload(reinterpret_cast<const float2*>(flow2), static_cast<size_t>(num) * npx, s);
    dMaskS2.upload(mask2, static_cast<size_t>(num) * npx, s);
  }
  SamplingArgs A{W, H, h->invAspect, matchSeparation, minDynamicDistance, dCorner.p, dyn ? dDyn.p : nullptr,
                 dyn ? dw : W, dyn ? dh : H};
  // batches: keys (2 x 8 B) and the output slab (8 B x width) per pixel, ~1 GiB at a time
  const int PB = static_cast<int>(std::max<size_t>(1, std::min<size_t>(num, (size_t(1) << 30) / (npx * (16 + 8 * width)))));
  if (static_cast<size_t>(PB) * npx > 0xFFFFFFFFull) throw std::runtime_error("sampling batch too large");
  dKeys.ensure(static_cast<size_t>(PB) * npx);
  dKeysOut.ensure(static_cast<size_t>(PB) * npx);
  dSlab.ensure(static_cast<size_t>(PB) * npx * width);
  dNValid.ensure(PB);
  dCount.ensure(PB);
  std::vector<unsigned int> seg(PB + 1);
  for (int i = 0; i <= PB; ++i) seg[i] = static_cast<unsigned int>(static_cast<size_t>(i) * npx);
  dSeg.upload(seg.data(), seg.size(), s);
  size_t tmpBytes = 0;
  HIP_CHECK(rocprim::segmented_radix_sort_keys_desc(nullptr, tmpBytes, dKeys.p, dKeysOut.p,
                                                    static_cast<unsigned int>(static_cast<size_t>(PB) * npx),
                                                    static_cast<unsigned int>(PB), dSeg.p, dSeg.p + 1, 0, 64, s));
  dTmp.ensure(tmpBytes);
  DevBuf<float2>& result = triplet ? h->dSampledTrip : h->dSampledLoc;
  std::vector<long long> off(num + 1, 0);
  std::vector<unsigned int> cnt(PB);
  result.ensure(1);
  for (int p0 = 0; p0 < num; p0 += PB) {
    const int nb = std::min(PB, num - p0);
    HIP_CHECK(hipMemsetAsync(dNValid.p, 0, sizeof(unsigned int) * nb, s));
    const dim3 gridC(static_cast<unsigned>((npx + 255) / 256), nb);
    if (triplet)
      hipLaunchKernelGGL(k_fc_triplet_candidates, gridC, dim3(256), 0, s, A, p0, dKeysF.p, dFlow.p, dMaskS.p, dFlow2.p,
                         dMaskS2.p, dKeys.p, dNValid.p);
    else
      hipLaunchKernelGGL(k_fc_candidates, gridC, dim3(256), 0, s, A, p0, dKeysF.p, dFlow.p, dMaskS.p, dKeys.p, dNValid.p);
    HIP_CHECK(hipGetLastError());
    size_t tb = tmpBytes;
    HIP_CHECK(rocprim::segmented_radix_sort_keys_desc(dTmp.p, tb, dKeys.p, dKeysOut.p,
                                                      static_cast<unsigned int>(static_cast<size_t>(nb) * npx),
                                                      static_cast<unsigned int>(nb), dSeg.p, dSeg.p + 1, 0, 64, s));
    const size_t ldsBytes = (npx + 31) / 32 * 4;
    if (triplet) {
      allowLds(k_fc_greedy<true>, ldsBytes);
      hipLaunchKernelGGL(k_fc_greedy<true>, dim3(nb), dim3(64), ldsBytes, s, A, p0, dKeysOut.p, dNValid.p, dFlow.p,
                         dFlow2.p, dSlab.p, dCount.p);
    } else {
      allowLds(k_fc_greedy<false>, ldsBytes);
      hipLaunchKernelGGL(k_fc_greedy<false>, dim3(nb), dim3(64), ldsBytes, s, A, p0, dKeysOut.p, dNValid.p, dFlow.p,
                         static_cast<const float2*>(nullptr), dSlab.p, dCount.p);
    }
    HIP_CHECK(hipGetLastError());
    dCount.download(cnt.data(), nb, s);
    HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < nb; ++i) off[p0 + i + 1] = off[p0 + i] + cnt[i];
    // grow the result buffer and compact this batch into it
    const size_t total = static_cast<size_t>(off[p0 + nb]) * width;
    if (total > result.n) {
      DevBuf<float2> bigger;
      bigger.ensure(std::max<size_t>(total, result.n * 2));
      if (off[p0] > 0)
        HIP_CHECK(hipMemcpyAsync(bigger.p, result.p, sizeof(float2) * off[p0] * width, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
      std::swap(bigger.p, result.p);
      std::swap(bigger.n, result.n);
    }
    dOff.upload(off.data(), off.size(), s);
    hipLaunchKernelGGL(k_fc_compact, dim3(16, nb), dim3(256), 0, s, static_cast<int>(npx), width, p0, dOff.p, dSlab.p,
                       result.p);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipStreamSynchronize(s));
  }
  (triplet ? h->sampledTripOff : h->sampledOff) = off;
  for (int i = 0; i <= num; ++i) offsets[i] = off[i];
}

// ---- dense consumers of the result (cvd_dense.h) ----------------------------------------------
// kind 0: DepthXform::apply -> f32 [n][H][W]; 1: GridDepthXform::paramMap -> f64 [n][H][W][N];
// 2: SpatialXform::warp -> f32 [n][h][w][2] for the raster (w, h).  Host buffer out; the device buffer is kept for
// the next call.  Returns the kernel time in ms through *kernelMs when asked (HIP events on the solver stream).
void denseMaps(cvd_handle* h, int kind, int first, int count, int w, int hh, void* out, double* kernelMs) {
  if (h->F <= 0) throw std::runtime_error("no video set");
  if (first < 0 || count < 0 || first + count > h->F) throw std::runtime_error("frame range out of bounds");
  if (!h->poseParamsValid) posesToParams(h);
  cvd_opt_params p;
  cvd_opt_params_default(&p);
  Layout L = makeLayout(h, p, 0.0, PK_POSE_STEP);
  int KD, KS;
  tapCounts(L, KD, KS);
  if (kind == 1 && L.depthType != CVD_DEPTH_GRID)
    throw std::runtime_error("Parameter map not implemented for this transform type.");  // reference :422-425
  if (kind != 2) { w = h->W; hh = h->H; }
  if (w < 2 || hh < 2) throw std::runtime_error("raster too small");
  uploadState(h, L, h->dX);
  hipStream_t s = h->stream;
  const size_t pixels = static_cast<size_t>(count) * hh * w;
  const size_t bytes = pixels * (kind == 0 ? sizeof(float) : kind == 1 ? sizeof(double) * std::max(L.N, 1) : sizeof(float2));
  h->dDense.ensure((bytes + 7) / 8);
  if (count == 0) return;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventRecord(e0, s)); }
  const dim3 grid((w * hh + 255) / 256, 1, count), block(256);
  if (kind == 0) {
    CVD_DISPATCH_KD(KD, {
      hipLaunchKernelGGL((k_apply_depth<KD>), grid, block, 0, s, L, w, hh, first, h->dDepth.p, h->dX.p,
                         reinterpret_cast<float*>(h->dDense.p));
    });
  } else if (kind == 1) {
    CVD_DISPATCH_KD(KD, {
      hipLaunchKernelGGL((k_param_map<KD>), grid, block, 0, s, L, w, hh, first, h->dDepth.p, h->dX.p, h->dDense.p);
    });
  } else {
    if (KS == 0) hipLaunchKernelGGL((k_warp_map<0>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
    else if (KS == 4) hipLaunchKernelGGL((k_warp_map<4>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
    else hipLaunchKernelGGL((k_warp_map<16>), grid, block, 0, s, L, w, hh, first, h->dX.p, reinterpret_cast<float2*>(h->dDense.p));
  }
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dDense.p, bytes, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

// cornerMinEigenVal of n BGR float images (kind 0) / chamfer distance transform of n 8-bit masks (kind 1)
void imageOps(cvd_handle* h, int kind, int n, int w, int hh, const void* in, float* out, double* kernelMs) {
  if (n < 0 || w < 1 || hh < 1) throw std::runtime_error("invalid image batch");
  if (n == 0) return;
  if (!in) throw std::runtime_error("null image input");
  hipStream_t s = h->stream;
  const size_t px = static_cast<size_t>(w) * hh, pixels = px * n;
  if (pixels > (1ull << 31)) throw std::runtime_error("image batch too large for one call");
  h->dImgOut.ensure(pixels);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); }
  if (kind == 0) {
    h->dImgIn.ensure(pixels * 3);
    h->dImgGray.ensure(pixels);
    h->dImgCov.ensure(pixels * 3);
    HIP_CHECK(hipMemcpyAsync(h->dImgIn.p, in, pixels * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    if (kernelMs) HIP_CHECK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(k_bgr_to_gray, dim3(static_cast<unsigned>((pixels + 255) / 256)), dim3(256), 0, s, h->dImgIn.p, pixels,
                       h->dImgGray.p);
    const dim3 grid(static_cast<unsigned>((px + 255) / 256), 1, n);
    hipLaunchKernelGGL(k_sobel_cov, grid, dim3(256), 0, s, h->dImgGray.p, w, hh, h->dImgCov.p);
    hipLaunchKernelGGL(k_box_min_eigenval, grid, dim3(256), 0, s, h->dImgCov.p, w, hh, h->dImgOut.p);
  } else {
    const size_t tmpPer = static_cast<size_t>(w + 4) * (hh + 4);
    h->dImgMask.ensure(pixels);
    h->dImgTmp.ensure(tmpPer * n);
    HIP_CHECK(hipMemcpyAsync(h->dImgMask.p, in, pixels, hipMemcpyHostToDevice, s));
    if (kernelMs) HIP_CHECK(hipEventRecord(e0, s));
    const size_t lds = kChamferThreads * sizeof(long long) + static_cast<size_t>(w) * sizeof(unsigned int);
    if (lds > 64 * 1024) throw std::runtime_error("mask too wide for the distance transform kernel");
    hipLaunchKernelGGL(k_chamfer_5x5, dim3(n), dim3(kChamferThreads), lds, s, h->dImgMask.p, w, hh, h->dImgTmp.p,
                       h->dImgOut.p);
  }
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dImgOut.p, pixels * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

// Quaternion (x, y, z, w) times vector, the way Eigen evaluates it in float (uv = 2 q.vec x v; v + w uv + q.vec x uv)
static void quatRotate(const float* q, const float* v, float* out) {
  float uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; ++i) uv[i] += uv[i];
  const float c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; ++i) out[i] = v[i] + q[3] * uv[i] + c[i];
}

void flowGuidedFilter(cvd_handle* h, int n, int first, int count, int w, int hh, int dw, int dh, float invAspect,
                             const float* depth, const float* cameras, const float* flowF, const uint8_t* maskF,
                             const float* flowB, const uint8_t* maskB, int frameRadius, int spatialRadius, int median,
                             float* out, double* kernelMs) {
  if (n < 1 || first < 0 || count < 0 || first + count > n) throw std::runtime_error("invalid frame batch");
  if (w < 1 || hh < 1 || dw < 1 || dh < 1 || !(invAspect > 0.f)) throw std::runtime_error("invalid raster");
  if (frameRadius < 0 || spatialRadius < 0) throw std::runtime_error("negative filter radius");
  if (!depth || !cameras || (n > 1 && frameRadius > 0 && (!flowF || !maskF || !flowB || !maskB)))
    throw std::runtime_error("null filter input");
  if (count == 0) return;
  const long long side = 2ll * spatialRadius + 1, maxSamples = side * side * (2ll * frameRadius + 1);
  if (median && maxSamples > 256)
    throw std::runtime_error("flow guided median filter: (2 spatialRadius + 1)^2 (2 frameRadius + 1) > 256 samples per pixel");
  hipStream_t s = h->stream;
  const size_t px = static_cast<size_t>(w) * hh, dpx = static_cast<size_t>(dw) * dh;
  std::vector<FilterCam> cams(n);
  for (int k = 0; k < n; ++k) {
    const float* c = cameras + static_cast<size_t>(k) * 9;
    const float ex[3] = {1.f, 0.f, 0.f}, ey[3] = {0.f, 1.f, 0.f}, ez[3] = {0.f, 0.f, -1.f};
    for (int i = 0; i < 3; ++i) cams[k].pos[i] = c[i];
    quatRotate(c + 3, ex, cams[k].right);
    quatRotate(c + 3, ey, cams[k].up);
    quatRotate(c + 3, ez, cams[k].front);
    cams[k].tanH = std::tan(c[7] / 2.f);
    cams[k].tanV = std::tan(c[8] / 2.f);
  }
  h->dFltCams.upload(cams.data(), n, s);
  h->dFltDepth.upload(depth, dpx * n, s);
  const size_t links = n > 1 && frameRadius > 0 ? static_cast<size_t>(n - 1) : 0;
  h->dFltFlowF.upload(flowF, links * px * 2, s);
  h->dFltFlowB.upload(flowB, links * px * 2, s);
  h->dFltMaskF.upload(maskF, links * px, s);
  h->dFltMaskB.upload(maskB, links * px, s);
  h->dFltOut.ensure(px * count);
  FilterArgs A;
  A.n = n; A.first = first; A.count = count; A.w = w; A.h = hh; A.dw = dw; A.dh = dh; A.invAspect = invAspect;
  A.frameRadius = links ? frameRadius : 0; A.spatialRadius = spatialRadius; A.median = median;
  A.depth = h->dFltDepth.p; A.cams = h->dFltCams.p;
  A.flowFwd = reinterpret_cast<const float2*>(h->dFltFlowF.p); A.maskFwd = h->dFltMaskF.p;
  A.flowBwd = reinterpret_cast<const float2*>(h->dFltFlowB.p); A.maskBwd = h->dFltMaskB.p;
  A.out = h->dFltOut.p;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (kernelMs) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventRecord(e0, s)); }
  const dim3 grid(static_cast<unsigned>((px + 255) / 256), 1, count), block(256);
  if (!median) hipLaunchKernelGGL((k_flow_guided_filter<0>), grid, block, 0, s, A);
  else if (maxSamples <= 16) hipLaunchKernelGGL((k_flow_guided_filter<16>), grid, block, 0, s, A);
  else if (maxSamples <= 64) hipLaunchKernelGGL((k_flow_guided_filter<64>), grid, block, 0, s, A);
  else hipLaunchKernelGGL((k_flow_guided_filter<256>), grid, block, 0, s, A);
  HIP_CHECK(hipGetLastError());
  if (kernelMs) HIP_CHECK(hipEventRecord(e1, s));
  if (out) HIP_CHECK(hipMemcpyAsync(out, h->dFltOut.p, px * count * sizeof(float), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  if (kernelMs) {
    float ms = 0.f;
    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *kernelMs = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
  }
}

// One kernel of this translation unit's code object is looked up at handle creation: the HIP runtime loads a unit's device
// code at its first use, ~20 ms per unit that would otherwise land in the first solve of a process (cvd_create: loadDeviceCode).
void touchModule_frontend() {
  hipFuncAttributes a;
  (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(k_bgr_to_gray));
}

}  // namespace cvd
