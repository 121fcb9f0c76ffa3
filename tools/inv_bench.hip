// Development aid (GPU): the block-Jacobi inverse kernels alone -- timing, per-phase shader-clock profile of the blocked
// MFMA sweep (CVD_INV_PROFILE) and agreement with the scalar sweep.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVD_INV_PROFILE -o /tmp/inv_bench tools/inv_bench.hip && /tmp/inv_bench [F] [B]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../robust_cvd_amd/csrc/cvd_device.h"
#include "../robust_cvd_amd/csrc/cvd_kernels.h"
using namespace cvd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <typename K, typename... A>
static float timeIt(const char* name, int reps, K kernel, dim3 g, dim3 b, size_t lds, A... args) {
  if (lds > 48 * 1024) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kernel, g, b, lds, 0, args...);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kernel, g, b, lds, 0, args...);
  CK(hipEventRecord(e1, 0));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("%-34s %8.1f us / launch\n", name, ms * 1e3f / reps);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int F = argc > 1 ? std::atoi(argv[1]) : 300, B = argc > 2 ? std::atoi(argv[2]) : 177;
  const size_t n = static_cast<size_t>(F) * B * B;
  std::vector<double> h(n);
  srand(1);
  for (int f = 0; f < F; ++f) {
    std::vector<double> g(static_cast<size_t>(B) * B);
    for (auto& v : g) v = rand() / static_cast<double>(RAND_MAX) - 0.5;
    for (int i = 0; i < B; ++i)
      for (int j = 0; j < B; ++j) {
        double a = 0;
        for (int k = 0; k < 8; ++k) a += g[static_cast<size_t>(i) * B + k] * g[static_cast<size_t>(j) * B + k];
        h[(static_cast<size_t>(f) * B + i) * B + j] = a + (i == j ? 1.0 + 0.01 * i : 0.0);
      }
  }
  double *dH, *dL; float *dM0, *dM1; int* dF;
  CK(hipMalloc(&dH, n * 8)); CK(hipMalloc(&dL, static_cast<size_t>(F) * B * 8)); CK(hipMalloc(&dM0, n * 4)); CK(hipMalloc(&dM1, n * 4)); CK(hipMalloc(&dF, 4));
  CK(hipMemcpy(dH, h.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dL, 0, static_cast<size_t>(F) * B * 8)); CK(hipMemset(dF, 0, 4));
  Layout L{};
  L.F = F; L.B = B;
  const int nb6 = (B + 5) / 6, nT6 = nb6 * (nb6 + 1) / 2;
  if (nT6 <= 512) timeIt("k_block_inverse_sweep<1,6>", 20, k_block_inverse_sweep<1, 6>, dim3(F), dim3(((nT6 + 63) / 64) * 64), 0, L, dH, dL, dM0, dF);
  const int nbm = (B + 15) / 16, nTm = nbm * (nbm + 1) / 2;
  const size_t lds = static_cast<size_t>(std::max(2 * nbm + 1, 16)) * kInvTile * 8;
  if (nTm <= 80) timeIt("k_block_inverse_mfma<8,10>", 20, k_block_inverse_mfma<8, 10>, dim3(F), dim3(512), lds, L, dH, dL, dM1, dF);
  else timeIt("k_block_inverse_mfma<16,9>", 20, k_block_inverse_mfma<16, 9>, dim3(F), dim3(1024), lds, L, dH, dL, dM1, dF);
  std::vector<float> m0(n), m1(n);
  CK(hipMemcpy(m0.data(), dM0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(m1.data(), dM1, n * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < n; ++i) { md = std::fmax(md, std::fabs(static_cast<double>(m0[i]) - m1[i])); mx = std::fmax(mx, std::fabs(static_cast<double>(m0[i]))); }
  int fl = 0; CK(hipMemcpy(&fl, dF, 4, hipMemcpyDeviceToHost));
  std::printf("max |sweep - mfma| = %.3e (max |M| %.3e), failed pivots %d\n", md, mx, fl);
#ifdef CVD_INV_PROFILE
  unsigned long long prof[16 * 8];
  CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_invProf), sizeof(prof)));
  const char* names[8] = {"load", "A publish", "A pivot inverse", "barrier after A", "B (-T panel)", "barriers B/C", "C update", "store"};
  std::printf("shader-clock cycles, workgroup 0 (summed over the %d block steps):\n", nbm);
  for (int q = 0; q < 8; ++q) {
    std::printf("  %-18s", names[q]);
    for (int w = 0; w < 8; ++w) std::printf(" %8llu", prof[w * 8 + q]);
    std::printf("\n");
  }
#endif
  return 0;
}
