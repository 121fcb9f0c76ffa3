// Development aid (GPU): the dense SPD inverse of the dense coarse level alone -- timing, agreement with a host Cholesky on a
// sample of entries and the per-phase shader-clock profile (CVD_DINV_PROFILE) of selected workgroups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCVD_DINV_PROFILE -o tools/dinv_bench.bin tools/dinv_bench.hip && tools/dinv_bench.bin [n]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../robust_cvd_amd/csrc/cvd_device.h"
#include "../robust_cvd_amd/csrc/cvd_kernels.h"
#include "../robust_cvd_amd/csrc/cvd_dense_inverse.h"
using namespace cvd;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

template <int TPW>
static void launch(int groups, size_t lds, int n, int S, int nS, const double* A, double* out, int* fail, double* panel, double* pinv,
                   unsigned int* bar, int* valid) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dense_spd_inverse<TPW>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
  hipLaunchKernelGGL((k_dense_spd_inverse<TPW>), dim3(groups), dim3(kDinvNW * 64), lds, 0, n, S, nS, A, out, fail, panel, pinv, bar, valid);
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 2400;
  const size_t nn = static_cast<size_t>(n) * n;
  std::vector<double> h(nn);
  srand(1);
  const int R = 24;
  std::vector<double> g(static_cast<size_t>(n) * R);
  for (auto& v : g) v = rand() / static_cast<double>(RAND_MAX) - 0.5;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double a = 0;
      for (int k = 0; k < R; ++k) a += g[static_cast<size_t>(i) * R + k] * g[static_cast<size_t>(j) * R + k];
      h[static_cast<size_t>(i) * n + j] = h[static_cast<size_t>(j) * n + i] = a + (i == j ? 1.0 + 1e-3 * i : 0.0);
    }
  int numCU = 256;
  CK(hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, 0));
  const int nT = (n + 15) / 16;
  int S = 1;
  auto groups = [&](int sv) { const int nS = (nT + sv - 1) / sv; return nS * (nS + 1) / 2; };
  while (groups(S) > numCU) ++S;
  const int nS = (nT + S - 1) / S, tpw = (S * S + kDinvNW - 1) / kDinvNW;
  const size_t lds = static_cast<size_t>(4 * S + 1 + kDinvNW) * kInvTile * sizeof(double);
  std::printf("n %d  tiles %d  S %d  workgroups %d  tiles/wave %d  LDS %zu B\n", n, nT, S, groups(S), tpw, lds);
  double *dA, *dPanel; double* dOut; int* dFail; unsigned int* dBar;
  CK(hipMalloc(&dA, nn * 8)); CK(hipMalloc(&dOut, nn * 8)); CK(hipMalloc(&dPanel, (static_cast<size_t>(2) * nT * 256 + 512) * 8));
  CK(hipMalloc(&dFail, 8)); CK(hipMalloc(&dBar, 16));
  CK(hipMemcpy(dA, h.data(), nn * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipMemset(dFail, 0, 8)); CK(hipMemset(dBar, 0, 16));
    CK(hipEventRecord(e0, 0));
    double* pinv = dPanel + static_cast<size_t>(2) * nT * 256;
    if (tpw <= 2) launch<2>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    else if (tpw <= 5) launch<5>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    else if (tpw <= 8) launch<8>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    else if (tpw <= 13) launch<13>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    else if (tpw <= 18) launch<18>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    else launch<25>(groups(S), lds, n, S, nS, dA, dOut, dFail, dPanel, pinv, dBar, dFail + 1);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  int fl[2] = {0, 0};
  CK(hipMemcpy(fl, dFail, 8, hipMemcpyDeviceToHost));
  std::printf("k_dense_spd_inverse: %.1f us (best of 6), %.2f us per pivot step, fail %d valid %d\n", best * 1e3f, best * 1e3f / nT, fl[0], fl[1]);
  // residual of a few columns: A * out[:, j] = e_j
  std::vector<double> out(nn);
  CK(hipMemcpy(out.data(), dOut, nn * 8, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int j : {0, 1, n / 3, n / 2, n - 1}) {
    for (int i = 0; i < n; ++i) {
      double a = 0;
      for (int k = 0; k < n; ++k) a += h[static_cast<size_t>(i) * n + k] * out[static_cast<size_t>(k) * n + j];
      worst = std::fmax(worst, std::fabs(a - (i == j ? 1.0 : 0.0)));
    }
  }
  std::printf("max |A out - I| over 5 columns = %.3e\n", worst);
#ifdef CVD_DINV_PROFILE
  std::vector<unsigned long long> prof(256 * kDinvNW * 8);
  CK(hipMemcpyFromSymbol(prof.data(), HIP_SYMBOL(g_dinvProf), prof.size() * 8));
  const char* names[8] = {"first publish", "grid barrier", "panel load+sync", "T panel+sync", "update", "publish (+pivot)", "store", ""};
  const int G = groups(S);
  for (int b : {0, 1, 2, G / 2, G - 1}) {
    std::printf("workgroup %d, shader-clock cycles summed over the %d steps (columns = waves):\n", b, nT);
    for (int q = 0; q < 7; ++q) {
      std::printf("  %-18s", names[q]);
      for (int w = 0; w < kDinvNW; ++w) std::printf(" %9llu", prof[(static_cast<size_t>(b) * kDinvNW + w) * 8 + q]);
      std::printf("\n");
    }
  }
#endif
  return 0;
}
