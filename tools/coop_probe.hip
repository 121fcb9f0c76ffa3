// Development aid (GPU): what the runtime says about cooperative launches of k_dense_spd_inverse on this device.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/coop_probe.bin tools/coop_probe.hip && tools/coop_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../robust_cvd_amd/csrc/cvd_device.h"
#include "../robust_cvd_amd/csrc/cvd_kernels.h"
#include "../robust_cvd_amd/csrc/cvd_dense_inverse.h"
using namespace cvd;

__global__ void k_trivial(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = 1; }
__global__ __launch_bounds__(512) void k_lds(int* p, int n) {
  extern __shared__ int smx[];
  for (int i = threadIdx.x; i < n; i += blockDim.x) smx[i] = i;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0) *p = smx[n - 1];
}

template <int TPW>
static void probe(size_t lds) {
  const void* f = reinterpret_cast<const void*>(&k_dense_spd_inverse<TPW>);
  hipFuncAttributes a{};
  hipError_t e = hipFuncGetAttributes(&a, f);
  (void)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  int nb = -1;
  hipError_t e2 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, f, kDinvNW * 64, lds);
  std::printf("TPW %2d: attr %s regs %d static lds %zu maxThreads %d | occupancy(%zu B dyn) -> %d blocks/CU (%s)\n", TPW, hipGetErrorName(e), a.numRegs,
              a.sharedSizeBytes, a.maxThreadsPerBlock, lds, nb, hipGetErrorName(e2));
}

int main() {
  int coop = -1, cus = 0;
  (void)hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  std::printf("cooperativeLaunch attr %d, CUs %d\n", coop, cus);
  int* d;
  (void)hipMalloc(&d, 4);
  void* args[] = {&d};
  for (int g : {1, 256, 2048, 4096, 100000}) {
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_trivial), dim3(g), dim3(64), args, 0, 0);
    std::printf("trivial coop launch grid %d: %s; sync %s\n", g, hipGetErrorName(e), hipGetErrorName(hipDeviceSynchronize()));
    (void)hipGetLastError();
  }
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int pass = 0; pass < 2; ++pass)
    for (size_t lds : {size_t(1024), size_t(28288), size_t(60000), size_t(80512)}) {
      int n = static_cast<int>(lds / 4);
      void* a2[] = {&d, &n};
      if (pass == 1) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
      for (hipStream_t q : {hipStream_t(nullptr), st}) {
        hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_lds), dim3(1), dim3(512), a2, static_cast<unsigned>(lds), q);
        std::printf("k_lds coop: attr %s, dyn %zu B, %s stream: %s; sync %s\n", pass ? "set" : "unset", lds, q ? "nonblocking" : "null",
                    hipGetErrorName(e), hipGetErrorName(hipDeviceSynchronize()));
        (void)hipGetLastError();
      }
    }
  {
    int n = 8, S = 1, nS = 1;
    double *A, *out, *panel; int* fail; unsigned int* bar;
    (void)hipMalloc(&A, 64 * 8); (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&panel, (2 * 256 + 512) * 8); (void)hipMalloc(&fail, 8); (void)hipMalloc(&bar, 16);
    (void)hipMemset(bar, 0, 16); (void)hipMemset(fail, 0, 8);
    double hA[64]; for (int i = 0; i < 64; ++i) hA[i] = (i / 8 == i % 8) ? 2.0 : 0.1;
    (void)hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice);
    double* pinv = panel + 2 * 256; int* valid = fail + 1;
    const double* Ac = A;
    void* a3[] = {&n, &S, &nS, &Ac, &out, &fail, &panel, &pinv, &bar, &valid};
    const size_t lds = static_cast<size_t>(4 * S + 1 + kDinvNW) * kInvTile * sizeof(double);
    hipError_t e = hipLaunchCooperativeKernel(reinterpret_cast<const void*>(&k_dense_spd_inverse<2>), dim3(1), dim3(kDinvNW * 64), a3, static_cast<unsigned>(lds), st);
    double o[2]; hipError_t es = hipDeviceSynchronize(); (void)hipMemcpy(o, out, 16, hipMemcpyDeviceToHost);
    std::printf("k_dense_spd_inverse<2> coop n = 8: %s; sync %s; out[0] %.6f\n", hipGetErrorName(e), hipGetErrorName(es), o[0]);
    (void)hipGetLastError();
  }
  probe<2>(28 * 1024);
  probe<8>(static_cast<size_t>(4 * 7 + 1 + kDinvNW) * kInvTile * 8);
  probe<13>(static_cast<size_t>(4 * 10 + 1 + kDinvNW) * kInvTile * 8);
  probe<25>(static_cast<size_t>(4 * 14 + 1 + kDinvNW) * kInvTile * 8);
  return 0;
}
