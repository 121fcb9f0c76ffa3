// Development aid (GPU): throughput of LDS read-modify-write forms on gfx950, per CU -- ds_add_f64 (what the pair-major product
// and the assembly accumulate their grid columns with), ds_add_f32, ds_add_u32, and a plain ds_read_b64 + v_add_f64 + ds_write_b64.
// Every lane issues ITER operations on addresses (lane * stride + k * 64 * stride) mod n: stride 1 = conflict-free, distinct
// addresses; `same` > 1 makes groups of `same` neighbouring lanes hit one address.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o tools/lds_atomic_bench.bin tools/lds_atomic_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
constexpr int ITER = 2048;
constexpr int NLDS = 4096;  // doubles per workgroup (32 KB)

template <int MODE>
__global__ __launch_bounds__(256) void k_bench(double* out, int same, unsigned long long* cycles) {
  __shared__ double sm[NLDS];
  for (int i = threadIdx.x; i < NLDS; i += 256) sm[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x;
  const int base = (lane / same);
  const double v = 1.0 + lane;
  const unsigned long long t0 = clock64();
  int a = base;
#pragma unroll 8
  for (int k = 0; k < ITER; ++k) {
    a = (a + 257) & (NLDS - 1);
    if (MODE == 0) {
      atomicAdd(&sm[a], v);
    } else if (MODE == 1) {
      atomicAdd(reinterpret_cast<float*>(sm) + a, static_cast<float>(v));
    } else if (MODE == 2) {
      atomicAdd(reinterpret_cast<unsigned int*>(sm) + a, static_cast<unsigned int>(lane));
    } else if (MODE == 3) {
      sm[a] += v;  // (not atomic: lanes of other waves may race; throughput only)
    } else if (MODE == 4) {
      sm[a] = v;
    }
  }
  __syncthreads();
  const unsigned long long t1 = clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 256 + threadIdx.x] = sm[threadIdx.x];
}

template <int MODE>
static void run(const char* name, int groupsPerCu, int same, int numCU) {
  double* out; unsigned long long* cyc;
  const int G = groupsPerCu * numCU;
  CK(hipMalloc(&out, sizeof(double) * 256 * G)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * G));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_bench<MODE>, dim3(G), dim3(256), 0, 0, out, same, cyc);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_bench<MODE>, dim3(G), dim3(256), 0, 0, out, same, cyc);
  CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  // lane-operations per second per CU, and per shader clock at 2.4 GHz
  const double ops = static_cast<double>(G) * 256 * ITER;
  const double perCuPerSec = ops / (ms * 1e-3) / numCU;
  std::printf("%-28s groups/CU %d same %2d : %8.3f ms  %7.2f lane-ops / clk / CU (at 2.4 GHz)\n", name, groupsPerCu, same, ms, perCuPerSec / 2.4e9);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  int numCU = 0; CK(hipDeviceGetAttribute(&numCU, hipDeviceAttributeMultiprocessorCount, 0));
  std::printf("CUs %d\n", numCU);
  for (int g : {1, 4}) {
    for (int same : {1, 2, 4}) {
      run<0>("ds_add_f64", g, same, numCU);
      run<1>("ds_add_f32", g, same, numCU);
      run<2>("ds_add_u32", g, same, numCU);
    }
    run<3>("read + add + write (b64)", g, 1, numCU);
    run<4>("ds_write_b64", g, 1, numCU);
  }
  return 0;
}
