// Microbenchmark: LDS accumulate throughput by data type -- ds_add_u32 / ds_add_f32 / ds_add_f64 / ds_add_u64, random
// addresses (hash without a division).  Dev tool (feeds DESIGN.md), not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <typename T>
__global__ __launch_bounds__(256) void k(float* out, int iters, int mask) {
  extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
  T* lds = reinterpret_cast<T*>(raw);
  for (int i = threadIdx.x; i <= mask; i += blockDim.x) lds[i] = (T)0;
  __syncthreads();
  unsigned st = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    const unsigned idx = (st >> 12) & (unsigned)mask;
    __hip_atomic_fetch_add(&lds[idx], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  double s = 0;
  for (int i = threadIdx.x; i <= mask; i += blockDim.x) s += (double)lds[i];
  if (s == -1.0) out[0] = (float)s;
}

template <typename T>
void run(const char* name, float* out, int words) {
  int blocks = 256 * 2, iters = 4096;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<T><<<blocks, 256, words * sizeof(T)>>>(out, iters, words - 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) k<T><<<blocks, 256, words * sizeof(T)>>>(out, iters, words - 1);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  double ops = (double)blocks * 256 * iters;
  printf("  %-4s words=%5d: %8.1f us  %8.1f G lane-ops/s chip = %.2f lane-ops/clk/CU @2.1GHz\n", name, words, ms * 1e3,
         ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  for (int words : {2048, 4096}) {
    run<unsigned>("u32", out, words); run<float>("f32", out, words);
    run<double>("f64", out, words); run<unsigned long long>("u64", out, words);
  }
  return 0;
}
