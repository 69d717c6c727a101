// Microbenchmark: LDS fp32 accumulate throughput -- ds_add_f32 vs plain read-add-write,
// random vs grouped addresses.  Dev tool (feeds DESIGN.md), not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// MODE 0: atomicAdd (ds_add_f32), 1: plain RMW, 2: ds_add_rtn (use result)
// GROUP: lanes per contiguous run (1 = fully random, 8 = 8 consecutive floats per random vertex)
template <int MODE, int GROUP>
__global__ __launch_bounds__(256) void k(float* out, int iters, int words) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned st = (blockIdx.x * 4 + wave) * 2654435761u + 12345u;
  float acc = 0.f;
  const unsigned nv = words / GROUP;
  for (int it = 0; it < iters; ++it) {
    st = st * 1664525u + 1013904223u;
    unsigned grp = lane / GROUP;
    unsigned h = (st ^ (grp * 0x9E3779B9u));
    h ^= h >> 15; h *= 0x85EBCA6Bu; h ^= h >> 13;
    unsigned idx = (h % nv) * GROUP + lane % GROUP;
    if (MODE == 0) atomicAdd(&lds[idx], 1.0f);
    else if (MODE == 1) lds[idx] = lds[idx] + 1.0f;
    else acc += atomicAdd(&lds[idx], 1.0f);
  }
  __syncthreads();
  float s = acc;
  for (int i = threadIdx.x; i < words; i += blockDim.x) s += lds[i];
  if (s == -1.f) out[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = s;
}

template <int MODE, int GROUP>
void run(float* out, int words) {
  int blocks = 256 * 2, iters = 2048;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE, GROUP><<<blocks, 256, words * 4>>>(out, iters, words);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 3; ++r) k<MODE, GROUP><<<blocks, 256, words * 4>>>(out, iters, words);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
  double ops = (double)blocks * 256 * iters;
  printf("  mode=%d group=%d words=%5d: %8.1f us  %8.1f G lane-ops/s chip  = %.2f lane-ops/clk/CU @2.1GHz\n", MODE, GROUP, words,
         ms * 1e3, ops / ms / 1e6, ops / (ms * 1e-3) / 256 / 2.1e9);
}

int main() {
  float* out; CK(hipMalloc(&out, 64));
  for (int words : {4096, 8192}) {
    run<0, 1>(out, words); run<0, 8>(out, words); run<0, 16>(out, words); run<0, 64>(out, words);
    run<1, 1>(out, words); run<1, 8>(out, words); run<1, 64>(out, words);
    run<2, 8>(out, words);
  }
  return 0;
}
