// Dev micro-benchmark: throughput of fp32 LDS atomic adds (ds_add_f32) on gfx950, random addresses in a 32 KB brick.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomic lds_atomic.hip && ./lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__global__ __launch_bounds__(256) void k(const int* __restrict__ idx, float* out, int per_thread) {
  __shared__ float brick[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) brick[i] = 0.f;
  __syncthreads();
  const int base = (blockIdx.x * 256 + threadIdx.x) * per_thread;
  for (int j = 0; j < per_thread; ++j) {
    const int a = idx[base + j] & 8191;
    if (MODE == 0) __hip_atomic_fetch_add(&brick[(a & ~63) | (threadIdx.x & 63)], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 1) atomicAdd(&brick[a], 1.0f);
    else __hip_atomic_fetch_add(&brick[a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 8192; i += 256) s += brick[i];
  if (s == -1.f) out[0] = s;
}

int main() {
  const int blocks = 1024, per_thread = 256;
  const size_t n = (size_t)blocks * 256 * per_thread;
  int* h = (int*)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) h[i] = rand();
  int* d; float* o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 4);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<blocks, 256>>>(d, o, per_thread);
      else if (mode == 1) k<1><<<blocks, 256>>>(d, o, per_thread);
      else k<2><<<blocks, 256>>>(d, o, per_thread);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("mode %d (%s): %.3f ms for %.1f M atomics = %.1f G atomics/s\n", mode,
           mode == 0 ? "conflict-free (lane = bank)" : mode == 1 ? "atomicAdd" : "__hip_atomic_fetch_add wg",
           best, n / 1e6, n / best / 1e6);
  }
  return 0;
}
