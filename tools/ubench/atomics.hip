// Microbenchmark: fp32 global atomic-add throughput vs. how many lanes of one
// instruction share a cache line.  Dev tool (feeds DESIGN.md), not product code.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// each wave instruction: 64/LPS segments of LPS consecutive dwords at random segment-aligned places
template <int LPS, int MODE>  // MODE 0 atomic, 1 plain store, 2 load+store (RMW non-atomic)
__global__ __launch_bounds__(256) void k(float* buf, const uint32_t* __restrict__ segidx, int iters, uint32_t nseg_mask) {
  int lane = threadIdx.x & 63;
  int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int sub = lane / LPS, within = lane % LPS;
  for (int it = 0; it < iters; ++it) {
    uint32_t s = segidx[(wave * iters + it) * (64 / LPS) + sub] & nseg_mask;
    float* p = buf + (int64_t)s * LPS + within;
    if (MODE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 4) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (MODE == 1) *p = 1.0f;
    else *p = *p + 1.0f;
  }
}

template <int LPS, int MODE>
void run(float* buf, size_t bytes, uint32_t* idx, int waves, int iters) {
  uint32_t nseg = (uint32_t)(bytes / 4 / LPS);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  int blocks = waves / 4;
  k<LPS, MODE><<<blocks, 256>>>(buf, idx, iters, nseg - 1);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) k<LPS, MODE><<<blocks, 256>>>(buf, idx, iters, nseg - 1);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
  double lanes = (double)waves * 64 * iters;
  printf("  LPS=%2d mode=%d: %8.1f us  %7.1f G lane-ops/s  %7.1f G segments/s\n", LPS, MODE, ms * 1e3,
         lanes / ms / 1e6, lanes / LPS / ms / 1e6);
}

int main(int argc, char** argv) {
  size_t mb = argc > 1 ? atoi(argv[1]) : 64;
  size_t bytes = mb << 20;
  float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  int waves = 4096 * 4, iters = 64;  // 67M lane-ops
  size_t nidx = (size_t)waves * iters * 64;
  std::vector<uint32_t> h(nidx);
  uint32_t st = 12345;
  for (auto& v : h) { st = st * 1664525u + 1013904223u; v = st >> 4; }
  uint32_t* idx; CK(hipMalloc(&idx, nidx * 4)); CK(hipMemcpy(idx, h.data(), nidx * 4, hipMemcpyHostToDevice));
  printf("buffer %zu MiB, %d waves x %d iters\n", mb, waves, iters);
  run<1, 0>(buf, bytes, idx, waves, iters); run<2, 0>(buf, bytes, idx, waves, iters);
  run<4, 0>(buf, bytes, idx, waves, iters); run<8, 0>(buf, bytes, idx, waves, iters);
  run<16, 0>(buf, bytes, idx, waves, iters); run<32, 0>(buf, bytes, idx, waves, iters);
  run<64, 0>(buf, bytes, idx, waves, iters);
  run<1, 3>(buf, bytes, idx, waves, iters); run<8, 3>(buf, bytes, idx, waves, iters);
  run<16, 3>(buf, bytes, idx, waves, iters); run<16, 4>(buf, bytes, idx, waves, iters);
  run<1, 1>(buf, bytes, idx, waves, iters); run<8, 1>(buf, bytes, idx, waves, iters);
  run<16, 1>(buf, bytes, idx, waves, iters); run<32, 1>(buf, bytes, idx, waves, iters);
  run<1, 2>(buf, bytes, idx, waves, iters); run<8, 2>(buf, bytes, idx, waves, iters);
  run<16, 2>(buf, bytes, idx, waves, iters);
  return 0;
}
