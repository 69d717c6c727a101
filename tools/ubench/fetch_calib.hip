// Calibration of the FETCH_SIZE counter on gfx950: known numbers of bytes / unique lines read by three
// access shapes -- a coalesced stream, 16-byte rows gathered at random, 16-byte rows at a 256-byte stride -- from a 2 GiB
// buffer (8 x the 256 MB Infinity Cache: every line comes from HBM).  Run under rocprofv3 --pmc (tools/fetch_calib.sh):
// FETCH_SIZE x 1 KiB against the known figure says whether the x2 of /opt/skills/guides/MI355X_MICROARCH.md applies to
// 16-byte gathers as it does to streams; the TCC_EA0_RDREQ_{32B,64B,128B} counters give the request sizes independently.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/fetch_calib.hip -o tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void calib_stream(const float4* __restrict__ buf, size_t n16, float* out) {      // n16 rows of 16 B, each once
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = buf[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *out = acc;
}
// row index of gather k: a bijective scramble of k over 2^bits rows of 16 B at a spacing of `spread` rows (the rows of
// different gathers are >= spread * 16 B apart: with spread >= 8 every gather touches its own 128-byte line)
__device__ __forceinline__ size_t scramble(size_t k, int bits) {
  const size_t mask = ((size_t)1 << bits) - 1;
  size_t x = k & mask;
  x = (x * 0x9E3779B97F4A7C15ull) & mask;            // odd multiplier: a bijection modulo 2^bits
  x ^= x >> (bits / 2);
  x = (x * 0xD6E8FEB86659FD93ull) & mask;
  return x;
}
__global__ void calib_gather_random(const float4* __restrict__ buf, size_t n_gather, int bits, int spread, float* out) {
  float acc = 0.f;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_gather; k += (size_t)gridDim.x * blockDim.x) {
    const float4 v = buf[scramble(k, bits) * spread]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *out = acc;
}
__global__ void calib_gather_stride(const float4* __restrict__ buf, size_t n_gather, int spread, float* out) {
  float acc = 0.f;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n_gather; k += (size_t)gridDim.x * blockDim.x) {
    const float4 v = buf[k * spread]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) *out = acc;
}

int main() {
  const size_t bytes = (size_t)2 << 30, n16 = bytes / 16;
  float4* buf; float* out;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
  CK(hipMemset(buf, 0, bytes));
  const int bits = 22;                      // 4 M gathers
  const size_t ng = (size_t)1 << bits;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CK(hipEventRecord(e0)); calib_stream<<<4096, 256>>>(buf, n16 / 4, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("calib_stream         %zu B known, %.3f ms (%.0f GB/s)\n", bytes / 4, ms, bytes / 4 / ms / 1e6);
    // flush the caches between shapes: stream another quarter of the buffer
    calib_stream<<<4096, 256>>>(buf + n16 / 2, n16 / 4, out);
    CK(hipEventRecord(e0)); calib_gather_random<<<4096, 256>>>(buf, ng, bits, 32, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("calib_gather_random  %zu rows of 16 B = %zu B useful, %zu unique 64-B / 128-B lines (= %zu / %zu B), %.3f ms\n", ng, ng * 16,
           ng, ng * 64, ng * 128, ms);
    calib_stream<<<4096, 256>>>(buf + n16 / 2, n16 / 4, out);
    CK(hipEventRecord(e0)); calib_gather_stride<<<4096, 256>>>(buf, ng, 16, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("calib_gather_stride  %zu rows of 16 B at 256 B = %zu B useful, %zu unique lines, %.3f ms\n", ng, ng * 16, ng, ms);
    calib_stream<<<4096, 256>>>(buf + n16 / 2, n16 / 4, out);
  }
  CK(hipDeviceSynchronize());
  return 0;
}
