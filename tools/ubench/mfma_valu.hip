// Does VALU work of one wavefront overlap with the MFMA chain of its SIMD neighbour on gfx950?
// Blocks of 512 threads = 8 waves = 2 per SIMD.  mode 0: every wave runs an MFMA chain;
// mode 1: every wave runs a VALU chain; mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per
// SIMD); mode 3: MFMA in waves 0-3 only (4-7 idle); mode 4: VALU in waves 4-7 only.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(int mode, int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
  const bool do_valu = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
  float r = 0.f;
  if (do_mfma) {
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  }
  if (do_valu) {
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f;
    // 4 independent FMA chains, 64 FMAs per iteration (= 256 VALU cycles, like 4 MFMAs' 256)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 1.0002f + 0.25f; v2 = v2 * 0.9999f + 0.125f; v3 = v3 * 0.9998f + 1.f;
      }
    }
    r = v0 + v1 + v2 + v3;
  }
  if (r == 123.456f) out[0] = r;
}

int main() {
  float* out; hipMalloc(&out, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4096;
  for (int mode = 0; mode < 5; ++mode) {
    k<<<256, 512>>>(mode, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<<<256, 512>>>(mode, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.1f us  (per iteration %.1f ns)\n", mode, ms * 1e3, ms * 1e6 / iters);
  }
  return 0;
}
