// Do bf16 matrix instructions of one wavefront and vector instructions of ANOTHER wavefront on the same SIMD overlap?
// (round 6; companion of mfma_valu.hip, which asked the same of the fp32 matrix instructions inside one wavefront)
// 256 workgroups of 8 wavefronts: wavefronts 0-3 run `nm` rounds of 16 independent v_mfma_f32_32x32x16_bf16, wavefronts
// 4-7 run `nv` rounds of 64 vector instructions (the split's mix: cvt_pk, and, shift, sub).  Timed: matrix only, vector
// only, both; and the same with the two roles inside ONE wavefront (interleaved by the compiler).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu_bf16.hip -o tools/ubench/mfma_valu_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t cvt_pk(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

template <int MODE>   // 0: roles by wavefront; 1: every wavefront does both (half the rounds each)
__global__ __launch_bounds__(512, 1) void k(float* out, int nm, int nv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = lane + j;
  u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u}, b = {0x3f803f80u, 0x3e803f80u + lane, 0x3f803f80u, 0x3f803f80u};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + lane * 0.001f + i;
  const bool do_m = MODE == 1 || wave < 4, do_v = MODE == 1 || wave >= 4;
  const int rm = MODE == 1 ? nm / 2 : nm, rv = MODE == 1 ? nv / 2 : nv;
  if (MODE == 0) {
    if (do_m)
      for (int it = 0; it < rm; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
      }
    if (do_v)
      for (int it = 0; it < rv; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; i += 2) {      // 8 instructions per pair, 4 pairs, twice = 64
            uint32_t h = cvt_pk(v[i], v[i + 1]);
            float ra = v[i] - __uint_as_float(h << 16), rb = v[i + 1] - __uint_as_float(h & 0xffff0000u);
            uint32_t m = cvt_pk(ra, rb);
            v[i] = ra + __uint_as_float(m << 16) * 0.5f;
            v[i + 1] = rb + __uint_as_float(m & 0xffff0000u) * 0.5f;
            asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]));
          }
      }
  } else {
    for (int it = 0; it < (rm > rv ? rm : rv); ++it) {
      if (it < rm) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
      }
      if (it < rv) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 8; i += 2) {
            uint32_t h = cvt_pk(v[i], v[i + 1]);
            float ra = v[i] - __uint_as_float(h << 16), rb = v[i + 1] - __uint_as_float(h & 0xffff0000u);
            uint32_t m = cvt_pk(ra, rb);
            v[i] = ra + __uint_as_float(m << 16) * 0.5f;
            v[i + 1] = rb + __uint_as_float(m & 0xffff0000u) * 0.5f;
            asm volatile("" : "+v"(v[i]), "+v"(v[i + 1]));
          }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  for (int i = 0; i < 8; ++i) s += v[i];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
static float run(float* d, int nm, int nv) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    k<MODE><<<256, 512>>>(d, nm, nv);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  const int R = 4000;
  // one round of 16 matrix instructions = 512 clocks of the pipe; one round of 64 vector instructions = 256+ clocks
  for (int ratio = 1; ratio <= 3; ++ratio) {
    const int nm = R, nv = R * ratio;
    float tm = run<0>(d, nm, 0), tv = run<0>(d, 0, nv), tb = run<0>(d, nm, nv), ts = run<1>(d, 2 * nm, 2 * nv);
    printf("16 MFMA x %d | 64 VALU x %d:  matrix alone %.3f ms (%.0f clk/round)  vector alone %.3f ms (%.0f clk/round)  "
           "side by side on one SIMD %.3f ms  [max %.3f, sum %.3f]   both in every wavefront (same total work) %.3f ms\n",
           nm, nv, tm, tm * 2.4e6 / nm, tv, tv * 2.4e6 / nv, tb, tm > tv ? tm : tv, tm + tv, ts);
  }
  return 0;
}
