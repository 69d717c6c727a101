// Micro-benchmark / layout check for miso_amd/csrc/mlp_split.hpp (round 6): one 64x64 decoder layer on 64 points,
// (a) bf16x3 split products on v_mfma_f32_32x32x16_bf16, (b) the exact fp32 chain on v_mfma_f32_32x32x2_f32,
// both against float64 on the host; then the feature-row (layer 0) operand path; then clocks per layer and wavefront
// with eight wavefronts per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/mlp_split.hip -o tools/ubench/mlp_split && tools/ubench/mlp_split
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../../miso_amd/csrc/mlp_split.hpp"

using namespace miso;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int H = 64, NP = 64, F = 24;

__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

template <bool RELU_LOOP>
__global__ __launch_bounds__(512, 1) void k_split(const uint32_t* __restrict__ apack, const float* __restrict__ X,
                                                  const float* __restrict__ bias, float* __restrict__ Y, int iters) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int nd = split_matrix_dwords(4, 2);
  for (int i = threadIdx.x * 4; i < nd; i += blockDim.x * 4)
    *reinterpret_cast<u32x4*>(lds + i) = *reinterpret_cast<const u32x4*>(apack + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  f32x16 in[2][2];
#pragma unroll
  for (int rp = 0; rp < 2; ++rp)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) in[rp][t][j] = X[(32 * t + (lane & 31)) * H + 32 * rp + row_of(j, hi)];
  f32x16 init[2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) init[r][j] = bias[32 * r + row_of(j, hi)];
  f32x16 out[2][2];
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
    Split3 B[4][2];
#pragma unroll
    for (int rp = 0; rp < 2; ++rp)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        B[2 * rp][t] = split_acc<0>(in[rp][t]);
        B[2 * rp + 1][t] = split_acc<1>(in[rp][t]);
      }
    mma_split<4, 2, 2>(lds, lane, B, out, init);
    if (RELU_LOOP) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j) in[r][t][j] = relu1(out[r][t][j]) * 0.25f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[(32 * t + (lane & 31)) * H + 32 * r + row_of(j, hi)] = out[r][t][j];
  }
}

template <bool RELU_LOOP>
__global__ __launch_bounds__(512, 1) void k_exact(const float* __restrict__ wp, const float* __restrict__ X,
                                                  const float* __restrict__ bias, float* __restrict__ Y, int iters) {
  // wp: [ks][64 lanes][RT] as sdf_fused.hip packs it
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  float* w = reinterpret_cast<float*>(lds);
  for (int i = threadIdx.x; i < 32 * 64 * 2; i += blockDim.x) w[i] = wp[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  f32x16 in[2][2];
#pragma unroll
  for (int rp = 0; rp < 2; ++rp)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) in[rp][t][j] = X[(32 * t + (lane & 31)) * H + 32 * rp + row_of(j, hi)];
  f32x16 init[2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) init[r][j] = bias[32 * r + row_of(j, hi)];
  f32x16 out[2][2];
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int rp = 0; rp < 2; ++rp)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int ks = rp * 16 + j;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const float a = w[(ks * 64 + lane) * 2 + r];
          out[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[rp][0][j], ks == 0 ? init[r] : out[r][0], 0, 0, 0);
          out[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, in[rp][1][j], ks == 0 ? init[r] : out[r][1], 0, 0, 0);
        }
      }
    if (RELU_LOOP) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j) in[r][t][j] = relu1(out[r][t][j]) * 0.25f;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < 64) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[(32 * t + (lane & 31)) * H + 32 * r + row_of(j, hi)] = out[r][t][j];
  }
}

// layer 0: lane = point, F = 24 features per lane -> two k-blocks (the second half empty), permlane32_swap of the pieces
__global__ __launch_bounds__(64) void k_split0(const uint32_t* __restrict__ apack, const float* __restrict__ X0,
                                               const float* __restrict__ bias, float* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int nd = split_matrix_dwords(2, 2);
  for (int i = threadIdx.x * 4; i < nd; i += blockDim.x * 4)
    *reinterpret_cast<u32x4*>(lds + i) = *reinterpret_cast<const u32x4*>(apack + i);
  __syncthreads();
  const int lane = threadIdx.x & 63, hi = lane >> 5;
  float f[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) f[i] = i < F ? X0[lane * F + i] : 0.0f;
  Split3 B[2][2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
    Split3 lo = split8(f[16 * kb + 0], f[16 * kb + 1], f[16 * kb + 2], f[16 * kb + 3], f[16 * kb + 4], f[16 * kb + 5],
                       f[16 * kb + 6], f[16 * kb + 7]);
    Split3 up = split8(f[16 * kb + 8], f[16 * kb + 9], f[16 * kb + 10], f[16 * kb + 11], f[16 * kb + 12], f[16 * kb + 13],
                       f[16 * kb + 14], f[16 * kb + 15]);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        auto sw = __builtin_amdgcn_permlane32_swap(lo.q[q][d], up.q[q][d], false, false);
        B[kb][0].q[q][d] = sw[0];
        B[kb][1].q[q][d] = sw[1];
      }
  }
  f32x16 init[2], out[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int j = 0; j < 16; ++j) init[r][j] = bias[32 * r + row_of(j, hi)];
  mma_split<2, 2, 2>(lds, lane, B, out, init);
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) Y[(32 * t + (lane & 31)) * H + 32 * r + row_of(j, hi)] = out[r][t][j];
}

static double urand() { return (double)rand() / RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-12)) * cos(6.283185307179586 * urand()); }

static void report(const char* name, const std::vector<float>& y, const std::vector<double>& ref, const std::vector<double>& mag) {
  double mx = 0, mean = 0, mxrel = 0;
  for (size_t i = 0; i < y.size(); ++i) {
    const double e = fabs((double)y[i] - ref[i]);
    mx = fmax(mx, e); mean += e; mxrel = fmax(mxrel, e / mag[i]);
  }
  printf("%-28s max |err| %.3e  mean |err| %.3e  max |err| / sum|a b| %.3e (2^%.1f)\n", name, mx, mean / y.size(), mxrel,
         log2(mxrel));
}

int main() {
  srand(1234);
  std::vector<float> W(H * H), X(NP * H), b(H), W0(H * F), X0(NP * F);
  for (auto& v : W) v = (float)(nrand() / 8.0);
  for (auto& v : W0) v = (float)(nrand() / 4.0);
  for (auto& v : b) v = (float)(nrand() * 0.1);
  for (auto& v : X) { double z = nrand() * 3.0; v = z > 0 ? (float)z : 0.0f; }
  for (auto& v : X0) v = (float)(nrand() * 0.5);
  // float64 references
  std::vector<double> ref(NP * H), mag(NP * H), ref0(NP * H), mag0(NP * H);
  for (int p = 0; p < NP; ++p)
    for (int m = 0; m < H; ++m) {
      double s = b[m], a = fabs((double)b[m]);
      for (int k = 0; k < H; ++k) { s += (double)W[m * H + k] * X[p * H + k]; a += fabs((double)W[m * H + k] * X[p * H + k]); }
      ref[p * H + m] = s; mag[p * H + m] = a;
      s = b[m]; a = fabs((double)b[m]);
      for (int k = 0; k < F; ++k) { s += (double)W0[m * F + k] * X0[p * F + k]; a += fabs((double)W0[m * F + k] * X0[p * F + k]); }
      ref0[p * H + m] = s; mag0[p * H + m] = a;
    }
  // packs
  std::vector<uint32_t> ap(split_matrix_dwords(4, 2), 0), ap0(split_matrix_dwords(2, 2), 0);
  for (int kb = 0; kb < 4; ++kb)
    for (int r = 0; r < 2; ++r)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          uint32_t pc[3];
          bf16_split3(W[(32 * r + (lane & 31)) * H + split_k_acc(kb, lane >> 5, i)], pc);
          for (int q = 0; q < 3; ++q) ap[split_a_dword(kb, r, q, lane, 2) + i / 2] |= pc[q] << (16 * (i & 1));
        }
  for (int kb = 0; kb < 2; ++kb)
    for (int r = 0; r < 2; ++r)
      for (int lane = 0; lane < 64; ++lane)
        for (int i = 0; i < 8; ++i) {
          const int k = split_k_feat(kb, lane >> 5, i);
          uint32_t pc[3];
          bf16_split3(k < F ? W0[(32 * r + (lane & 31)) * F + k] : 0.0f, pc);
          for (int q = 0; q < 3; ++q) ap0[split_a_dword(kb, r, q, lane, 2) + i / 2] |= pc[q] << (16 * (i & 1));
        }
  std::vector<float> wp(32 * 64 * 2);
  for (int ks = 0; ks < 32; ++ks)
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 2; ++r)
        wp[(ks * 64 + lane) * 2 + r] = W[(32 * r + (lane & 31)) * H + 32 * (ks / 16) + row_of(ks % 16, lane >> 5)];

  uint32_t *d_ap, *d_ap0; float *d_wp, *d_X, *d_X0, *d_b, *d_Y;
  CK(hipMalloc(&d_ap, ap.size() * 4)); CK(hipMalloc(&d_ap0, ap0.size() * 4)); CK(hipMalloc(&d_wp, wp.size() * 4));
  CK(hipMalloc(&d_X, X.size() * 4)); CK(hipMalloc(&d_X0, X0.size() * 4)); CK(hipMalloc(&d_b, b.size() * 4));
  CK(hipMalloc(&d_Y, NP * H * 4));
  CK(hipMemcpy(d_ap, ap.data(), ap.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_ap0, ap0.data(), ap0.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wp, wp.data(), wp.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_X0, X0.data(), X0.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> y(NP * H);

  const size_t lds_s = ap.size() * 4, lds_e = wp.size() * 4, lds_0 = ap0.size() * 4;
  CK(hipFuncSetAttribute((const void*)k_split<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
  CK(hipFuncSetAttribute((const void*)k_split<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
  k_split<false><<<1, 64, lds_s>>>(d_ap, d_X, d_b, d_Y, 1);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), d_Y, y.size() * 4, hipMemcpyDeviceToHost));
  report("hidden layer, bf16x3 split", y, ref, mag);
  k_exact<false><<<1, 64, lds_e>>>(d_wp, d_X, d_b, d_Y, 1);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), d_Y, y.size() * 4, hipMemcpyDeviceToHost));
  report("hidden layer, exact fp32", y, ref, mag);
  k_split0<<<1, 64, lds_0>>>(d_ap0, d_X0, d_b, d_Y);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(y.data(), d_Y, y.size() * 4, hipMemcpyDeviceToHost));
  report("layer 0 (F = 24), bf16x3", y, ref0, mag0);

  // throughput: 256 workgroups of eight wavefronts (two per SIMD), `iters` layers each
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      if (mode == 0) k_split<true><<<256, 512, lds_s>>>(d_ap, d_X, d_b, d_Y, iters);
      else k_exact<true><<<256, 512, lds_e>>>(d_wp, d_X, d_b, d_Y, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep == 2) {
        // two wavefronts per SIMD -> per SIMD 2 * iters layers
        const double ns_layer = ms * 1e6 / (2.0 * iters);
        printf("%s: %.3f ms, %.1f ns per (layer, 64 points) per SIMD = %.0f clocks at 2.4 GHz; %.1f TFLOP/s nominal\n",
               mode == 0 ? "bf16x3 split + relu" : "exact fp32 + relu ", ms, ns_layer, ns_layer * 2.4,
               2.0 * 64 * 64 * 64 * 2.0 * iters * 1024 / (ms * 1e-3) / 1e12);
      }
    }
  }
  return 0;
}
