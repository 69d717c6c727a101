// Mapping loss + its gradient w.r.t. the predicted SDF in one pass.
//   sdf term        grid_opt/loss.py:594-635  miso_loss_regression (L1 | L2, valid mask,
//                   per-sample weights, mean over ALL rows incl. masked ones)
//   free-space term grid_opt/loss.py:668-700  miso_loss_free_space (sign == 1 rows:
//                   max(relu(pred - bound), relu(trunc - pred)), mean over all rows)
// as combined by MisoLossMappingBase.compute (loss.py:776-806).  Replaces ~20
// elementwise launches + 2 reductions + their autograd backward by one kernel.
#include "common.hpp"

namespace miso {

// VEC: all arrays 16-B aligned and n % 4 == 0 -> one float4 per array per thread
template <bool VEC>
__global__ __launch_bounds__(256) void mapping_loss_kernel(MapLossK p, const float* __restrict__ pred,
                                                          const float* __restrict__ targ,
                                                          const float* __restrict__ valid,
                                                          const float* __restrict__ sign,
                                                          const float* __restrict__ weight, int64_t n,
                                                          float* __restrict__ gpred,
                                                          float* __restrict__ gpred_fs,
                                                          float* __restrict__ loss_out) {
  const float inv_n = 1.0f / (float)n;
  float s_sdf = 0.f, s_fs = 0.f;
  const bool want_fs = p.w_fs > 0.f && sign != nullptr;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (VEC) {
    const int64_t n4 = n / 4;
    for (int64_t i = i0; i < n4; i += stride) {
      const float4 s4 = reinterpret_cast<const float4*>(pred)[i], t4 = reinterpret_cast<const float4*>(targ)[i];
      const float4 one = make_float4(1.f, 1.f, 1.f, 1.f), zero = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 w4 = weight ? reinterpret_cast<const float4*>(weight)[i] : one;
      const float4 v4 = valid ? reinterpret_cast<const float4*>(valid)[i] : one;
      const float4 f4 = want_fs ? reinterpret_cast<const float4*>(sign)[i] : zero;
      float g[4], gf[4];
      map_loss_one(p, s4.x, t4.x, w4.x, v4.x == 1.f, f4.x == 1.f, g[0], gf[0], s_sdf, s_fs);
      map_loss_one(p, s4.y, t4.y, w4.y, v4.y == 1.f, f4.y == 1.f, g[1], gf[1], s_sdf, s_fs);
      map_loss_one(p, s4.z, t4.z, w4.z, v4.z == 1.f, f4.z == 1.f, g[2], gf[2], s_sdf, s_fs);
      map_loss_one(p, s4.w, t4.w, w4.w, v4.w == 1.f, f4.w == 1.f, g[3], gf[3], s_sdf, s_fs);
      reinterpret_cast<float4*>(gpred)[i] = make_float4((g[0] + gf[0]) * inv_n, (g[1] + gf[1]) * inv_n,
                                                        (g[2] + gf[2]) * inv_n, (g[3] + gf[3]) * inv_n);
      if (gpred_fs)   // free-space share, for callers weighting the terms apart
        reinterpret_cast<float4*>(gpred_fs)[i] = make_float4(gf[0] * inv_n, gf[1] * inv_n, gf[2] * inv_n, gf[3] * inv_n);
    }
  } else {
    for (int64_t i = i0; i < n; i += stride) {
      float g, gf;
      map_loss_one(p, pred[i], targ[i], weight ? weight[i] : 1.0f, valid ? (valid[i] == 1.0f) : true,
                   want_fs && sign[i] == 1.0f, g, gf, s_sdf, s_fs);
      gpred[i] = (g + gf) * inv_n;
      if (gpred_fs) gpred_fs[i] = gf * inv_n;
    }
  }
  // wave reduction -> block reduction in LDS -> one atomic pair per block (same-address
  // atomics serialise at ~13 ns each: 8192 of them cost 106 us, 128 cost nothing)
  for (int o = 32; o > 0; o >>= 1) { s_sdf += __shfl_down(s_sdf, o); s_fs += __shfl_down(s_fs, o); }
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s_sdf; red[1][wave] = s_fs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    float b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    atomic_add_f32(loss_out + 0, p.w_sdf * a * inv_n);
    atomic_add_f32(loss_out + 1, p.w_fs * b * inv_n);
  }
}

// The same loss over interleaved label rows {target, valid, sign, weight} (the (N,4) layout MappingStep keeps for the
// binned forward): one 16-B load per point instead of four strided column copies in front of the launch.
__global__ __launch_bounds__(256) void mapping_loss_rows_kernel(MapLossK p, const float* __restrict__ pred,
                                                               const float4* __restrict__ rows, int64_t n,
                                                               float* __restrict__ gpred, float* __restrict__ loss_out) {
  const float inv_n = 1.0f / (float)n;
  float s_sdf = 0.f, s_fs = 0.f;
  const bool want_fs = p.w_fs > 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float4 r = rows[i];
    float g, gf;
    map_loss_one(p, pred[i], r.x, r.w, r.y == 1.0f, want_fs && r.z == 1.0f, g, gf, s_sdf, s_fs);
    gpred[i] = (g + gf) * inv_n;
  }
  for (int o = 32; o > 0; o >>= 1) { s_sdf += __shfl_down(s_sdf, o); s_fs += __shfl_down(s_fs, o); }
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wave] = s_sdf; red[1][wave] = s_fs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    float b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    atomic_add_f32(loss_out + 0, p.w_sdf * a * inv_n);
    atomic_add_f32(loss_out + 1, p.w_fs * b * inv_n);
  }
}

// Small accumulators are cleared by a kernel, not by hipMemsetAsync: inside a captured HIP graph the 8-byte memset
// node in front of the atomics of mapping_loss_kernel left garbage in the two sums now and then (seen on ROCm 7.2
// with the unsorted trainer step replayed back to back: the loss read 0xFEFE.. / NaN while every input was finite).
__global__ void zero_words_kernel(uint32_t* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

// Large buffers (a level's gradient before an atomic scatter), for the same reason: 16 B per lane, grid-stride.
__global__ __launch_bounds__(256) void zero_fill_kernel(float* __restrict__ p, int64_t n) {
  const int64_t head = min(n, (int64_t)((16 - ((uintptr_t)p & 15u)) & 15u) >> 2);   // floats up to 16-B alignment
  const int64_t n4 = (n - head) >> 2;
  float4* p4 = reinterpret_cast<float4*>(p + head);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x, t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t < head) p[t] = 0.0f;
  const int64_t tail = head + 4 * n4 + t;
  if (tail < n) p[tail] = 0.0f;
}

hipError_t launch_zero_fill(float* p, int64_t n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  const int64_t want = (n / 4 + 255) / 256;
  const unsigned blocks = (unsigned)max((int64_t)1, min(want, (int64_t)256 * 16));
  zero_fill_kernel<<<blocks, 256, 0, s>>>(p, n);
  return hipGetLastError();
}

hipError_t launch_zero_words(void* p, int n_words, hipStream_t s) {
  if (n_words <= 0) return hipSuccess;
  zero_words_kernel<<<(n_words + 255) / 256, 256, 0, s>>>(reinterpret_cast<uint32_t*>(p), n_words);
  return hipGetLastError();
}

hipError_t launch_mapping_loss(int loss_type, float w_sdf, float w_fs, float trunc, const float* pred,
                               const float* targ, const float* valid, const float* sign,
                               const float* weight, int64_t n, float* gpred, float* gpred_fs,
                               float* loss_out, hipStream_t s) {
  hipError_t e = launch_zero_words(loss_out, 2, s);
  if (e != hipSuccess || n == 0) return e;
  MapLossK p{loss_type, w_sdf, w_fs, trunc};
  const uintptr_t al = (uintptr_t)pred | (uintptr_t)targ | (uintptr_t)valid | (uintptr_t)sign | (uintptr_t)weight |
                       (uintptr_t)gpred | (uintptr_t)gpred_fs;
  const bool vec = (al & 15u) == 0 && (n % 4) == 0;
  const int64_t items = vec ? n / 4 : n;
  // every block ends with one atomic pair on the same two addresses (~13 ns each, serialised):
  // 64 blocks keep that tail under 2 us while 16 K threads still cover the 6 MB of streams
  unsigned blocks = (unsigned)((items + 255) / 256);
  if (blocks > 64u) blocks = 64u;
  if (vec) mapping_loss_kernel<true><<<blocks, 256, 0, s>>>(p, pred, targ, valid, sign, weight, n, gpred, gpred_fs, loss_out);
  else mapping_loss_kernel<false><<<blocks, 256, 0, s>>>(p, pred, targ, valid, sign, weight, n, gpred, gpred_fs, loss_out);
  return hipGetLastError();
}

hipError_t launch_mapping_loss_rows(int loss_type, float w_sdf, float w_fs, float trunc, const float* pred,
                                    const float* rows, int64_t n, float* gpred, float* loss_out, hipStream_t s) {
  hipError_t e = launch_zero_words(loss_out, 2, s);
  if (e != hipSuccess || n == 0) return e;
  MapLossK p{loss_type, w_sdf, w_fs, trunc};
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 64u) blocks = 64u;
  mapping_loss_rows_kernel<<<blocks, 256, 0, s>>>(p, pred, reinterpret_cast<const float4*>(rows), n, gpred, loss_out);
  return hipGetLastError();
}

}  // namespace miso
