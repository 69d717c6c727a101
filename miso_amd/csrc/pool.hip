// utils.grid_pool_3d_avg (grid_opt/utils/utils.py:239-291): the features of the points that fall into each cell of a regular
// grid, averaged -- what the learned initialisation pools its residual signals with (models/encoder.py:
// compute_encoder_inputs_from_residuals).  The reference forms three index tensors, two zero grids, two scatter_add_ and a
// division (~15 launches, an (N, d) expanded index tensor); here: clear, scatter (one float atomic per point and channel,
// one integer atomic per point), normalise.  Cell index op for op as the reference: ((p - bound_min) / cell_size) truncated
// like .long(), clamped to [0, size - 1]; linear index (ix ny + iy) nz + iz.
#include "common.hpp"

namespace miso {
namespace {

__global__ __launch_bounds__(256) void pool_clear_kernel(float* __restrict__ acc, int32_t* __restrict__ cnt, int64_t cells,
                                                         int32_t d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cells * d) acc[i] = 0.0f;
  if (i < cells) cnt[i] = 0;
}

__global__ __launch_bounds__(256) void pool_scatter_kernel(const float* __restrict__ coords, const float* __restrict__ feat,
                                                           int64_t n, int32_t d, int64_t ld, float b0, float b1, float b2,
                                                           float cell, int32_t nx, int32_t ny, int32_t nz,
                                                           float* __restrict__ acc, int32_t* __restrict__ cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float p[3] = {coords[3 * i], coords[3 * i + 1], coords[3 * i + 2]};
  const float bm[3] = {b0, b1, b2};
  const int size[3] = {nx, ny, nz};
  int idx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float q = __fdiv_rn(__fsub_rn(p[a], bm[a]), cell);
    // .long(): truncation towards zero; then clamp(0, size - 1) (a NaN coordinate lands in cell 0)
    float t = truncf(q);
    t = (t != t) ? 0.0f : fminf(fmaxf(t, 0.0f), (float)(size[a] - 1));
    idx[a] = (int)t;
  }
  const int64_t lin = ((int64_t)idx[0] * ny + idx[1]) * nz + idx[2];
  for (int c = 0; c < d; ++c) atomic_add_f32(acc + lin * d + c, feat[i * ld + c]);
  atomicAdd(cnt + lin, 1);
}

__global__ __launch_bounds__(256) void pool_normalise_kernel(float* __restrict__ acc, const int32_t* __restrict__ cnt,
                                                             int64_t cells, int32_t d) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= cells * d) return;
  const int32_t c = cnt[i / d];
  acc[i] = __fdiv_rn(acc[i], (float)(c < 1 ? 1 : c));      // G /= counts.clamp(min=1)
}

}  // namespace

hipError_t launch_grid_pool_avg(const float* coords, const float* feat, int64_t n, int32_t d, int64_t ld, const float* bmin,
                                float cell, int32_t nx, int32_t ny, int32_t nz, float* acc, int32_t* cnt, hipStream_t s) {
  const int64_t cells = (int64_t)nx * ny * nz;
  const int64_t tot = cells * d;
  pool_clear_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(acc, cnt, cells, d);
  if (n > 0)
    pool_scatter_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(coords, feat, n, d, ld, bmin[0], bmin[1], bmin[2], cell,
                                                                   nx, ny, nz, acc, cnt);
  pool_normalise_kernel<<<(unsigned)((tot + 255) / 256), 256, 0, s>>>(acc, cnt, cells, d);
  return hipGetLastError();
}

}  // namespace miso
