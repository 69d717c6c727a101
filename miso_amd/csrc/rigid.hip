// Per-keyframe rigid map of a sample batch: y_i = R[idx_i] x_i + t[idx_i]  (or R[idx_i]^T x_i, the cotangent
// of x).  The reference walks the keyframes in Python -- nonzero, index, matmul, index_put per keyframe, with a
// host sync each (grid_opt/loss.py:763-774, loss_isdf.py:52-61, align/miso.py:44-53; the map itself is
// transform_points_to, utils/utils_geometry.py:214-225).  A gather formulation in tensor ops moves an (N,3,3)
// temporary through HBM three times; here it is one pass: 8 B of index + 12 B in + 12 B out per row, the K
// poses (K <= a few hundred) come from L1/L2.
#include "common.hpp"

namespace miso {
namespace {

template <bool TRANSPOSE>
__global__ __launch_bounds__(256) void rigid_by_index_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                             const int64_t* __restrict__ idx,
                                                             const float* __restrict__ x, int64_t n, int32_t K,
                                                             float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int64_t k = idx[i];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);           // an index outside [0, K) would read past the pose table
  const float* r = R + k * 9;
  const float a = x[3 * i], b = x[3 * i + 1], c = x[3 * i + 2];
  float o[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    // the order of torch's row-times-matrix product: ((a r0) + b r1) + c r2, then the translation
    const float r0 = TRANSPOSE ? r[j] : r[3 * j], r1 = TRANSPOSE ? r[3 + j] : r[3 * j + 1],
                r2 = TRANSPOSE ? r[6 + j] : r[3 * j + 2];
    float s = __fmul_rn(a, r0);
    s = __fmaf_rn(b, r1, s);
    s = __fmaf_rn(c, r2, s);
    o[j] = t ? __fadd_rn(s, t[k * 3 + j]) : s;
  }
  y[3 * i] = o[0];
  y[3 * i + 1] = o[1];
  y[3 * i + 2] = o[2];
}

// One launch in front of a captured mapping step: the batch's frame ids looked up in the keyframe table
// (MisoLossMapping.world_coords: table[clamp(id)] -> pose index), the rigid map above, and the four label columns
// interleaved into the (N,4) rows the fused forward / the loss kernel read -- what the trainer otherwise does with a
// clamp, an index, rigid_by_index, and a cat (or four strided copies).
__global__ __launch_bounds__(256) void mapping_batch_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                            int32_t K, const int64_t* __restrict__ table,
                                                            int64_t table_len, const int64_t* __restrict__ frame_ids,
                                                            const float* __restrict__ x, const float* __restrict__ target,
                                                            const void* __restrict__ valid,
                                                            const float* __restrict__ sign,
                                                            const float* __restrict__ weight, int64_t n,
                                                            float* __restrict__ y, float4* __restrict__ rows,
                                                            int64_t s_target, int64_t s_valid, int64_t s_sign,
                                                            int64_t s_weight, int valid_is_bool, int sanitize) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // sanitize: torch.nan_to_num on everything read (the trainer's prepare_batch, utils.py:487-493, folded in)
  auto clean = [&](float v) -> float {
    if (!sanitize) return v;
    if (v != v) return 0.0f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
  };
  int64_t f = frame_ids[i];
  f = f < 0 ? 0 : (f >= table_len ? table_len - 1 : f);
  int64_t k = table[f];
  k = k < 0 ? 0 : (k >= K ? K - 1 : k);
  const float* r = R + k * 9;
  const float a = clean(x[3 * i]), b = clean(x[3 * i + 1]), c = clean(x[3 * i + 2]);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float s = __fmul_rn(a, r[3 * j]);
    s = __fmaf_rn(b, r[3 * j + 1], s);
    s = __fmaf_rn(c, r[3 * j + 2], s);
    y[3 * i + j] = __fadd_rn(s, t[k * 3 + j]);
  }
  float v = 1.0f;
  if (valid)
    v = valid_is_bool ? (reinterpret_cast<const unsigned char*>(valid)[i * s_valid] ? 1.0f : 0.0f)
                      : clean(reinterpret_cast<const float*>(valid)[i * s_valid]);
  rows[i] = make_float4(clean(target[i * s_target]), v, sign ? clean(sign[i * s_sign]) : 0.0f,
                        weight ? clean(weight[i * s_weight]) : 1.0f);
}

}  // namespace

hipError_t launch_mapping_batch(const float* R, const float* t, int32_t K, const int64_t* table, int64_t table_len,
                                const int64_t* frame_ids, const float* x, const float* target, const void* valid,
                                const float* sign, const float* weight, int64_t n, float* y, float* rows,
                                const int64_t* strides, int valid_is_bool, int sanitize, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int64_t one[4] = {1, 1, 1, 1};
  const int64_t* st = strides ? strides : one;
  mapping_batch_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(R, t, K, table, table_len, frame_ids, x, target,
                                                                   valid, sign, weight, n, y,
                                                                   reinterpret_cast<float4*>(rows), st[0], st[1], st[2],
                                                                   st[3], valid_is_bool, sanitize);
  return hipGetLastError();
}

hipError_t launch_rigid_by_index(const float* R, const float* t, const int64_t* idx, const float* x, int64_t n,
                                 int32_t K, int transpose, float* y, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256);
  if (transpose)
    rigid_by_index_kernel<true><<<blocks, 256, 0, s>>>(R, t, idx, x, n, K, y);
  else
    rigid_by_index_kernel<false><<<blocks, 256, 0, s>>>(R, t, idx, x, n, K, y);
  return hipGetLastError();
}

}  // namespace miso
