// Kernel-side structs of the fused alignment iteration (align.hip, capi.hip).
#pragma once
#include "common.hpp"

namespace miso {

struct AlignLayout {
  int64_t params, pose, out, cnt, pair_loss, flat, adam_m, adam_v, ctrl, ring, ring_row, total;
};

__host__ __device__ inline int64_t up4(int64_t v) { return (v + 3) / 4 * 4; }

AlignLayout align_layout(int S, int P, int ring_iters, int save_poses);

enum { CTRL_STEP = 0, CTRL_STOPPED = 1, CTRL_ITER = 2, CTRL_SKIPPED = 3 };

struct AlignK {
  int S, P, loss_type, ring_iters, save_poses;
  float align_weight, overlap_thresh, reg_weight, reg_rad, reg_m, rel_thresh;
  double lr, b1, b2, eps;
  const float* R0;
  const float* t0;
  const AlignPairK* plan;
  float* state;
  AlignLayout L;
};

// so3_exp_map(log_rot, eps = 1e-4) of pytorch3d as restated in miso_amd/so3.py, fp32, same op order
__device__ __forceinline__ void so3_exp(const float w[3], float E[9]) {
#pragma clang fp contract(off)
  const float sq = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  const float ang = sqrtf(fmaxf(sq, 1e-4f));
  const float a = sinf(ang) / ang, b = (1.0f - cosf(ang)) / (ang * ang);
  const float K[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  float KK[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) KK[i * 3 + j] = (K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j]) + K[i * 3 + 2] * K[6 + j];
#pragma unroll
  for (int i = 0; i < 9; ++i) E[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + a * K[i] + b * KK[i];
}

}  // namespace miso
