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

}  // namespace miso
