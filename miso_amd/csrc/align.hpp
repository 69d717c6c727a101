// Kernel-side structs of the fused alignment iteration (align.hip, capi.hip).
#pragma once
#include "common.hpp"

namespace miso {

struct AlignLayout {
  int64_t params, pose, out, cnt, pair_loss, flat, adam_m, adam_v, adam_t, ctrl, ring, ring_row, order, total;
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

// d loss / d w for R = R0 Exp(w), given gR = d loss / d R: G = R0^T gR, Exp = I + a K + b K^2 with
// a = sin(th)/th, b = (1 - cos th)/th^2, th = sqrt(max(|w|^2, eps)) -- the clamp passes no gradient below eps
// (torch.clamp backward), so near w = 0 only the a dK and b d(K^2) terms remain.  In double: three values per
// submap per iteration, and the a', b' differences cancel badly in fp32.
__device__ __forceinline__ void so3_exp_backward(const float w_[3], const double G[9], double gw[3]) {
  const double w[3] = {w_[0], w_[1], w_[2]};
  const float sqf = (w_[0] * w_[0] + w_[1] * w_[1]) + w_[2] * w_[2];
  const bool live = sqf >= 1e-4f;
  const double th = sqrt(fmax((double)sqf, 1e-4));
  const double s = sin(th), c = cos(th);
  const double a = s / th, b = (1.0 - c) / (th * th);
  const double K[9] = {0., -w[2], w[1], w[2], 0., -w[0], -w[1], w[0], 0.};
  double KK[9], M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) KK[i * 3 + j] = K[i * 3] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
  // M = G K^T + K^T G
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double v = 0.;
      for (int q = 0; q < 3; ++q) v += G[i * 3 + q] * K[j * 3 + q] + K[q * 3 + i] * G[q * 3 + j];
      M[i * 3 + j] = v;
    }
  double gk = 0., gkk = 0.;
  for (int i = 0; i < 9; ++i) { gk += G[i] * K[i]; gkk += G[i] * KK[i]; }
  const double da = live ? (th * c - s) / (th * th) / th : 0.;                     // a'(th) / th
  const double db = live ? (th * s - 2.0 * (1.0 - c)) / (th * th * th) / th : 0.;  // b'(th) / th
  const double veeG[3] = {G[7] - G[5], G[2] - G[6], G[3] - G[1]};
  const double veeM[3] = {M[7] - M[5], M[2] - M[6], M[3] - M[1]};
  for (int q = 0; q < 3; ++q) gw[q] = a * veeG[q] + b * veeM[q] + (da * gk + db * gkk) * w[q];
}

}  // namespace miso
