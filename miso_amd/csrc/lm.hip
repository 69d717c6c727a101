// Damped Gauss-Newton normal equations of the keyframe tracker in one pass.
//
// Reference: Tracker.lm_step, grid_opt/slam/tracker.py:148-212 -- after one SDF forward and one
// coordinate backward it builds, with ~10 separate tensor ops, for every sample i of the keyframe
//     c_i = hat(R x_i) grad_i = (R x_i) x grad_i          (tracker.py:182-184, pytorch3d hat)
//     J_i = [ c_i^T R , grad_i^T ]                         (1 x 6: rotation | translation)
//     r_i = sdf_i - gt_i,   w_i = 1 (L2)  |  c / (c + r_i^2)^2 (GM, tracker.py:139-146)
// and the dense products H = J^T diag(w) J, g = J^T diag(w) r.  Here: one launch, 21 + 6 + 2
// block-reduced sums (upper triangle of H, g, sum w r^2, count).  lambda*I and the 6x6 solve
// stay with the caller (they are 36 floats).
#include "common.hpp"
#include "align.hpp"

namespace miso {

struct LmK {
  const float* x;      // (N,3) samples in the keyframe frame
  const float* R;      // device, 9 floats row-major: keyframe -> submap rotation
  const float* grad;   // (N,3) d sdf / d x_submap
  const float* sdf;    // (N) predicted
  const float* gt;     // (N) measured, element stride s_gt
  int64_t s_gt;
  float trunc;         // >= 0: only rows with |gt| < trunc count (the tracker's SDF truncation filter); < 0: all rows
  int64_t n;
  int loss_type;       // 2 = L2, 3 = GM
  float gm_scale;
  float* out;          // 32 floats: H upper triangle row-major [0,21), g [21,27), sum w r^2 [27], n [28]
};

__global__ __launch_bounds__(256) void lm_normal_eq_kernel(LmK k) {
  float R[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = k.R[i];
  float acc[29];
#pragma unroll
  for (int i = 0; i < 29; ++i) acc[i] = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float x0 = k.x[i * 3], x1 = k.x[i * 3 + 1], x2 = k.x[i * 3 + 2];
    const float g0 = k.grad[i * 3], g1 = k.grad[i * 3 + 1], g2 = k.grad[i * 3 + 2];
    const float y0 = R[0] * x0 + R[1] * x1 + R[2] * x2, y1 = R[3] * x0 + R[4] * x1 + R[5] * x2,
                y2 = R[6] * x0 + R[7] * x1 + R[8] * x2;
    const float c0 = y1 * g2 - y2 * g1, c1 = y2 * g0 - y0 * g2, c2 = y0 * g1 - y1 * g0;   // (R x) x grad
    float J[6];
    J[0] = c0 * R[0] + c1 * R[3] + c2 * R[6];       // c^T R
    J[1] = c0 * R[1] + c1 * R[4] + c2 * R[7];
    J[2] = c0 * R[2] + c1 * R[5] + c2 * R[8];
    J[3] = g0; J[4] = g1; J[5] = g2;
    const float gti = k.gt[i * k.s_gt];
    if (k.trunc >= 0.0f && !(fabsf(gti) < k.trunc)) continue;
    const float r = k.sdf[i] - gti;
    float w = 1.0f;
    if (k.loss_type == 3) { const float d = k.gm_scale + r * r; w = k.gm_scale / (d * d); }
    int o = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      const float wa = w * J[a];
#pragma unroll
      for (int b = a; b < 6; ++b) acc[o++] += wa * J[b];
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) acc[21 + a] += w * J[a] * r;
    acc[27] += w * r * r;
    acc[28] += 1.0f;
  }
  // the fan-in of a workgroup (64 lanes, 4 waves) in double; what crosses workgroups is at most 128 fp32 partials per
  // sum (the outputs stay the 32 floats the 6x6 fp32 solve reads, as torch.linalg.solve does upstream)
  __shared__ double red[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 29; ++i) {
    double v = (double)acc[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    const float v = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    if (v != 0.0f) atomic_add_f32(k.out + threadIdx.x, v);
  }
}

hipError_t launch_lm_normal_eq(const float* x, const float* R, const float* grad, const float* sdf, const float* gt,
                               int64_t n, int loss_type, float gm_scale, float* out, hipStream_t s) {
  hipError_t e = launch_zero_words(out, 32, s);
  if (e != hipSuccess || n == 0) return e;
  LmK k{x, R, grad, sdf, gt, 1, -1.0f, n, loss_type, gm_scale, out};
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 128u) blocks = 128u;      // <= 128 same-address atomics per sum
  lm_normal_eq_kernel<<<blocks, 256, 0, s>>>(k);
  return hipGetLastError();
}

// ---- one whole Levenberg-Marquardt step of the tracker on the device (Tracker.lm_step, tracker.py:148-212) -----------
// pose of the keyframe from its corrections -> samples into the submap frame (+ the bookkeeping the reference does
// with host round trips: the truncation filter, the frame-id / validity asserts, the field-of-view overlap) -> [SDF
// and its spatial gradient: the fused forward / coordinate backward, launched by the caller between the two halves]
// -> normal equations -> damped 6x6 solve -> pose corrections updated in place -> eight floats for the host.
__global__ void lm_pose_kernel(LmTrackK k) {
  const int t = threadIdx.x;
  if (t < 36) k.sums[t] = 0.0f;
  if (t == 0) {
    const float w[3] = {k.dr[0], k.dr[1], k.dr[2]};
    float E[9];
    so3_exp(w, E);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        k.pose[i * 3 + j] = (k.Rwk[i * 3] * E[j] + k.Rwk[i * 3 + 1] * E[3 + j]) + k.Rwk[i * 3 + 2] * E[6 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) k.pose[9 + i] = k.twk[i] + k.dt[i];
  }
}

__global__ __launch_bounds__(256) void lm_transform_kernel(LmTrackK k) {
  float R[9], t[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) R[i] = k.pose[i];
#pragma unroll
  for (int i = 0; i < 3; ++i) t[i] = k.pose[9 + i];
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  // clean != nullptr: torch.nan_to_num of what is read (prepare_batch, utils.py:487-493, folded in), kept for the
  // kernels after this one
  auto clean = [&](float v) -> float {
    if (!k.clean) return v;
    if (v != v) return 0.0f;
    return fminf(fmaxf(v, -3.4028234663852886e38f), 3.4028234663852886e38f);
  };
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float a = clean(k.x[3 * i]), b = clean(k.x[3 * i + 1]), cc = clean(k.x[3 * i + 2]);
    float y[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {      // transform_points_to: the row-times-matrix order of rigid.hip
      float s = __fmul_rn(a, R[3 * j]);
      s = __fmaf_rn(b, R[3 * j + 1], s);
      s = __fmaf_rn(cc, R[3 * j + 2], s);
      y[j] = __fadd_rn(s, t[j]);
      k.xw[3 * i + j] = y[j];
    }
    const float g = clean(k.gt[i * k.s_gt]);
    bool ok = true;
    if (k.valid)
      ok = k.valid_is_bool ? reinterpret_cast<const unsigned char*>(k.valid)[i * k.s_valid] != 0
                           : clean(reinterpret_cast<const float*>(k.valid)[i * k.s_valid]) == 1.0f;
    if (k.clean) {
      k.clean[3 * i] = a; k.clean[3 * i + 1] = b; k.clean[3 * i + 2] = cc;
      k.clean[3 * k.n + i] = g;
      k.clean[4 * k.n + i] = ok ? 1.0f : 0.0f;
    }
    if (k.trunc >= 0.0f && !(fabsf(g) < k.trunc)) continue;
    c[0] += 1.0f;
    if (y[0] >= k.bmin[0] && y[0] <= k.bmax[0] && y[1] >= k.bmin[1] && y[1] <= k.bmax[1] && y[2] >= k.bmin[2] &&
        y[2] <= k.bmax[2])
      c[1] += 1.0f;
    if (k.frame_ids && k.frame_ids[i * k.s_fid] != k.kf) c[2] += 1.0f;
    if (!ok) c[3] += 1.0f;
  }
  __shared__ float red[4][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = c[q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][q] = v;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0.0f) atomic_add_f32(k.sums + 32 + threadIdx.x, v);      // integer-valued: exact below 2^24
  }
}

// H delta = -g with H = J^T W J + lambda I: LU with partial pivoting in fp32 (what torch.linalg.solve does for a
// general 6x6), then the corrections move in place and the host gets its eight numbers.
__global__ void lm_solve_kernel(LmTrackK k) {
  if (threadIdx.x != 0) return;
  float A[6][7];
  int o = 0;
  for (int a = 0; a < 6; ++a)
    for (int b = a; b < 6; ++b) { A[a][b] = k.sums[o]; A[b][a] = k.sums[o]; ++o; }
  float gn = 0.0f;
  for (int a = 0; a < 6; ++a) {
    A[a][a] += k.lm_lambda;
    A[a][6] = -k.sums[21 + a];
    gn += k.sums[21 + a] * k.sums[21 + a];
  }
  for (int c = 0; c < 6; ++c) {
    int p = c;
    for (int r = c + 1; r < 6; ++r)
      if (fabsf(A[r][c]) > fabsf(A[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 7; ++j) { const float tmp = A[c][j]; A[c][j] = A[p][j]; A[p][j] = tmp; }
    const float inv = 1.0f / A[c][c];
    for (int r = c + 1; r < 6; ++r) {
      const float f = A[r][c] * inv;
      for (int j = c; j < 7; ++j) A[r][j] -= f * A[c][j];
    }
  }
  float d[6];
  for (int r = 5; r >= 0; --r) {
    float v = A[r][6];
    for (int j = r + 1; j < 6; ++j) v -= A[r][j] * d[j];
    d[r] = v / A[r][r];
  }
  // (the reference asserts on frame ids and validity before it touches the pose: leave it alone if they fail)
  if (k.sums[34] == 0.0f && k.sums[35] == 0.0f)
    for (int i = 0; i < 3; ++i) { k.dr[i] += d[i]; k.dt[i] += d[3 + i]; }
  k.info[0] = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
  k.info[1] = sqrtf((d[3] * d[3] + d[4] * d[4]) + d[5] * d[5]);
  k.info[2] = sqrtf(gn);
  k.info[3] = k.sums[33];
  k.info[4] = k.sums[32];
  k.info[5] = k.sums[34];
  k.info[6] = k.sums[35];
  k.info[7] = 0.0f;
}

// ---- the tracker's Adam solver (Tracker.track_window with MisoLossTracking, tracker.py:95-118 + loss.py:517-586 + the
// Trainer step) as a launch sequence without a host round trip: pose -> samples into the submap frame -> fused forward
// -> residual / loss / d loss / d sdf -> fused coordinate backward -> pose cotangents -> so3_exp backward + Adam on the
// six numbers of the keyframe's corrections.  sums: [0,9) d loss / d R, [9,12) d loss / d t, [12] loss sum.
__global__ __launch_bounds__(256) void track_loss_kernel(TrackAdamK k) {
  const LmTrackK& s = k.s;
  const float inv_n = 1.0f / (float)s.n;
  float acc = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g = s.gt[i * s.s_gt];
    bool ok = true;
    if (s.valid)
      ok = s.valid_is_bool ? reinterpret_cast<const unsigned char*>(s.valid)[i * s.s_valid] != 0
                           : reinterpret_cast<const float*>(s.valid)[i * s.s_valid] == 1.0f;
    if (s.trunc >= 0.0f) ok = ok && (fabsf(g) < s.trunc);
    const float r = ok ? k.sdf[i] - g : 0.0f;        // torch.where(valid == 1, pred - gt, 0)
    float d;
    if (k.loss_type == 2) { acc += r * r; d = 2.0f * r; }
    else if (k.loss_type == 1) { acc += fabsf(r); d = (r > 0.f) ? 1.f : ((r < 0.f) ? -1.f : 0.f); }
    else { const float q = k.gm_scale + r * r, w = k.gm_scale / (q * q); acc += w * r * r; d = w * 2.0f * r; }
    k.gpred[i] = ok ? k.weight_sdf * d * inv_n : 0.0f;
  }
  double accd = (double)acc;
  for (int o = 32; o > 0; o >>= 1) accd += __shfl_down(accd, o);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = accd;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (float)((red[0] + red[1]) + (red[2] + red[3]));
    if (v != 0.0f || v != v) atomic_add_f32(s.sums + 12, k.weight_sdf * v * inv_n);
  }
}

__global__ __launch_bounds__(256) void track_reduce_kernel(TrackAdamK k) {
  const LmTrackK& s = k.s;
  float acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < s.n; i += (int64_t)gridDim.x * blockDim.x) {
    const float g0 = k.gx[3 * i], g1 = k.gx[3 * i + 1], g2 = k.gx[3 * i + 2];
    const float x0 = s.x[3 * i], x1 = s.x[3 * i + 1], x2 = s.x[3 * i + 2];
    acc[0] += g0 * x0; acc[1] += g0 * x1; acc[2] += g0 * x2;      // d loss / d R = sum g x^T   (y = R x + t)
    acc[3] += g1 * x0; acc[4] += g1 * x1; acc[5] += g1 * x2;
    acc[6] += g2 * x0; acc[7] += g2 * x1; acc[8] += g2 * x2;
    acc[9] += g0; acc[10] += g1; acc[11] += g2;
  }
  __shared__ double red[4][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    double v = (double)acc[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 12) {
    const float v = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
    if (v != 0.0f || v != v) atomic_add_f32(s.sums + threadIdx.x, v);
  }
}

__global__ void track_adam_kernel(TrackAdamK k) {
  if (threadIdx.x != 0) return;
  const LmTrackK& s = k.s;
  int32_t* cnt = reinterpret_cast<int32_t*>(k.state + 12);      // {steps taken, skipped, iterations}
  const int it = cnt[2];
  const float loss = s.sums[12];
  if (it < k.ring_len) k.ring[it] = loss;
  cnt[2] = it + 1;
  s.info[0] = loss;
  if (!(loss == loss)) { cnt[1] += 1; return; }                  // "Loss is nan! Skip backward step."
  // R = Rwk E(dr):  d loss / d E = Rwk^T (d loss / d R), then through the exponential map (as the alignment epilogue)
  double G[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      G[i * 3 + j] = (double)s.Rwk[i] * s.sums[j] + (double)s.Rwk[3 + i] * s.sums[3 + j] + (double)s.Rwk[6 + i] * s.sums[6 + j];
  const float w[3] = {s.dr[0], s.dr[1], s.dr[2]};
  double gw[3];
  so3_exp_backward(w, G, gw);
  const float g6[6] = {(float)gw[0], (float)gw[1], (float)gw[2], s.sums[9], s.sums[10], s.sums[11]};
  const int t = cnt[0] + 1;
  cnt[0] = t;
  const AdamScalars a = k.table[min(t, k.table_len) - 1];
  float* m = k.state;
  float* v = k.state + 6;
  for (int i = 0; i < 3; ++i) {
    adam_one(s.dr[i], g6[i], m[i], v[i], a);
    adam_one(s.dt[i], g6[3 + i], m[3 + i], v[3 + i], a);
  }
}

hipError_t launch_track_loss(const TrackAdamK& k, hipStream_t st) {
  if (k.s.n <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((k.s.n + 255) / 256);
  if (blocks > 128u) blocks = 128u;
  track_loss_kernel<<<blocks, 256, 0, st>>>(k);
  return hipGetLastError();
}

hipError_t launch_track_tail(const TrackAdamK& k, hipStream_t st) {
  if (k.s.n > 0) {
    unsigned blocks = (unsigned)((k.s.n + 255) / 256);
    if (blocks > 128u) blocks = 128u;
    track_reduce_kernel<<<blocks, 256, 0, st>>>(k);
  }
  track_adam_kernel<<<1, 64, 0, st>>>(k);
  return hipGetLastError();
}

hipError_t launch_lm_track_head(const LmTrackK& k, hipStream_t s) {
  lm_pose_kernel<<<1, 64, 0, s>>>(k);
  if (k.n > 0) {
    unsigned blocks = (unsigned)((k.n + 255) / 256);
    if (blocks > 128u) blocks = 128u;
    lm_transform_kernel<<<blocks, 256, 0, s>>>(k);
  }
  return hipGetLastError();
}

hipError_t launch_lm_track_tail(const LmTrackK& k, const float* grad, const float* sdf, int loss_type, float gm_scale,
                                hipStream_t s) {
  if (k.n > 0) {
    LmK q{k.x, k.pose, grad, sdf, k.gt, k.s_gt, k.trunc, k.n, loss_type, gm_scale, k.sums};
    unsigned blocks = (unsigned)((k.n + 255) / 256);
    if (blocks > 128u) blocks = 128u;
    lm_normal_eq_kernel<<<blocks, 256, 0, s>>>(q);
  }
  lm_solve_kernel<<<1, 64, 0, s>>>(k);
  return hipGetLastError();
}

}  // namespace miso
