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

namespace miso {

struct LmK {
  const float* x;      // (N,3) samples in the keyframe frame
  const float* R;      // device, 9 floats row-major: keyframe -> submap rotation
  const float* grad;   // (N,3) d sdf / d x_submap
  const float* sdf;    // (N) predicted
  const float* gt;     // (N) measured
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
    const float r = k.sdf[i] - k.gt[i];
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
  __shared__ float red[4][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 29; ++i) {
    float v = acc[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < 29) {
    const float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (v != 0.0f) atomic_add_f32(k.out + threadIdx.x, v);
  }
}

hipError_t launch_lm_normal_eq(const float* x, const float* R, const float* grad, const float* sdf, const float* gt,
                               int64_t n, int loss_type, float gm_scale, float* out, hipStream_t s) {
  hipError_t e = launch_zero_words(out, 32, s);
  if (e != hipSuccess || n == 0) return e;
  LmK k{x, R, grad, sdf, gt, n, loss_type, gm_scale, out};
  unsigned blocks = (unsigned)((n + 255) / 256);
  if (blocks > 128u) blocks = 128u;      // <= 128 same-address atomics per sum
  lm_normal_eq_kernel<<<blocks, 256, 0, s>>>(k);
  return hipGetLastError();
}

}  // namespace miso
