// Dense Adam step over one tensor (torch.optim.Adam semantics, amsgrad=False,
// weight_decay=0, maximize=False), as driven by grid_opt/trainer.py:196-228.
// Dense on purpose: moments decay and parameters keep moving for voxels the
// batch did not touch, exactly like the reference.  HBM-bound: 16 B read +
// 12 B written per element (+4 B when the gradient is cleared in the same pass).
#include <math.h>

#include "common.hpp"

namespace miso {

struct AdamScalars {
  float one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& a) {
  m = m + a.one_minus_b1 * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (a.one_minus_b2 * g) * g;          // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom = sqrtf(v) / a.bc2_sqrt + a.eps;      // (sqrt / bias_correction2_sqrt).add_(eps)
  p = p + (a.neg_step_size * m) / denom;            // addcdiv_(exp_avg, denom, value=-step_size)
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v,
                                                  int64_t n4, int64_t n, AdamScalars a) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    adam_one(pp.x, gg.x, mm.x, vv.x, a); adam_one(pp.y, gg.y, mm.y, vv.y, a);
    adam_one(pp.z, gg.z, mm.z, vv.z, a); adam_one(pp.w, gg.w, mm.w, vv.w, a);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // scalar tail (numel % 4), handled by the first threads of block 0
  int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    adam_one(p[t], g[t], m[t], v[t], a);
    if (ZERO) g[t] = 0.0f;
  }
}

hipError_t launch_adam(float* p, float* g, float* m, float* v, int64_t n, double lr, double b1,
                       double b2, double eps, int step, int zero_grad, hipStream_t s) {
  if (n == 0) return hipSuccess;
  double bc1 = 1.0 - pow(b1, (double)step);
  double bc2 = 1.0 - pow(b2, (double)step);
  AdamScalars a;
  a.one_minus_b1 = (float)(1.0 - b1);
  a.b2 = (float)b2;
  a.one_minus_b2 = (float)(1.0 - b2);
  a.neg_step_size = (float)(-(lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15u) == 0;
  int64_t n4 = aligned ? n / 4 : 0;
  int64_t work = n4 > 0 ? n4 : n;
  unsigned blocks = (unsigned)((work + 255) / 256);
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  if (!aligned) {
    // unaligned storage: scalar path over everything (tail loop covers [0, n))
    blocks = (unsigned)((n + 255) / 256);
  }
  if (zero_grad) adam_kernel<true><<<blocks, 256, 0, s>>>(p, g, m, v, n4, n, a);
  else adam_kernel<false><<<blocks, 256, 0, s>>>(p, g, m, v, n4, n, a);
  return hipGetLastError();
}

}  // namespace miso
