// GridAtlas.query_feature / GridAtlas.forward in ONE launch (round 6).
//
// Reference (grid_opt/models/grid_atlas.py:374-399): for every active submap s -- world points into the submap frame
// (utils_geometry.transfrom_points_from), coords_in_bound mask, a full multi-level grid_interp_regular of EVERY point
// (masked-out ones too), mask * feats added to an (N,F) running sum, the mask to an (N,1) count -- then count == 0 -> 1,
// sum / count, and submap 0's decoder on the mean (utils.grid_decode).  Every demo's final global mesh runs it at
// resolution 512 (134 M points x S submaps: demo/align_submaps.py:99, full_slam_scannet.py:116).
//
// Here: one wavefront per 64 points, lane = point.  Per submap (poses and bounds are wave-uniform: scalar registers) the
// frame change and the bound test; a submap that no point of the wavefront is inside costs nothing more (one ballot), and
// inside ones are encoded for the lanes that are inside only.  Sum, count and mean stay in registers, the mean feeds the
// decoder chain of sdf_fwd_kernel (decoder.hpp: bf16x3 split products, or the exact fp32 chains behind MISO_F_EXACT_F32)
// without touching HBM.  Lattice queries (utils_sdf.extract_fields: the points ARE a linspace^3 meshgrid) generate their
// coordinates from the point index and three short axis tables: no (N,3) tensor is ever formed.
// A chunk with no point inside any submap (the empty corners of a scene's bounding box) decodes the all-zero feature row:
// that value is computed once per wavefront and stored.
#include <stdlib.h>

#include "decoder.hpp"

namespace miso {

template <int C, int L, int H, int NH, bool SPLIT>
__global__ __launch_bounds__(256, 2) void atlas_sdf_kernel(AtlasK a, const float* __restrict__ packed) {
  constexpr int F = C * L, RT = H / 32, KS0 = (F + 1) / 2;
  constexpr int MW = (NH + 1) * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PackLayout pl(F, H, NH);
  const int n_split = pl.s_fwd_end - pl.s_w0;
  if (a.sdf) {      // (a feature-only query stages nothing)
    if (SPLIT) {
      for (int i = threadIdx.x * 4; i < n_split; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + pl.s_w0 + i);
      for (int i = threadIdx.x * 4; i < pl.n_bias(); i += blockDim.x * 4)
        *reinterpret_cast<float4*>(smem + n_split + i) = *reinterpret_cast<const float4*>(packed + pl.o_b0 + i);
    } else {
      for (int i = threadIdx.x * 4; i < pl.fwd_end; i += blockDim.x * 4)
        *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + i);
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hi = lane >> 5;
  const uint32_t* s_fwd = reinterpret_cast<const uint32_t*>(smem);
  const float* s_bias = smem + n_split;
  const float bo = SPLIT ? s_bias[pl.o_bo - pl.o_b0] : smem[pl.o_bo];

  auto decode = [&](const float (&f)[2 * KS0]) __attribute__((always_inline)) -> float {
    uint32_t mw[MW];
    float p0 = 0.0f, p1 = 0.0f, poison = 0.0f;
    if constexpr (SPLIT) {
      u32x4 no_mask[H / 16][2];
      decoder_fwd_split<F, H, NH, false, false, false>(s_fwd, s_bias, lane, f, mw, no_mask, p0, p1, poison);
    } else {
      decoder_fwd_exact<F, H, NH>(smem + pl.o_w0, smem + pl.o_wh, smem + pl.o_b0, smem + pl.o_bh, smem + pl.o_wo, lane, f,
                                  mw, p0, p1);
    }
    p0 += __shfl_xor(p0, 32);
    p1 += __shfl_xor(p1, 32);
    return SPLIT ? ((hi ? p1 : p0) + bo) + poison : (hi ? p1 : p0) + bo;
  };

  // the decoder's answer to an all-zero feature row (a point inside no submap): once per wavefront
  float sdf_empty = 0.0f;
  if (a.sdf) {
    float z[2 * KS0];
#pragma unroll
    for (int i = 0; i < 2 * KS0; ++i) z[i] = 0.0f;
    sdf_empty = decode(z);
  }

  const int64_t nchunks = (a.n + 63) / 64;
  const uint32_t nyz = (uint32_t)a.dim[1] * (uint32_t)a.dim[2];
  for (int64_t chunk = (int64_t)blockIdx.x * 4 + wave; chunk < nchunks; chunk += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");      // (keeps the LDS reads of weights / biases inside the loop: sdf_fwd_kernel)
    const int64_t p = chunk * 64 + lane;
    const bool valid = p < a.n;
    float wx = 0.f, wy = 0.f, wz = 0.f;
    if (valid) {
      if (a.x) {
        wx = a.x[p * 3 + 0]; wy = a.x[p * 3 + 1]; wz = a.x[p * 3 + 2];
      } else {      // (a lattice launch has n < 2^31: checked on the host)
        const uint32_t q = (uint32_t)p, i = q / nyz, r = q - i * nyz, j = r / (uint32_t)a.dim[2], k = r - j * (uint32_t)a.dim[2];
        wx = a.ax[0][i]; wy = a.ax[1][j]; wz = a.ax[2][k];
      }
    }
    float sum[2 * KS0];
#pragma unroll
    for (int i = 0; i < 2 * KS0; ++i) sum[i] = 0.0f;
    float cnt = 0.0f;
    bool any_inside = false;
    for (int s = 0; s < a.n_submaps; ++s) {
      const float* ps = a.poses + s * 12;
      const GridK& g = a.submaps[s];
      // transfrom_points_from (utils_geometry.py:227-240) = transform_points_to with (R^T, -R^T t), both formed by the
      // caller with the reference's own tensor ops; the row-times-matrix product in torch's order: ((x r0) + y r1) + z r2, + t
      float xl[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float v = __fmul_rn(wx, ps[3 * j]);
        v = __fmaf_rn(wy, ps[3 * j + 1], v);
        v = __fmaf_rn(wz, ps[3 * j + 2], v);
        xl[j] = __fadd_rn(v, ps[9 + j]);
      }
      // coords_in_bound (utils_geometry.py:11-27): min <= x <= max on every axis
      const bool inside = valid && (a.no_bound || (xl[0] >= g.bmin[0] && xl[0] <= g.bmax[0] && xl[1] >= g.bmin[1] &&
                                                   xl[1] <= g.bmax[1] && xl[2] >= g.bmin[2] && xl[2] <= g.bmax[2]));
      if (!__any(inside)) continue;
      any_inside = true;
      if (inside) {
        cnt += 1.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          if ((g.ignore_mask >> l) & 1u) continue;      // (zeros: utils.py:160-163; the atlas queries never set it)
          const LevelK lv = g.lv[l];
          Axis ax = axis_coord(xl[0], g.bmin[0], g.bmax[0], lv.X, g.flags);
          Axis ay = axis_coord(xl[1], g.bmin[1], g.bmax[1], lv.Y, g.flags);
          Axis az = axis_coord(xl[2], g.bmin[2], g.bmax[2], lv.Z, g.flags);
          Cell c = make_cell(ax, ay, az, lv);
          float fl[C];
          gather_level<C>(lv, c, fl);
#pragma unroll
          for (int q = 0; q < C; ++q) sum[l * C + q] += fl[q];      // sum_feats += mask * feats, submap by submap
        }
      }
    }
    // sum_weights[sum_weights == 0] = 1; mean = sum / weights
    const float den = cnt == 0.0f ? 1.0f : cnt;
    float mean[2 * KS0];
#pragma unroll
    for (int i = 0; i < 2 * KS0; ++i) mean[i] = (i < F) ? __fdiv_rn(sum[i], den) : 0.0f;
    if (a.feats && valid) {
      float* dst = a.feats + p * a.ld;
#pragma unroll
      for (int i = 0; i < F; ++i) dst[i] = mean[i];
    }
    if (a.sdf) {
      if (!any_inside) {      // wave-uniform
        if (valid) a.sdf[p] = sdf_empty;
      } else {
        const float v = decode(mean);
        if (valid) a.sdf[p] = v;
      }
    }
  }
}

#define MISO_ATLAS_SHAPES(X) \
  X(4, 1, 32, 1) X(4, 1, 64, 1) X(4, 2, 32, 1) X(4, 2, 64, 1) X(4, 3, 64, 1) X(4, 4, 64, 1) \
  X(8, 1, 64, 1) X(8, 2, 64, 1) X(8, 3, 64, 1) X(8, 4, 64, 1) X(8, 3, 32, 1)

template <int C, int L, int H, int NH>
static hipError_t launch_atlas_t(const AtlasK& a, const float* packed, bool split, hipStream_t s) {
  PackLayout pl(C * L, H, NH);
  size_t lds = (size_t)(split ? pl.s_fwd_end - pl.s_w0 + (pl.n_bias() + 3) / 4 * 4 : (pl.fwd_end + 3) / 4 * 4) * sizeof(float);
  if (!a.sdf) lds = 16;
  const int64_t nchunks = (a.n + 63) / 64;
  unsigned blocks = (unsigned)((nchunks + 3) / 4);
  if (blocks > 2048u) blocks = 2048u;
  auto k = split ? atlas_sdf_kernel<C, L, H, NH, true> : atlas_sdf_kernel<C, L, H, NH, false>;
  hipError_t e = allow_dynamic_lds((const void*)k, lds);
  if (e != hipSuccess) return e;
  k<<<blocks, 256, lds, s>>>(a, packed);
  return hipGetLastError();
}

hipError_t launch_atlas_sdf(int C, int L, int H, int NH, const AtlasK& a, const float* packed, bool exact, hipStream_t s) {
  if (a.n == 0) return hipSuccess;
  static const bool env_exact = [] { const char* e = getenv("MISO_EXACT_F32"); return e && atoi(e) != 0; }();
  const bool split = !exact && !env_exact;
#define X(c, l, h, nh) \
  if (C == c && L == l && H == h && NH == nh) return launch_atlas_t<c, l, h, nh>(a, packed, split, s);
  MISO_ATLAS_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

}  // namespace miso
