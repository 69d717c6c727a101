// Generic multi-level trilinear encode: forward, first backward, second backward.
//
// One lane per point; any channel count and any strides (the reference's NCDHW
// layout as well as channels-last).  With channels-last and C % 4 == 0 a corner
// is read / scattered as float4 vectors (one 16-B access per 4 channels instead
// of 4 strided dwords).  These kernels back the autograd Functions that must be
// differentiable to second order; the fused MFMA kernels in sdf_fused.hip cover
// the frozen-decoder fast path.
//
// Math: SURVEY.md Appendix B; it restates ATen grid_sampler_3d (bilinear) and
// the double backward of third_party/cuda_gridsample_grad2/gridsample_cuda.cu:443-531.
#include "common.hpp"

namespace miso {

template <bool VEC4>
__global__ __launch_bounds__(256) void encode_fwd_kernel(GridK g, const float* __restrict__ x,
                                                        int64_t n, float* __restrict__ out,
                                                        int64_t ld, const int* __restrict__ perm) {
  // perm != nullptr: x is the tile-sorted (pre-normalised) copy of the batch; rows go back to the
  // caller's order, out[perm[p]]
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float px, py, pz;
  load_point(g, x, p, px, py, pz);
  float* o = out + (perm ? (int64_t)perm[p] : p) * ld;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if ((g.ignore_mask >> l) & 1u) {
      for (int c = 0; c < lv.C; ++c) o[lv.foff + c] = 0.0f;
      continue;
    }
    Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
    Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
    Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
    Cell c = make_cell(ax, ay, az, lv);
    int off[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      bool in = c.inx[dx] && c.iny[dy] && c.inz[dz];
      w[k] = in ? (c.wx[dx] * c.wy[dy]) * c.wz[dz] : 0.0f;
      off[k] = in ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
    }
    if (VEC4) {
      for (int ch = 0; ch < lv.C; ch += 4) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(lv.data + off[k] + ch);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          acc.x += v[k].x * w[k]; acc.y += v[k].y * w[k];
          acc.z += v[k].z * w[k]; acc.w += v[k].w * w[k];
        }
        o[lv.foff + ch + 0] = acc.x; o[lv.foff + ch + 1] = acc.y;
        o[lv.foff + ch + 2] = acc.z; o[lv.foff + ch + 3] = acc.w;
      }
    } else {
      for (int ch = 0; ch < lv.C; ++ch) {
        const float* base = lv.data + (int64_t)ch * lv.sC;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += base[off[k]] * w[k];
        o[lv.foff + ch] = acc;
      }
    }
  }
}

// First backward: scatter grad_feats into level grads (where non-null) and/or
// grad_x = sum_l mult * sum_c gF[c] * sum_corners (dw/dix) * G[c,corner].
template <bool VEC4>
__global__ __launch_bounds__(256) void encode_bwd_kernel(GridK g, const float* __restrict__ x,
                                                        int64_t n, const float* __restrict__ gf,
                                                        int64_t ld, float* __restrict__ gx,
                                                        const int* __restrict__ perm) {
  // perm != nullptr: x is the tile-sorted (pre-normalised) copy; gf / gx rows in the caller's order
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float px, py, pz;
  load_point(g, x, p, px, py, pz);
  const int64_t po = perm ? (int64_t)perm[p] : p;
  const float* go = gf + po * ld;
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if ((g.ignore_mask >> l) & 1u) continue;
    Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
    Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
    Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
    Cell c = make_cell(ax, ay, az, lv);
    int off[8];
    float w[8], dwx[8], dwy[8], dwz[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      bool in = c.inx[dx] && c.iny[dy] && c.inz[dz];
      inb[k] = in;
      float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
      w[k] = in ? (c.wx[dx] * c.wy[dy]) * c.wz[dz] : 0.0f;
      dwx[k] = in ? sx * c.wy[dy] * c.wz[dz] : 0.0f;
      dwy[k] = in ? sy * c.wx[dx] * c.wz[dz] : 0.0f;
      dwz[k] = in ? sz * c.wx[dx] * c.wy[dy] : 0.0f;
      off[k] = in ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
    }
    float ax_ = 0.f, ay_ = 0.f, az_ = 0.f;
    if (VEC4) {
      for (int ch = 0; ch < lv.C; ch += 4) {
        float4 gv = make_float4(go[lv.foff + ch], go[lv.foff + ch + 1], go[lv.foff + ch + 2],
                                go[lv.foff + ch + 3]);
        if (gx) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float4 v = *reinterpret_cast<const float4*>(lv.data + off[k] + ch);
            float d = gv.x * v.x + gv.y * v.y + gv.z * v.z + gv.w * v.w;
            ax_ += d * dwx[k]; ay_ += d * dwy[k]; az_ += d * dwz[k];
          }
        }
        if (lv.grad) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            if (!inb[k]) continue;
            float* t = lv.grad + off[k] + ch;
            atomic_add_f32(t + 0, gv.x * w[k]); atomic_add_f32(t + 1, gv.y * w[k]);
            atomic_add_f32(t + 2, gv.z * w[k]); atomic_add_f32(t + 3, gv.w * w[k]);
            touch_chunk(lv, off[k] + ch);
          }
        }
      }
    } else {
      for (int ch = 0; ch < lv.C; ++ch) {
        float gv = go[lv.foff + ch];
        int64_t cb = (int64_t)ch * lv.sC;
        if (gx) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            float d = gv * lv.data[cb + off[k]];
            ax_ += d * dwx[k]; ay_ += d * dwy[k]; az_ += d * dwz[k];
          }
        }
        if (lv.grad) {
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (inb[k]) { atomic_add_f32(lv.grad + cb + off[k], gv * w[k]); touch_chunk(lv, cb + off[k]); }
        }
      }
    }
    gpx += ax_ * (g.gscale[0] * ax.mult); gpy += ay_ * (g.gscale[1] * ay.mult); gpz += az_ * (g.gscale[2] * az.mult);
  }
  if (gx) { gx[po * 3 + 0] = gpx; gx[po * 3 + 1] = gpy; gx[po * 3 + 2] = gpz; }
}

// grad_x alone (channels-last levels, the grid gradient formed elsewhere or not wanted): the eight corner values
// contract with gF first, d_k = <gF, G[:, corner k]>, and grad_x is the gradient of the trilinear interpolant of that
// ONE scalar per corner -- a lerp tree instead of 24 derivative weights.
__global__ __launch_bounds__(256) void encode_bwd_x_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                          const float* __restrict__ gf, int64_t ld,
                                                          float* __restrict__ gx, const int* __restrict__ perm) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float px, py, pz;
  load_point(g, x, p, px, py, pz);
  const int64_t po = perm ? (int64_t)perm[p] : p;
  const float* go = gf + po * ld;
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if ((g.ignore_mask >> l) & 1u) continue;
    Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
    Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
    Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
    Cell c = make_cell(ax, ay, az, lv);
    float d[8];
    int off[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      inb[k] = c.inx[dx] && c.iny[dy] && c.inz[dz];
      off[k] = inb[k] ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
      d[k] = 0.0f;
    }
    for (int ch = 0; ch < lv.C; ch += 4) {
      const float4 gv = *reinterpret_cast<const float4*>(go + lv.foff + ch);
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(lv.data + off[k] + ch);
#pragma unroll
      for (int k = 0; k < 8; ++k) d[k] += gv.x * v[k].x + gv.y * v[k].y + gv.z * v[k].z + gv.w * v[k].w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = inb[k] ? d[k] : 0.0f;
    const float tx = c.wx[1], ty = c.wy[1], tz = c.wz[1];
    const float d00 = d[1] - d[0], d10 = d[3] - d[2], d01 = d[5] - d[4], d11 = d[7] - d[6];
    const float a00 = d[0] + tx * d00, a10 = d[2] + tx * d10, a01 = d[4] + tx * d01, a11 = d[6] + tx * d11;
    const float e0 = a10 - a00, e1 = a11 - a01;
    const float b0 = a00 + ty * e0, b1 = a01 + ty * e1;
    const float gx0 = d00 + ty * (d10 - d00), gx1 = d01 + ty * (d11 - d01);
    gpx += (gx0 + tz * (gx1 - gx0)) * (g.gscale[0] * ax.mult);
    gpy += (e0 + tz * (e1 - e0)) * (g.gscale[1] * ay.mult);
    gpz += (b1 - b0) * (g.gscale[2] * az.mult);
  }
  gx[po * 3 + 0] = gpx; gx[po * 3 + 1] = gpy; gx[po * 3 + 2] = gpz;
}

// Second backward.  Inputs: ggG (= lv.gg, cotangent of grad_grid, may be null),
// ggx (cotangent of grad_x, may be null), gF (= grad_feats of the first
// backward).  Outputs: ggF (N,F); gG scatter (lv.grad, may be null); gx (may be null).
//   ggF[c]      = sum inb * ( w * ggG[c,k] + G[c,k] * (grad w . d) ),   d = mult o ggx
//   gG[c,k]    += inb * (grad w . d) * gF[c]
//   gx_x        = mult_x * sum_c gF[c] * sum_k inb * ( dwx*ggG[c,k]
//                          + G[c,k] * (d_y * dwxy + d_z * dwxz) )        (d2w/dx2 = 0)
// VEC4: channels-last levels with C % 4 == 0 (one 16-B access per corner and channel quad).
// perm != nullptr: x is the tile-sorted (pre-normalised) copy of the batch and every per-point row
// (gf, ggx, ggo, gx) is addressed through perm, i.e. stays in the caller's order.
template <bool VEC4>
__global__ __launch_bounds__(256) void encode_bwd2_kernel(GridK g, const float* __restrict__ x,
                                                         int64_t n, const float* __restrict__ gf,
                                                         int64_t ld, const float* __restrict__ ggx,
                                                         float* __restrict__ ggo, int64_t ldgg,
                                                         float* __restrict__ gx, const int* __restrict__ perm) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float px, py, pz;
  load_point(g, x, p, px, py, pz);
  const int64_t po = perm ? (int64_t)perm[p] : p;
  const float* go = gf + po * ld;
  float* ggout = ggo + po * ldgg;
  float ex = 0.f, ey = 0.f, ez = 0.f;
  if (ggx) { ex = ggx[po * 3 + 0]; ey = ggx[po * 3 + 1]; ez = ggx[po * 3 + 2]; }
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if ((g.ignore_mask >> l) & 1u) {
      for (int c = 0; c < lv.C; ++c) ggout[lv.foff + c] = 0.0f;
      continue;
    }
    Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
    Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
    Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
    Cell c = make_cell(ax, ay, az, lv);
    // g.gscale restores d xn / d x when the points were pre-normalised (sorted batches)
    const float mx = g.gscale[0] * ax.mult, my = g.gscale[1] * ay.mult, mz = g.gscale[2] * az.mult;
    float dxi = ex * mx, dyi = ey * my, dzi = ez * mz;
    int off[8];
    float w[8], dwx[8], dwy[8], dwz[8], gwd[8], cx[8], cy[8], cz[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      bool in = c.inx[dx] && c.iny[dy] && c.inz[dz];
      inb[k] = in;
      float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
      float m = in ? 1.0f : 0.0f;
      w[k] = m * (c.wx[dx] * c.wy[dy]) * c.wz[dz];
      dwx[k] = m * sx * c.wy[dy] * c.wz[dz];
      dwy[k] = m * sy * c.wx[dx] * c.wz[dz];
      dwz[k] = m * sz * c.wx[dx] * c.wy[dy];
      float dwxy = m * sx * sy * c.wz[dz];
      float dwxz = m * sx * sz * c.wy[dy];
      float dwyz = m * sy * sz * c.wx[dx];
      gwd[k] = dxi * dwx[k] + dyi * dwy[k] + dzi * dwz[k];
      cx[k] = dyi * dwxy + dzi * dwxz;   // multiplies G in gx_x
      cy[k] = dxi * dwxy + dzi * dwyz;
      cz[k] = dxi * dwxz + dyi * dwyz;
      off[k] = in ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
    }
    float ax_ = 0.f, ay_ = 0.f, az_ = 0.f;
    if (VEC4) {
      for (int ch = 0; ch < lv.C; ch += 4) {
        const float gv[4] = {go[lv.foff + ch], go[lv.foff + ch + 1], go[lv.foff + ch + 2], go[lv.foff + ch + 3]};
        float ggv[4] = {0.f, 0.f, 0.f, 0.f};
        float4 vv[8], g2v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          vv[k] = *reinterpret_cast<const float4*>(lv.data + off[k] + ch);
          g2v[k] = lv.gg ? *reinterpret_cast<const float4*>(lv.gg + off[k] + ch) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float val[4] = {vv[k].x, vv[k].y, vv[k].z, vv[k].w};
          const float g2[4] = {g2v[k].x, g2v[k].y, g2v[k].z, g2v[k].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {     // same operation order per channel as the scalar path
            ggv[e] += g2[e] * w[k] + val[e] * gwd[k];
            ax_ += gv[e] * (g2[e] * dwx[k] + val[e] * cx[k]);
            ay_ += gv[e] * (g2[e] * dwy[k] + val[e] * cy[k]);
            az_ += gv[e] * (g2[e] * dwz[k] + val[e] * cz[k]);
            if (lv.grad && inb[k] && ggx) {
              atomic_add_f32(lv.grad + off[k] + ch + e, gwd[k] * gv[e]);
              if (e == 0) touch_chunk(lv, off[k] + ch);
            }
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) ggout[lv.foff + ch + e] = ggv[e];
      }
    } else {
      for (int ch = 0; ch < lv.C; ++ch) {
        int64_t cb = (int64_t)ch * lv.sC;
        float gv = go[lv.foff + ch];
        float ggv = 0.0f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float val = lv.data[cb + off[k]];
          float g2 = lv.gg ? lv.gg[cb + off[k]] : 0.0f;
          ggv += g2 * w[k] + val * gwd[k];
          ax_ += gv * (g2 * dwx[k] + val * cx[k]);
          ay_ += gv * (g2 * dwy[k] + val * cy[k]);
          az_ += gv * (g2 * dwz[k] + val * cz[k]);
          if (lv.grad && inb[k] && ggx) { atomic_add_f32(lv.grad + cb + off[k], gwd[k] * gv); touch_chunk(lv, cb + off[k]); }
        }
        ggout[lv.foff + ch] = ggv;
      }
    }
    gpx += ax_ * mx; gpy += ay_ * my; gpz += az_ * mz;
  }
  if (gx) { gx[po * 3 + 0] = gpx; gx[po * 3 + 1] = gpy; gx[po * 3 + 2] = gpz; }
}

// The same second backward for the case the binned path produces (channels-last levels, no grid scatter from here: the
// pull forms that gradient): value, first and mixed second derivatives of the trilinear interpolant by a lerp tree on
// the zero-padded corner values instead of eight arrays of corner weights.  The weight form above keeps 64 weight
// registers per level next to 64 gathered corner values (252 VGPRs: two wavefronts per SIMD on a kernel bound by the
// latency of its gathers).
//   ggF[c] = f(ggG_c) + d . grad f(G_c)
//   gx_a   = mult_a * sum_c gF[c] * ( d_a f(ggG_c) + sum_{b != a} d_b d_ab f(G_c) )          (d_aa f = 0)
struct Lerp3 {
  float f, fx, fy, fz, fxy, fxz, fyz;
};
template <bool SECOND>
__device__ __forceinline__ Lerp3 lerp_tree(const float v[8], float tx, float ty, float tz) {
  Lerp3 r;
  const float d00 = v[1] - v[0], d10 = v[3] - v[2], d01 = v[5] - v[4], d11 = v[7] - v[6];       // d/dx on the (y,z) edges
  const float a00 = v[0] + tx * d00, a10 = v[2] + tx * d10, a01 = v[4] + tx * d01, a11 = v[6] + tx * d11;
  const float e0 = a10 - a00, e1 = a11 - a01;                                                     // d/dy at z = 0, 1
  const float b0 = a00 + ty * e0, b1 = a01 + ty * e1;
  r.f = b0 + tz * (b1 - b0);
  r.fz = b1 - b0;
  r.fy = e0 + tz * (e1 - e0);
  const float gx0 = d00 + ty * (d10 - d00), gx1 = d01 + ty * (d11 - d01);
  r.fx = gx0 + tz * (gx1 - gx0);
  if (SECOND) {
    r.fyz = e1 - e0;
    r.fxy = (d10 - d00) + tz * ((d11 - d01) - (d10 - d00));
    r.fxz = (d01 - d00) + ty * ((d11 - d10) - (d01 - d00));
  } else {
    r.fyz = r.fxy = r.fxz = 0.0f;
  }
  return r;
}

template <bool HAS_GG>
__global__ __launch_bounds__(256) void encode_bwd2_lean_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                              const float* __restrict__ gf, int64_t ld,
                                                              const float* __restrict__ ggx, float* __restrict__ ggo,
                                                              int64_t ldgg, float* __restrict__ gx,
                                                              const int* __restrict__ perm) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float px, py, pz;
  load_point(g, x, p, px, py, pz);
  const int64_t po = perm ? (int64_t)perm[p] : p;
  const float* go = gf + po * ld;
  float* ggout = ggo + po * ldgg;
  float ex = 0.f, ey = 0.f, ez = 0.f;
  if (ggx) { ex = ggx[po * 3 + 0]; ey = ggx[po * 3 + 1]; ez = ggx[po * 3 + 2]; }
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if ((g.ignore_mask >> l) & 1u) {
      for (int c = 0; c < lv.C; ++c) ggout[lv.foff + c] = 0.0f;
      continue;
    }
    Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
    Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
    Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
    Cell c = make_cell(ax, ay, az, lv);
    const float mx = g.gscale[0] * ax.mult, my = g.gscale[1] * ay.mult, mz = g.gscale[2] * az.mult;
    const float dxi = ex * mx, dyi = ey * my, dzi = ez * mz;
    int off[8];
    bool inb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
      inb[k] = c.inx[dx] && c.iny[dy] && c.inz[dz];
      off[k] = inb[k] ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
    }
    const float tx = c.wx[1], ty = c.wy[1], tz = c.wz[1];
    const bool gg_here = HAS_GG && lv.gg != nullptr;
    float ax_ = 0.f, ay_ = 0.f, az_ = 0.f;
    for (int ch = 0; ch < lv.C; ch += 4) {
      const float4 gv4 = *reinterpret_cast<const float4*>(go + lv.foff + ch);
      const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w};
      float4 vv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
      {   // (loaded whatever the flag: offset 0 of an out-of-range corner is a valid address; a load under a condition is a
          // branch per corner and the eight gathers then issue one by one -- pair_latent.hip, round 5)
        vv[k] = *reinterpret_cast<const float4*>(lv.data + off[k] + ch);
        if (!inb[k]) vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float ggv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = e == 0 ? vv[k].x : (e == 1 ? vv[k].y : (e == 2 ? vv[k].z : vv[k].w));
        const Lerp3 t = lerp_tree<true>(v, tx, ty, tz);
        ggv[e] = dxi * t.fx + dyi * t.fy + dzi * t.fz;
        ax_ += gv[e] * (dyi * t.fxy + dzi * t.fxz);
        ay_ += gv[e] * (dxi * t.fxy + dzi * t.fyz);
        az_ += gv[e] * (dxi * t.fxz + dyi * t.fyz);
      }
      if (gg_here) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
          {
            vv[k] = *reinterpret_cast<const float4*>(lv.gg + off[k] + ch);
            if (!inb[k]) vv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v[8];
#pragma unroll
          for (int k = 0; k < 8; ++k) v[k] = e == 0 ? vv[k].x : (e == 1 ? vv[k].y : (e == 2 ? vv[k].z : vv[k].w));
          const Lerp3 t = lerp_tree<false>(v, tx, ty, tz);
          ggv[e] += t.f;
          ax_ += gv[e] * t.fx; ay_ += gv[e] * t.fy; az_ += gv[e] * t.fz;
        }
      }
      *reinterpret_cast<float4*>(ggout + lv.foff + ch) = make_float4(ggv[0], ggv[1], ggv[2], ggv[3]);
    }
    gpx += ax_ * mx; gpy += ay_ * my; gpz += az_ * mz;
  }
  if (gx) { gx[po * 3 + 0] = gpx; gx[po * 3 + 1] = gpy; gx[po * 3 + 2] = gpz; }
}

// ---- host-side launch helpers (called from capi.hip) -----------------------
static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

hipError_t launch_encode_fwd(const GridK& g, bool vec4, const float* x, int64_t n, float* out,
                             int64_t ld, const int* perm, hipStream_t s) {
  if (n == 0) return hipSuccess;
  if (vec4) encode_fwd_kernel<true><<<blocks_for(n), 256, 0, s>>>(g, x, n, out, ld, perm);
  else encode_fwd_kernel<false><<<blocks_for(n), 256, 0, s>>>(g, x, n, out, ld, perm);
  return hipGetLastError();
}

hipError_t launch_encode_bwd(const GridK& g, bool vec4, const float* x, int64_t n, const float* gf,
                             int64_t ld, float* gx, const int* perm, hipStream_t s) {
  if (n == 0) return hipSuccess;
  bool scatter = false;
  for (int l = 0; l < g.n_levels; ++l) scatter = scatter || g.lv[l].grad != nullptr;
  if (vec4 && gx && !scatter && ld % 4 == 0 && ((uintptr_t)gf & 15u) == 0 && !getenv("MISO_ENCODE_NO_LEAN")) {   // dev / tests: the weight-form kernel
    encode_bwd_x_kernel<<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, gx, perm);
    return hipGetLastError();
  }
  if (vec4) encode_bwd_kernel<true><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, gx, perm);
  else encode_bwd_kernel<false><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, gx, perm);
  return hipGetLastError();
}

hipError_t launch_encode_bwd2(const GridK& g, bool vec4, const float* x, int64_t n, const float* gf, int64_t ld,
                              const float* ggx, float* ggo, int64_t ldgg, float* gx, const int* perm,
                              hipStream_t s) {
  if (n == 0) return hipSuccess;
  // no grid scatter from this launch (the binned path leaves it to the pull, or no cotangent of grad_x): lean kernel,
  // given 16-B aligned rows
  bool scatter = false, has_gg = false;
  for (int l = 0; l < g.n_levels; ++l) {
    scatter = scatter || (g.lv[l].grad != nullptr && ggx != nullptr);
    has_gg = has_gg || g.lv[l].gg != nullptr;
  }
  const bool no_lean = getenv("MISO_ENCODE_NO_LEAN") != nullptr;      // dev / tests: the weight-form kernel
  if (vec4 && !scatter && !no_lean && ld % 4 == 0 && ldgg % 4 == 0 && ((uintptr_t)gf & 15u) == 0 &&
      ((uintptr_t)ggo & 15u) == 0) {
    if (has_gg) encode_bwd2_lean_kernel<true><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, ggx, ggo, ldgg, gx, perm);
    else encode_bwd2_lean_kernel<false><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, ggx, ggo, ldgg, gx, perm);
    return hipGetLastError();
  }
  if (vec4) encode_bwd2_kernel<true><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, ggx, ggo, ldgg, gx, perm);
  else encode_bwd2_kernel<false><<<blocks_for(n), 256, 0, s>>>(g, x, n, gf, ld, ggx, ggo, ldgg, gx, perm);
  return hipGetLastError();
}

}  // namespace miso
