// LDS-reduced scatter of the grid gradient for levels whose bricks are too large for the owner-computes pull
// ("brick push", round 5).  Semantics: the grid half of grid_sampler_3d_backward -- grad_input[c, corner] +=
// w_corner * gOut[c] for the eight corners of every sample (ATen's kernel; the second-order sibling is
// third_party/cuda_gridsample_grad2/gridsample_cuda.cu:462-481), which the reference issues as float atomics into HBM.
//
// What it replaces.  A level with more than 8 vertices per sort tile and axis (ScanNet's 200 x 100 x 200 over 16 tiles:
// 13 x 7 x 13, configs/rgbd/scannet.yaml:23-24) is beyond the pull kernels; until round 4 the training kernel scattered
// it itself -- 540 000 samples x 8 corners x 4 channels = 17 M float atomics at the ~156 G adds/s the memory side sustains,
// 100 of the 221 us of that kernel, plus a zero-fill of the level in front and the matrix-core push of the crowded
// coarse level beside it.  Owner-computes forms of it lost five times (tools/experiments/README.md): a tile's catchment
// has to be swept and routed, and that costs more than the atomics.
//
// Here a tile PUSHES, but into LDS.  The batch is binned by tile (sort.hip), so the samples of a tile touch only the
// tile's own brick plus one layer of vertices around it: (13+2) x (7+2) x (13+2) x 4 channels.  One workgroup per tile
// keeps that region in LDS IN DOUBLE -- ds_add_f64 sustains 3.5 lane-adds per clock and CU where ds_add_f32 does 0.38
// (tools/ubench/lds_atomics2.hip), ten times the rate of the atomics it replaces -- lane = sample, 8 C adds per level,
// no routing, no sweep of neighbours, every sample read exactly once.  Sums are formed in fp64 (each term w d is the
// reference's fp32 product), so the result no longer depends on the order of the samples: run-to-run reproducible.
//   brick_accumulate_kernel  tile -> its region, rounded to fp32, into a per-tile stage buffer (plain coalesced stores)
//   brick_gather_kernel      tile -> every vertex it OWNS = sum of the (up to 8, at most 27) stage regions that cover it,
//                            in a fixed order; plain stores (overwrite) or read-add-store; grad_touched flags
// No float atomics, no zero-fill of the level: a tile nobody contributes to is written as zeros, or -- when the caller
// says the gradient is zero already (MISO_F_GRAD_ZEROED: the optimizer clears what it consumes) -- not at all.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"

namespace miso {

constexpr int BRICK_MAXL = 4;
constexpr int BRICK_THREADS = 512;
constexpr size_t BRICK_LDS_MAX = 79 * 1024;      // two workgroups per CU

struct BrickLv {
  float* grad;
  unsigned char* touched;
  int size[3];               // X Y Z
  int stride[3];             // element strides of grad along x y z (channels contiguous)
  int foff, live;
  int soff;                  // first float of the level's region in a tile's stage record (= first double in LDS)
};

struct BrickK {
  int T[3];
  const int* tile_off;
  const float4* xn;          // tile-sorted normalised coordinates
  const float* dfeat;        // d-feat rows: row p (tile-sorted order) or, with perm, row perm[p]
  const int* perm;
  int64_t ld;
  int nl;
  int overwrite, zeroed;
  float* stage;              // (tiles, stride) floats
  int64_t stride;
  BrickLv lv[BRICK_MAXL];
};

__host__ __device__ inline int brick_fdiv(int a, int b) {   // floor(a / b), b > 0
  const int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// Vertices the samples binned into tile t (of T along an axis of `size` vertices) can touch: [lo, hi].  A sample of the
// tile sits at pos in [t size / T - 1/2, (t + 1) size / T - 1/2) -- widened by one numerator unit, 1 / (2 T) >= 0.015
// vertices, against the rounding of the sort's tile_of and of the position (both < 1e-3) -- its corners are
// floor(pos) and floor(pos) + 1; corners outside the grid carry no gradient (zeros padding).
__host__ __device__ inline void brick_region(int t, int size, int T, int& lo, int& hi) {
  lo = brick_fdiv(2 * t * size - T - 1, 2 * T);
  hi = brick_fdiv(2 * (t + 1) * size - T + 1, 2 * T) + 1;
  lo = lo < 0 ? 0 : lo;
  hi = hi > size - 1 ? size - 1 : hi;
}
// ... and the vertices tile t OWNS: [v0, v1) -- a partition of [0, size)
__host__ __device__ inline void brick_owned(int t, int size, int T, int& v0, int& v1) {
  v0 = (int)(((int64_t)t * size) / T);
  v1 = (int)(((int64_t)(t + 1) * size) / T);
}

// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(BRICK_THREADS) void brick_accumulate_kernel(BrickK k) {
  extern __shared__ __attribute__((aligned(16))) double brick[];
  const int t3[3] = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  const int tile = (t3[2] * k.T[1] + t3[1]) * k.T[0] + t3[0];
  const int p0 = k.tile_off[tile], p1 = k.tile_off[tile + 1];
  if (p0 >= p1) return;                    // an empty tile leaves no record: the gather looks at tile_off, not at the stage
  int lo[BRICK_MAXL][3], R[BRICK_MAXL][3];
  int total = 0;                           // doubles in use by this tile (its regions are packed level after level)
#pragma unroll
  for (int d = 0; d < BRICK_MAXL; ++d) {
    if (d >= k.nl) continue;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int hi;
      brick_region(t3[a], k.lv[d].size[a], k.T[a], lo[d][a], hi);
      R[d][a] = hi - lo[d][a] + 1;
    }
    total = k.lv[d].soff + R[d][0] * R[d][1] * R[d][2] * C;
  }
  for (int i = threadIdx.x * 2; i < total; i += BRICK_THREADS * 2) *reinterpret_cast<double2*>(brick + i) = make_double2(0.0, 0.0);
  __syncthreads();
  for (int p = p0 + (int)threadIdx.x; p < p1; p += BRICK_THREADS) {
    const float4 x4 = k.xn[p];
    const int64_t row = k.perm ? (int64_t)k.perm[p] : (int64_t)p;
    const float xs[3] = {x4.x, x4.y, x4.z};
#pragma unroll
    for (int d = 0; d < BRICK_MAXL; ++d) {
      if (d >= k.nl || !k.lv[d].live) continue;
      const BrickLv& lv = k.lv[d];
      int i0[3];
      float w[3][2];
      bool in[3][2];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        // position op for op as common.hpp:axis_coord on a normalised coordinate; weights as ATen forms them
        const float pos = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xs[a], 1.0f), (float)lv.size[a]), 1.0f), 0.5f);
        float f = fminf(fmaxf(floorf(pos), -2.0f), (float)lv.size[a] + 1.0f);
        int i = (pos == pos) ? (int)f : -4;
        w[a][0] = __fsub_rn(f + 1.0f, pos);
        w[a][1] = __fsub_rn(pos, f);
        i0[a] = i - lo[d][a];                                    // relative to the region
        // in the grid AND in the region (the second is implied by the binning; it keeps a stray sample out of LDS it
        // does not own rather than trusting the bound)
        in[a][0] = i >= 0 && i < lv.size[a] && i0[a] >= 0 && i0[a] < R[d][a];
        in[a][1] = i + 1 >= 0 && i + 1 < lv.size[a] && i0[a] + 1 >= 0 && i0[a] + 1 < R[d][a];
      }
      float dv[C];
      const float* dr = k.dfeat + row * k.ld + lv.foff;
#pragma unroll
      for (int c = 0; c < C; c += 4) {
        const float4 q = *reinterpret_cast<const float4*>(dr + c);
        dv[c] = q.x; dv[c + 1] = q.y; dv[c + 2] = q.z; dv[c + 3] = q.w;
      }
      const int base = lv.soff + ((i0[2] * R[d][1] + i0[1]) * R[d][0] + i0[0]) * C;
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int dx = kk & 1, dy = (kk >> 1) & 1, dz = kk >> 2;
        if (!(in[0][dx] && in[1][dy] && in[2][dz])) continue;
        const float wgt = (w[0][dx] * w[1][dy]) * w[2][dz];
        double* dst = brick + base + ((dz * R[d][1] + dy) * R[d][0] + dx) * C;
#pragma unroll
        for (int c = 0; c < C; ++c)
          __hip_atomic_fetch_add(dst + c, (double)(wgt * dv[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  float* out = k.stage + (int64_t)tile * k.stride;
  for (int i = threadIdx.x * 4; i < total; i += BRICK_THREADS * 4) {
    const double2 a = *reinterpret_cast<const double2*>(brick + i), b = *reinterpret_cast<const double2*>(brick + i + 2);
    *reinterpret_cast<float4*>(out + i) = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// One workgroup per tile: every vertex the tile owns = sum over the stage regions that cover it.  The tiles that can
// reach it lie within two tiles per axis (a region extends one vertex past the owned range on either side, two where the
// guard of brick_region bites; every tile owns at least one vertex per axis: plan_brick asks size >= T).
constexpr int GATHER_THREADS = 256;
constexpr int GATHER_MAXN = 125;

template <int C>
__global__ __launch_bounds__(GATHER_THREADS) void brick_gather_kernel(BrickK k) {
  __shared__ int s_ok[GATHER_MAXN];
  __shared__ int s_n;
  __shared__ int s_tile[GATHER_MAXN], s_lo[GATHER_MAXN][3], s_R[GATHER_MAXN][3];
  const int t3[3] = {(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  for (int d = 0; d < k.nl; ++d) {
    const BrickLv& lv = k.lv[d];
    int v0[3], v1[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) brick_owned(t3[a], lv.size[a], k.T[a], v0[a], v1[a]);
    if (v0[0] >= v1[0] || v0[1] >= v1[1] || v0[2] >= v1[2]) continue;
    __syncthreads();                       // (the previous level's list is still being read)
    // candidates: the 5 x 5 x 5 tiles around this one that hold samples and whose region meets the owned box
    if (threadIdx.x < GATHER_MAXN) {
      const int j = threadIdx.x;
      const int n3[3] = {t3[0] + j % 5 - 2, t3[1] + (j / 5) % 5 - 2, t3[2] + j / 25 - 2};
      int ok = lv.live && n3[0] >= 0 && n3[0] < k.T[0] && n3[1] >= 0 && n3[1] < k.T[1] && n3[2] >= 0 && n3[2] < k.T[2];
      if (ok) {
        const int nt = (n3[2] * k.T[1] + n3[1]) * k.T[0] + n3[0];
        ok = k.tile_off[nt + 1] > k.tile_off[nt];
        s_tile[j] = nt;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          int lo, hi;
          brick_region(n3[a], lv.size[a], k.T[a], lo, hi);
          s_lo[j][a] = lo; s_R[j][a] = hi - lo + 1;
          ok = ok && lo < v1[a] && hi >= v0[a];
        }
      }
      s_ok[j] = ok;
    }
    __syncthreads();
    if (threadIdx.x == 0) {                // compact in candidate order: the order of the sum is fixed
      int n = 0;
      for (int j = 0; j < GATHER_MAXN; ++j)
        if (s_ok[j]) {
          s_tile[n] = s_tile[j];
          for (int a = 0; a < 3; ++a) { s_lo[n][a] = s_lo[j][a]; s_R[n][a] = s_R[j][a]; }
          ++n;
        }
      s_n = n;
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0 && (!k.overwrite || k.zeroed)) continue;      // nothing to add / already zero
    const int E0 = v1[0] - v0[0], E1 = v1[1] - v0[1], E2 = v1[2] - v0[2];
    constexpr int Q = C / 4;               // float4 pieces per vertex
    const int items = E0 * E1 * E2 * Q;
    for (int it = threadIdx.x; it < items; it += GATHER_THREADS) {
      const int q = it % Q, v = it / Q;
      const int x = v0[0] + v % E0, y = v0[1] + (v / E0) % E1, z = v0[2] + v / (E0 * E1);
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int e = 0; e < n; ++e) {
        const int rx = x - s_lo[e][0], ry = y - s_lo[e][1], rz = z - s_lo[e][2];
        if ((unsigned)rx >= (unsigned)s_R[e][0] || (unsigned)ry >= (unsigned)s_R[e][1] || (unsigned)rz >= (unsigned)s_R[e][2])
          continue;
        const float4 s4 = *reinterpret_cast<const float4*>(
            k.stage + (int64_t)s_tile[e] * k.stride + lv.soff + ((rz * s_R[e][1] + ry) * s_R[e][0] + rx) * C + 4 * q);
        sum.x += s4.x; sum.y += s4.y; sum.z += s4.z; sum.w += s4.w;
      }
      const int64_t eo = (int64_t)z * lv.stride[2] + (int64_t)y * lv.stride[1] + (int64_t)x * lv.stride[0] + 4 * q;
      float4* dst = reinterpret_cast<float4*>(lv.grad + eo);
      if (lv.touched && (sum.x != 0.0f || sum.y != 0.0f || sum.z != 0.0f || sum.w != 0.0f))
        lv.touched[eo >> ADAM_CHUNK_SHIFT] = 1;
      if (!k.overwrite) {
        const float4 g4 = *dst;
        sum.x += g4.x; sum.y += g4.y; sum.z += g4.z; sum.w += g4.w;
      }
      *dst = sum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side.  Region sizes are bounded per level and axis by the largest region of any tile.
static int brick_rmax(int size, int T) {
  int m = 0;
  for (int t = 0; t < T; ++t) {
    int lo, hi;
    brick_region(t, size, T, lo, hi);
    if (hi - lo + 1 > m) m = hi - lo + 1;
  }
  return m;
}

static bool brick_level_ok(const GridK& g, const LevelK& lv, const int T[3]) {
  if (!lv.grad || (lv.C != 4 && lv.C != 8) || lv.sC != 1 || lv.C != g.lv[0].C) return false;
  const int size[3] = {lv.X, lv.Y, lv.Z};
  for (int a = 0; a < 3; ++a)
    if (size[a] < T[a] || (int64_t)size[a] * T[a] >= (1 << 28)) return false;      // every tile owns a vertex; no overflow
  return true;
}

static int64_t brick_level_floats(const LevelK& lv, const int T[3]) {
  return (int64_t)brick_rmax(lv.X, T[0]) * brick_rmax(lv.Y, T[1]) * brick_rmax(lv.Z, T[2]) * lv.C;
}

// Levels the brick push forms for this grid and binning, given the levels the pull kernels would own (`pull`,
// plan_grad_pull).  It is taken when a level exists that the pull cannot own -- those used to be scattered with float
// atomics from the backward kernel -- and then covers every level that fits beside it in LDS (the crowded coarse level
// that went through the matrix-core push, a level the pull would have formed: one pair of launches for all of them).
// MISO_BRICK=0 (dev / tests) switches it off.
uint32_t plan_brick(const GridK& g, int tiles, int64_t n, uint32_t pull) {
  static const bool off = [] { const char* e = getenv("MISO_BRICK"); return e && atoi(e) == 0; }();
  int T[3];
  if (off || !tiles_xyz(tiles, T) || n <= 0 || n >= (1ll << 31)) return 0;
  if (g.flags & (MISO_F_ALIGN_CORNERS | MISO_F_PAD_BORDER)) return 0;
  uint32_t mask = 0;
  size_t lds = 0;
  int cnt = 0;
  for (int pass = 0; pass < 2; ++pass)        // first the levels the pull cannot own, then the others
    for (int l = 0; l < g.n_levels && l < 16; ++l) {
      const LevelK& lv = g.lv[l];
      const bool beyond = lv.grad && !((pull >> l) & 1u) && !((g.ignore_mask >> l) & 1u);
      if ((pass == 0) != beyond || !brick_level_ok(g, lv, T) || cnt >= BRICK_MAXL) continue;
      const size_t need = (size_t)brick_level_floats(lv, T) * sizeof(double);
      if (lds + need > BRICK_LDS_MAX) continue;
      lds += need; mask |= 1u << l; ++cnt;
    }
  bool any_beyond = false;
  for (int l = 0; l < g.n_levels && l < 16; ++l)
    any_beyond = any_beyond || (((mask >> l) & 1u) && !((pull >> l) & 1u));
  return any_beyond ? mask : 0u;
}

// floats of the stage buffer (miso_sorted_t.brick_stage) the brick push of `mask` needs: one record per tile
int64_t brick_stage_floats(const GridK& g, int tiles, uint32_t mask) {
  int T[3];
  if (!mask || !tiles_xyz(tiles, T)) return 0;
  int64_t per = 0;
  for (int l = 0; l < g.n_levels && l < 16; ++l)
    if ((mask >> l) & 1u) per += brick_level_floats(g.lv[l], T);
  per = (per + 3) / 4 * 4;
  return per * T[0] * T[1] * T[2];
}

hipError_t launch_brick(const GridK& g, int C, int tiles, const int* tile_off, const float* xn, const float* dfeat,
                        int64_t ld, const int* perm, uint32_t mask, int overwrite, int zeroed, float* stage,
                        int64_t stage_floats, hipStream_t s) {
  BrickK k;
  memset(&k, 0, sizeof(k));
  if (!tiles_xyz(tiles, k.T) || !stage || stage_floats < brick_stage_floats(g, tiles, mask)) return hipErrorInvalidValue;
  k.tile_off = tile_off; k.xn = reinterpret_cast<const float4*>(xn); k.dfeat = dfeat; k.perm = perm; k.ld = ld;
  k.overwrite = overwrite; k.zeroed = zeroed; k.stage = stage;
  int64_t per = 0;
  for (int l = 0; l < g.n_levels && l < 16; ++l)
    if ((mask >> l) & 1u) {
      if (k.nl >= BRICK_MAXL) return hipErrorInvalidValue;
      const LevelK& lv = g.lv[l];
      BrickLv& o = k.lv[k.nl++];
      o.grad = lv.grad; o.touched = lv.touched;
      o.size[0] = lv.X; o.size[1] = lv.Y; o.size[2] = lv.Z;
      o.stride[0] = lv.sX; o.stride[1] = lv.sY; o.stride[2] = lv.sZ;
      o.foff = lv.foff; o.live = ((g.ignore_mask >> l) & 1u) ? 0 : 1;
      o.soff = (int)per;
      per += brick_level_floats(lv, k.T);
    }
  per = (per + 3) / 4 * 4;
  k.stride = per;
  const size_t lds = (size_t)per * sizeof(double);
  const dim3 grid((unsigned)k.T[0], (unsigned)k.T[1], (unsigned)k.T[2]);
  void (*ka)(BrickK) = C == 8 ? brick_accumulate_kernel<8> : brick_accumulate_kernel<4>;
  void (*kg)(BrickK) = C == 8 ? brick_gather_kernel<8> : brick_gather_kernel<4>;
  hipError_t e = allow_dynamic_lds((const void*)ka, lds);
  if (e != hipSuccess) return e;
  ka<<<grid, BRICK_THREADS, lds, s>>>(k);
  kg<<<grid, GATHER_THREADS, 0, s>>>(k);
  return hipGetLastError();
}

}  // namespace miso
