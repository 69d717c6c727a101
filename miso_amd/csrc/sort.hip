// Spatial binning of a point batch: counting sort by coarse tile (T^3 tiles over
// the submap bound).  Points of a tile become contiguous, which (a) makes the
// corner gathers of neighbouring lanes hit the same L2 lines and (b) lets the
// backward form the grid gradient owner-computes, one wavefront per tile
// (grad_pull.hip), instead of scattering it with float atomics.
//
// Three small launches, no global atomics and nothing to zero:
//   hist    <= 64 blocks of 1024 threads: LDS histogram of the block's slice of the
//           batch -> bh[block][tile], tile_id[i]
//   prefix  one thread per tile: exclusive prefix over blocks written back into bh,
//           total -> count[tile]
//   scatter every block re-derives the exclusive scan of count in LDS (block 0
//           publishes it as tile_offsets), seeds its cursors with tile_off + its bh
//           prefix, then perm[pos] = i, xn[pos] = normalised x[i] (and xs[pos] = x[i]
//           when asked).  Blocks own disjoint runs, so only the order inside one
//           (block, tile) run depends on LDS atomic timing; that order touches the
//           fp32 summation order of the pull backward and nothing else.
// The reference has no counterpart (it samples every level with independent
// random gathers, grid_opt/models/grid_modules.py:86-94).
#include <stdlib.h>

#include "common.hpp"

namespace miso {

struct Tiles3 { int t[3]; };   // tiles per axis (x, y, z)

__device__ __forceinline__ int tile_of(float px, float py, float pz, const GridK& g, const Tiles3& T) {
  float u[3] = {px, py, pz};
  int t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = (g.flags & MISO_F_COORDS_NORMALIZED) ? 0.5f * (u[a] + 1.0f)
                                                    : (u[a] - g.bmin[a]) / (g.bmax[a] - g.bmin[a]);
    v = v * (float)T.t[a];
    if (!(v == v)) v = 0.0f;
    v = fminf(fmaxf(floorf(v), 0.0f), (float)(T.t[a] - 1));
    t[a] = (int)v;
  }
  return (t[2] * T.t[1] + t[1]) * T.t[0] + t[0];
}

constexpr int SORT_THREADS = 1024;
constexpr int SORT_MAX_BLOCKS = 256;

__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                                Tiles3 T, int64_t seg, int* __restrict__ bh,
                                                                uint16_t* __restrict__ tile_id) {
  extern __shared__ int hist[];
  const int nt = T.t[0] * T.t[1] * T.t[2];
  for (int i = threadIdx.x; i < nt; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * seg, hi = min(n, lo + seg);
  constexpr int UP = 4;      // independent points per trip (loads in flight together)
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)UP * blockDim.x) {
    float v[UP][3];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < hi) { v[u][0] = x[i * 3]; v[u][1] = x[i * 3 + 1]; v[u][2] = x[i * 3 + 2]; }
    }
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (i < hi) {
        const int t = tile_of(v[u][0], v[u][1], v[u][2], g, T);
        tile_id[i] = (uint16_t)t;
        atomicAdd(&hist[t], 1);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt; i += blockDim.x) bh[(int64_t)blockIdx.x * nt + i] = hist[i];
}

// 64 tiles per workgroup, lane = tile; the blocks' histograms are split over the PREFIX_WAVES waves:
// bh[b][t] <- sum_{b' < b} bh[b'][t]; count[t] <- column total
constexpr int PREFIX_WAVES = 16;
__global__ __launch_bounds__(64 * PREFIX_WAVES) void sort_prefix_kernel(int* __restrict__ bh, int nb, int nt,
                                                                       int* __restrict__ count) {
  __shared__ int part[PREFIX_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const int per = (nb + PREFIX_WAVES - 1) / PREFIX_WAVES, b0 = min(nb, wave * per), b1 = min(nb, b0 + per);
  int sum = 0;
  if (t < nt) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      int v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bh[(int64_t)(b + k) * nt + t];
#pragma unroll
      for (int k = 0; k < 8; ++k) sum += v[k];
    }
    for (; b < b1; ++b) sum += bh[(int64_t)b * nt + t];
  }
  part[wave][lane] = sum;
  __syncthreads();
  int run = 0;
  for (int w = 0; w < wave; ++w) run += part[w][lane];
  if (t < nt) {
    int b = b0;
    for (; b + 8 <= b1; b += 8) {
      int v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bh[(int64_t)(b + k) * nt + t];
#pragma unroll
      for (int k = 0; k < 8; ++k) { bh[(int64_t)(b + k) * nt + t] = run; run += v[k]; }
    }
    for (; b < b1; ++b) { int v = bh[(int64_t)b * nt + t]; bh[(int64_t)b * nt + t] = run; run += v; }
    if (wave == PREFIX_WAVES - 1) count[t] = run;
  }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                                   int nt, int64_t seg, const int* __restrict__ bh,
                                                                   const int* __restrict__ count,
                                                                   const uint16_t* __restrict__ tile_id,
                                                                   float* __restrict__ xs, float* __restrict__ xn,
                                                                   int* __restrict__ perm,
                                                                   int* __restrict__ tile_off) {
  extern __shared__ int cursor[];      // nt cursors, then 16 wave sums
  int* wsum = cursor + nt;
  // exclusive scan of count: contiguous chunk per thread, wave scan, scan of the wave sums
  const int per = (nt + SORT_THREADS - 1) / SORT_THREADS;
  const int b0 = threadIdx.x * per;
  int sum = 0;
  // (this block's prefixes are requested together with the counts: one round trip in front of the scan, not one more
  // behind it; 16^3 tiles: four per thread)
  constexpr int PER_REG = 4;
  int cnt_r[PER_REG], bh_r[PER_REG];
#pragma unroll
  for (int i = 0; i < PER_REG; ++i) {
    cnt_r[i] = bh_r[i] = 0;
    if (per <= PER_REG && i < per && b0 + i < nt) { cnt_r[i] = count[b0 + i]; bh_r[i] = bh[(int64_t)blockIdx.x * nt + b0 + i]; }
  }
  if (per <= PER_REG) {
#pragma unroll
    for (int i = 0; i < PER_REG; ++i) sum += cnt_r[i];
  } else {
    for (int i = 0; i < per; ++i) if (b0 + i < nt) sum += count[b0 + i];
  }
  int inc = sum;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wsum[w];
  int run = wbase + inc - sum;
  if (per <= PER_REG) {
#pragma unroll
    for (int i = 0; i < PER_REG; ++i)
      if (i < per && b0 + i < nt) {
        cursor[b0 + i] = run + bh_r[i];
        if (blockIdx.x == 0) tile_off[b0 + i] = run;
        run += cnt_r[i];
      }
  } else {
    for (int i = 0; i < per; ++i)
      if (b0 + i < nt) {
        cursor[b0 + i] = run + bh[(int64_t)blockIdx.x * nt + b0 + i];
        if (blockIdx.x == 0) tile_off[b0 + i] = run;
        run += count[b0 + i];
      }
  }
  if (blockIdx.x == 0 && threadIdx.x == SORT_THREADS - 1) tile_off[nt] = run;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * seg, hi = min(n, lo + seg);
  // four points per thread per trip: their loads, LDS atomics and stores are independent, so they are
  // issued together instead of as four dependent round trips
  constexpr int UP = 4;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += (int64_t)UP * blockDim.x) {
    int t[UP];
    float v[UP][3];
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      t[u] = -1;
      if (i < hi) {
        t[u] = tile_id[i];
        v[u][0] = x[i * 3 + 0]; v[u][1] = x[i * 3 + 1]; v[u][2] = x[i * 3 + 2];
      }
    }
    int pos[UP];
#pragma unroll
    for (int u = 0; u < UP; ++u) pos[u] = (t[u] >= 0) ? atomicAdd(&cursor[t[u]], 1) : 0;
#pragma unroll
    for (int u = 0; u < UP; ++u) {
      if (t[u] < 0) continue;
      const int64_t i = i0 + (int64_t)u * blockDim.x;
      if (perm) perm[pos[u]] = (int)i;      // (optional: the index also rides in xn[pos].w)
      if (xs) {
        xs[(int64_t)pos[u] * 3 + 0] = v[u][0]; xs[(int64_t)pos[u] * 3 + 1] = v[u][1];
        xs[(int64_t)pos[u] * 3 + 2] = v[u][2];
      }
      if (xn) {
        // normalised coordinates exactly as common.hpp:axis_coord forms them, one float4 per
        // point: the sorted kernels read these and never repeat the division
        float w[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
          w[a] = (g.flags & MISO_F_COORDS_NORMALIZED)
                     ? v[u][a]
                     : __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(v[u][a], g.bmin[a])), __fsub_rn(g.bmax[a], g.bmin[a])), 1.0f);
        // .w: the point's original index as an integer bit pattern -- a consumer that has the float4 in hand anyway
        // (sdf_train_kernel) needs no perm[] then, and the sort one 4-byte scattered store per point less
        reinterpret_cast<float4*>(xn)[pos[u]] = make_float4(w[0], w[1], w[2], __int_as_float((int)i));
      }
    }
  }
}

static inline int sort_blocks(int64_t n) {
  constexpr int per = 2048;      // points per block (512 / 1024 / 2048 / 4096 measured: 21.4 / 21.1 / 20.1 / 22.9 us)
  int64_t b = (n + per - 1) / per;
  if (b > SORT_MAX_BLOCKS) b = SORT_MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

static inline int64_t a256(int64_t v) { return (v + 255) / 256 * 256; }

int64_t sort_workspace_bytes(int64_t n, int tiles) {
  int T3[3];
  if (!tiles_xyz(tiles, T3)) return 0;
  int64_t nt = (int64_t)T3[0] * T3[1] * T3[2];
  return a256((int64_t)sort_blocks(n) * nt * sizeof(int)) + a256(nt * sizeof(int)) + a256(n * 2);
}

hipError_t launch_sort(const GridK& g, const float* x, int64_t n, int tiles, void* ws, float* xs, float* xn,
                       int* perm, int* tile_off, hipStream_t s) {
  Tiles3 T;
  if (!tiles_xyz(tiles, T.t)) return hipErrorInvalidValue;
  const int nt = T.t[0] * T.t[1] * T.t[2];
  const int nb = sort_blocks(n);
  const int64_t seg = (n + nb - 1) / nb;
  char* w = reinterpret_cast<char*>(ws);
  int* bh = reinterpret_cast<int*>(w);            w += a256((int64_t)nb * nt * sizeof(int));
  int* count = reinterpret_cast<int*>(w);         w += a256(nt * sizeof(int));
  uint16_t* tid = reinterpret_cast<uint16_t*>(w);
  {   // a fine binning's histogram (32^3 tiles: 128 KB) is past the default dynamic-LDS limit
    hipError_t e = allow_dynamic_lds((const void*)sort_hist_kernel, (size_t)nt * sizeof(int));
    if (e == hipSuccess) e = allow_dynamic_lds((const void*)sort_scatter_kernel, (size_t)(nt + 16) * sizeof(int));
    if (e != hipSuccess) return e;
  }
  sort_hist_kernel<<<nb, SORT_THREADS, nt * sizeof(int), s>>>(g, x, n, T, seg, bh, tid);
  sort_prefix_kernel<<<(nt + 63) / 64, 64 * PREFIX_WAVES, 0, s>>>(bh, nb, nt, count);
  sort_scatter_kernel<<<nb, SORT_THREADS, (nt + 16) * sizeof(int), s>>>(g, x, n, nt, seg, bh, count, tid, xs, xn,
                                                                        perm, tile_off);
  return hipGetLastError();
}

}  // namespace miso
