// Spatial binning of a point batch: counting sort by coarse tile (T^3 tiles over
// the submap bound).  Points of a tile become contiguous, which (a) makes the
// corner gathers of neighbouring lanes hit the same L2 lines and (b) lets the
// backward pre-reduce the coarse levels' gradient in LDS per tile before it
// touches the L2 atomics (sdf_fused.hip, tiled variant).
//
// Three small launches (+ one memset):
//   hist    per-block LDS histogram of tile ids -> bh[block][tile], tile_id[i], count[tile]
//   scan    exclusive prefix of count -> tile_off (and the scatter cursors)
//   scatter every block reserves its run per tile (one returning atomic per (block,
//           tile)), then perm[pos] = i, xs[pos] = x[i].  The order inside a tile is
//           not deterministic; only fp32 summation order depends on it.
// The reference has no counterpart (it samples every level with independent
// random gathers, grid_opt/models/grid_modules.py:86-94).
#include "common.hpp"

namespace miso {

__device__ __forceinline__ int tile_of(float px, float py, float pz, const GridK& g, int T) {
  float u[3] = {px, py, pz};
  int t[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float v = (g.flags & MISO_F_COORDS_NORMALIZED) ? 0.5f * (u[a] + 1.0f)
                                                    : (u[a] - g.bmin[a]) / (g.bmax[a] - g.bmin[a]);
    v = v * (float)T;
    if (!(v == v)) v = 0.0f;
    v = fminf(fmaxf(floorf(v), 0.0f), (float)(T - 1));
    t[a] = (int)v;
  }
  return (t[2] * T + t[1]) * T + t[0];
}

__global__ __launch_bounds__(256) void sort_hist_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                       int T, int64_t seg, int* __restrict__ bh,
                                                       int* __restrict__ count,
                                                       uint16_t* __restrict__ tile_id) {
  extern __shared__ int hist[];
  const int nt = T * T * T;
  for (int i = threadIdx.x; i < nt; i += blockDim.x) hist[i] = 0;
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * seg, hi = min(n, lo + seg);
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int t = tile_of(x[i * 3], x[i * 3 + 1], x[i * 3 + 2], g, T);
    tile_id[i] = (uint16_t)t;
    atomicAdd(&hist[t], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nt; i += blockDim.x) {
    int v = hist[i];
    bh[(int64_t)blockIdx.x * nt + i] = v;
    if (v) atomicAdd(&count[i], v);
  }
}

// one block: exclusive scan of the per-tile counts -> tile_off; cursor = tile_off
__global__ __launch_bounds__(1024) void sort_scan_kernel(const int* __restrict__ count, int nt,
                                                        int* __restrict__ tile_off,
                                                        int* __restrict__ cursor) {
  extern __shared__ int sm[];  // nt
  // per-thread serial chunk + block scan of the chunk sums
  const int per = (nt + blockDim.x - 1) / blockDim.x;
  const int b0 = threadIdx.x * per;
  int sum = 0;
  for (int i = 0; i < per; ++i) if (b0 + i < nt) sum += count[b0 + i];
  sm[threadIdx.x] = sum;
  __syncthreads();
  for (int o = 1; o < (int)blockDim.x; o <<= 1) {
    int v = (threadIdx.x >= (unsigned)o) ? sm[threadIdx.x - o] : 0;
    __syncthreads();
    sm[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sm[threadIdx.x] - sum;  // exclusive prefix of this thread's chunk
  for (int i = 0; i < per; ++i)
    if (b0 + i < nt) {
      tile_off[b0 + i] = run;
      cursor[b0 + i] = run;
      run += count[b0 + i];
    }
  if (threadIdx.x == blockDim.x - 1) tile_off[nt] = sm[threadIdx.x];
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(GridK g, const float* __restrict__ x, int64_t n,
                                                          int nt, int64_t seg, const int* __restrict__ bh,
                                                          int* __restrict__ gcursor,
                                                          const uint16_t* __restrict__ tile_id,
                                                          float* __restrict__ xs, float* __restrict__ xn,
                                                          int* __restrict__ perm) {
  extern __shared__ int cursor[];
  // reserve this block's run inside every tile with one returning atomic per (block, tile)
  for (int i = threadIdx.x; i < nt; i += blockDim.x) {
    int v = bh[(int64_t)blockIdx.x * nt + i];
    cursor[i] = v ? atomicAdd(&gcursor[i], v) : 0;
  }
  __syncthreads();
  const int64_t lo = (int64_t)blockIdx.x * seg, hi = min(n, lo + seg);
  for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
    int t = tile_id[i];
    int pos = atomicAdd(&cursor[t], 1);
    perm[pos] = (int)i;
    xs[(int64_t)pos * 3 + 0] = x[i * 3 + 0];
    xs[(int64_t)pos * 3 + 1] = x[i * 3 + 1];
    xs[(int64_t)pos * 3 + 2] = x[i * 3 + 2];
    if (xn) {
      // normalised coordinates exactly as common.hpp:axis_coord forms them, one float4 per
      // point, so the pull backward (grad_pull.hip) lands on the same cells with one load
      float v[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        v[a] = x[i * 3 + a];
        if (!(g.flags & MISO_F_COORDS_NORMALIZED))
          v[a] = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(v[a], g.bmin[a])), __fsub_rn(g.bmax[a], g.bmin[a])), 1.0f);
      }
      reinterpret_cast<float4*>(xn)[pos] = make_float4(v[0], v[1], v[2], 0.0f);
    }
  }
}

static inline int sort_blocks(int64_t n) {
  int64_t b = (n + 1023) / 1024;
  if (b > 256) b = 256;
  if (b < 1) b = 1;
  return (int)b;
}

static inline int64_t a256(int64_t v) { return (v + 255) / 256 * 256; }

int64_t sort_workspace_bytes(int64_t n, int T) {
  int64_t nt = (int64_t)T * T * T;
  return a256((int64_t)sort_blocks(n) * nt * sizeof(int)) + 2 * a256(nt * sizeof(int)) + a256(n * 2);
}

hipError_t launch_sort(const GridK& g, const float* x, int64_t n, int T, void* ws, float* xs, float* xn,
                       int* perm, int* tile_off, hipStream_t s) {
  const int nt = T * T * T;
  const int nb = sort_blocks(n);
  const int64_t seg = (n + nb - 1) / nb;
  char* w = reinterpret_cast<char*>(ws);
  int* bh = reinterpret_cast<int*>(w);            w += a256((int64_t)nb * nt * sizeof(int));
  int* count = reinterpret_cast<int*>(w);         w += a256(nt * sizeof(int));
  int* cursor = reinterpret_cast<int*>(w);        w += a256(nt * sizeof(int));
  uint16_t* tid = reinterpret_cast<uint16_t*>(w);
  hipError_t e = hipMemsetAsync(count, 0, nt * sizeof(int), s);
  if (e != hipSuccess) return e;
  sort_hist_kernel<<<nb, 256, nt * sizeof(int), s>>>(g, x, n, T, seg, bh, count, tid);
  sort_scan_kernel<<<1, 1024, 1024 * sizeof(int), s>>>(count, nt, tile_off, cursor);
  sort_scatter_kernel<<<nb, 256, nt * sizeof(int), s>>>(g, x, n, nt, seg, bh, cursor, tid, xs, xn, perm);
  return hipGetLastError();
}

}  // namespace miso
