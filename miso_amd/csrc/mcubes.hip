// Marching cubes on a dense fp32 volume u[x][y][z] (z fastest) -- the step after the path
// (SURVEY 8f-2): the reference copies the res^3 SDF volume of extract_fields to the host and calls
// mcubes.marching_cubes (grid_opt/utils/utils_sdf.py:89-101).  Here the volume stays in HBM.
//
// Two sweeps over the cells, 256 consecutive cells (z fastest) per block, no intermediate per-cell array:
//   mc_count_kernel  sign case of every cell -> triangle count, reduced to one int per block;
//   (the caller turns the block counts into exclusive offsets: one small cumsum)
//   mc_emit_kernel   recomputes the cases, scans the counts inside the block and writes, for every triangle
//                    corner, the KEY of the lattice edge it sits on: 3 * linear_index(low sample) + axis.
// Cells are visited in x-major order and a cell's triangles in table order, so the triangle list is the
// one a serial sweep produces.  The caller welds corners by key (sort/unique) and mc_vertices_kernel places
// one vertex per unique key at the linear crossing of u - iso along its edge.
// Both sweeps are HBM-bound: 4 B per sample each (the 8 corner reads of neighbouring cells hit L1/L2),
// plus 24 B per emitted triangle.
#include "common.hpp"

namespace miso {
namespace {

__device__ const int8_t kMcTable[256][16] = {
#include "mc_table.inc"
};

constexpr int MC_BLOCK = 256;

struct McDims {
  int32_t nx, ny, nz;      // samples per axis
  int32_t cy, cz;          // cells along y, z
  int32_t n_cells;
};

// sign case of cell `cell` (bit c: corner x + 2y + 4z has u < iso) and its sample coordinates
__device__ __forceinline__ int mc_case(const float* __restrict__ u, const McDims& d, int32_t cell, float iso, int& x,
                                       int& y, int& z) {
  z = cell % d.cz;
  const int32_t r = cell / d.cz;
  y = r % d.cy;
  x = r / d.cy;
  const int64_t base = ((int64_t)x * d.ny + y) * d.nz + z;
  const int64_t sx = (int64_t)d.ny * d.nz, sy = d.nz;
  int c = 0;
  c |= (u[base] < iso) << 0;
  c |= (u[base + sx] < iso) << 1;
  c |= (u[base + sy] < iso) << 2;
  c |= (u[base + sx + sy] < iso) << 3;
  c |= (u[base + 1] < iso) << 4;
  c |= (u[base + sx + 1] < iso) << 5;
  c |= (u[base + sy + 1] < iso) << 6;
  c |= (u[base + sx + sy + 1] < iso) << 7;
  return c;
}

__global__ __launch_bounds__(MC_BLOCK) void mc_count_kernel(const float* __restrict__ u, McDims d, float iso,
                                                            int32_t* __restrict__ block_counts) {
  __shared__ int32_t wave_sum[MC_BLOCK / 64];
  const int32_t cell = blockIdx.x * MC_BLOCK + threadIdx.x;
  int n = 0;
  if (cell < d.n_cells) {
    int x, y, z;
    n = kMcTable[mc_case(u, d, cell, iso, x, y, z)][15];
  }
  for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
  if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(MC_BLOCK) void mc_emit_kernel(const float* __restrict__ u, McDims d, float iso,
                                                           const int64_t* __restrict__ block_offsets,
                                                           int64_t capacity, int64_t* __restrict__ keys) {
  __shared__ int32_t wave_sum[MC_BLOCK / 64];
  const int32_t cell = blockIdx.x * MC_BLOCK + threadIdx.x;
  int n = 0, c = 0, x = 0, y = 0, z = 0;
  if (cell < d.n_cells) {
    c = mc_case(u, d, cell, iso, x, y, z);
    n = kMcTable[c][15];
  }
  // exclusive scan of n over the block: inside the wave by shuffles, across the 4 waves through LDS
  int incl = n;
  const int lane = threadIdx.x & 63;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wave_sum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int before = 0;
  for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) before += wave_sum[w];
  if (n == 0) return;
  int64_t tri = block_offsets[blockIdx.x] + before + (incl - n);
  const int8_t* row = kMcTable[c];
  for (int t = 0; t < n; ++t, ++tri) {
    if (tri >= capacity) return;
    for (int k = 0; k < 3; ++k) {
      const int e = row[3 * t + k];
      const int axis = e >> 2, a = e & 1, b = (e >> 1) & 1;
      // low sample of the edge: the two coordinates other than `axis`, in increasing axis order, take (a, b)
      const int ex = x + (axis == 0 ? 0 : a);
      const int ey = y + (axis == 0 ? a : (axis == 1 ? 0 : b));
      const int ez = z + (axis == 2 ? 0 : b);
      const int64_t lin = ((int64_t)ex * d.ny + ey) * d.nz + ez;
      keys[3 * tri + k] = 3 * lin + axis;
    }
  }
}

__global__ __launch_bounds__(256) void mc_vertices_kernel(const float* __restrict__ u, int32_t ny, int32_t nz,
                                                          float iso, const int64_t* __restrict__ keys, int64_t n,
                                                          float* __restrict__ verts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t key = keys[i];
  const int64_t lin = key / 3;
  const int axis = (int)(key - 3 * lin);
  const int64_t xy = lin / nz;
  float p[3];
  p[2] = (float)(lin - xy * nz);
  p[0] = (float)(xy / ny);
  p[1] = (float)(xy - (xy / ny) * ny);
  const int64_t step = axis == 0 ? (int64_t)ny * nz : (axis == 1 ? nz : 1);
  const float ua = u[lin], ub = u[lin + step];
  const float t = (iso - ua) / (ub - ua);
  p[axis] += t;
  verts[3 * i + 0] = p[0];
  verts[3 * i + 1] = p[1];
  verts[3 * i + 2] = p[2];
}

McDims make_dims(int32_t nx, int32_t ny, int32_t nz) {
  McDims d;
  d.nx = nx; d.ny = ny; d.nz = nz;
  d.cy = ny - 1; d.cz = nz - 1;
  d.n_cells = (nx - 1) * (ny - 1) * (nz - 1);
  return d;
}

}  // namespace

int64_t mc_blocks(int32_t nx, int32_t ny, int32_t nz) {
  const int64_t cells = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  return (cells + MC_BLOCK - 1) / MC_BLOCK;
}

hipError_t launch_mc_count(const float* u, int32_t nx, int32_t ny, int32_t nz, float iso, int32_t* block_counts,
                           hipStream_t s) {
  const int64_t blocks = mc_blocks(nx, ny, nz);
  if (blocks == 0) return hipSuccess;
  mc_count_kernel<<<(uint32_t)blocks, MC_BLOCK, 0, s>>>(u, make_dims(nx, ny, nz), iso, block_counts);
  return hipGetLastError();
}

hipError_t launch_mc_emit(const float* u, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* block_offsets,
                          int64_t capacity, int64_t* keys, hipStream_t s) {
  const int64_t blocks = mc_blocks(nx, ny, nz);
  if (blocks == 0 || capacity == 0) return hipSuccess;
  mc_emit_kernel<<<(uint32_t)blocks, MC_BLOCK, 0, s>>>(u, make_dims(nx, ny, nz), iso, block_offsets, capacity, keys);
  return hipGetLastError();
}

hipError_t launch_mc_vertices(const float* u, int32_t ny, int32_t nz, float iso, const int64_t* keys, int64_t n,
                              float* verts, hipStream_t s) {
  if (n == 0) return hipSuccess;
  mc_vertices_kernel<<<(uint32_t)((n + 255) / 256), 256, 0, s>>>(u, ny, nz, iso, keys, n, verts);
  return hipGetLastError();
}

void mc_copy_table(int8_t* out) {
  static const int8_t host[256][16] = {
#include "mc_table.inc"
  };
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 16; ++j) out[16 * i + j] = host[i][j];
}

}  // namespace miso
