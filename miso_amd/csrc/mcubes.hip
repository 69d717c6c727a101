// Marching cubes on a dense fp32 volume u[x][y][z] (z fastest) -- the step after the path
// (SURVEY 8f-2): the reference copies the res^3 SDF volume of extract_fields to the host and calls
// mcubes.marching_cubes (grid_opt/utils/utils_sdf.py:89-101).  Here the volume stays in HBM, is read
// once, and nothing is sorted: everything after the first sweep works on 64-bit words, one bit per sample.
//
// A sample ROW is the nz samples of one (x, y), row id = x*ny + y; a CHUNK is 64 consecutive z, one bit per
// sample in a 64-bit word, one sample per lane of a wave.
//   mc_signs_kernel     the only sweep over the volume: a wave ballot of (u < iso) per (row, chunk) -> the sign
//                       bitmap (HBM-bound, 4 B per sample read).
//   mc_edges_kernel     one THREAD per (row, chunk), 64-bit logic on the sign words of the four rows (x,y) (x+1,y)
//                       (x,y+1) (x+1,y+1) and their one-bit shifts (z+1):
//                       * the crossed lattice edges out of each sample along x / y / z are XORs of sign words;
//                         these three words per (row, chunk) are the VERTEX bitmap, their popcounts the counts;
//                       * the cells the surface touches are the bits where the eight corner words disagree; only
//                         those are walked for their sign case -> triangle count (case table);
//                       * chunks with triangles / with vertices are appended to two work lists (one atomic per
//                         wave; the order does not matter, every chunk knows its output offsets).
//   (the caller turns the counts into exclusive offsets: two cumsums)
//   mc_emit_kernel      one wave per listed chunk, one cell per lane: cases again from the sign words, a wave scan
//                       of the triangle counts, and vertex INDICES written directly: the index of the vertex on an
//                       edge is offset[word] + popcount(word below its bit) -- vertices are numbered by
//                       (row, axis, z), i.e. by the key ((x*ny + y)*3 + axis)*nz + z.
//   mc_vertices_kernel  one wave per listed chunk: each set bit's lane places its vertex at the linear crossing
//                       of u - iso.
// Triangles come out in cell order (x-major), a cell's triangles in table order: the list a serial sweep makes.
// Scratch: 5 bits per sample (+ the counts / offsets).  Written: 24 B per triangle, 12 B per vertex.
#include "common.hpp"

namespace miso {
namespace {

__device__ __attribute__((aligned(16))) const int8_t kMcTable[256][16] = {
#include "mc_table.inc"
};

constexpr int MC_SIGN_WORDS = 4;   // independent loads per lane in flight in mc_signs_kernel

struct McDims {
  int32_t nx, ny, nz;
  int32_t n_rows;     // nx * ny
  int32_t n_chunks;   // ceil(nz / 64)
  int32_t any_cells;  // every axis has at least two samples
};

struct McWorkspace {
  uint64_t* signs;    // [n_rows][n_chunks]
  uint64_t* words;    // [n_rows][3][n_chunks]
  int32_t* tri_list;  // (row, chunk) ids with triangles, counts[4W] of them
  int32_t* vert_list; // (row, chunk) ids with vertices, counts[4W + 1] of them
};

__host__ __device__ inline McWorkspace mc_carve(void* base, const McDims& d) {
  McWorkspace w;
  const int64_t n_sign = (int64_t)d.n_rows * d.n_chunks;
  w.signs = reinterpret_cast<uint64_t*>(base);
  w.words = w.signs + n_sign;
  w.tri_list = reinterpret_cast<int32_t*>(w.words + 3 * n_sign);
  w.vert_list = w.tri_list + n_sign;
  return w;
}

__global__ __launch_bounds__(256) void mc_signs_kernel(const float* __restrict__ u, McDims d, float iso,
                                                       uint64_t* __restrict__ signs, int32_t* __restrict__ counts) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x < 2) counts[4 * d.n_rows * d.n_chunks + threadIdx.x] = 0;   // work-list lengths
  // one wave per sample row: no index divisions, MC_SIGN_WORDS independent 256-byte loads in flight per wave
  const int32_t row = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (row >= d.n_rows) return;
  const float* __restrict__ src = u + (int64_t)row * d.nz;
  uint64_t* __restrict__ dst = signs + (int64_t)row * d.n_chunks;
  for (int c0 = 0; c0 < d.n_chunks; c0 += MC_SIGN_WORDS) {
    bool below[MC_SIGN_WORDS];
#pragma unroll
    for (int k = 0; k < MC_SIGN_WORDS; ++k) {
      const int32_t z = (c0 + k) * 64 + lane;
      below[k] = z < d.nz && src[z] < iso;
    }
#pragma unroll
    for (int k = 0; k < MC_SIGN_WORDS; ++k) {
      const uint64_t word = __ballot(below[k]);
      if (lane == 0 && c0 + k < d.n_chunks) dst[c0 + k] = word;
    }
  }
}

// Sign words around cell row (x, y), chunk ch: a.. at z, t.. at z + 1 (one-bit shift with the next chunk's carry);
// index [dx + 2*dy].  cells = lanes whose cell (x, y, z) exists.
struct McCellWords {
  uint64_t a[4], t[4];
  uint64_t in_z, in_z1, cells;   // lanes with z / z + 1 inside the row; lanes whose cell exists
};

__device__ __forceinline__ McCellWords mc_cell_words(const McDims& d, const uint64_t* __restrict__ signs, int32_t row,
                                                     bool hx, bool hy, int ch) {
  McCellWords c;
  const int64_t o[4] = {0, (int64_t)d.ny * d.n_chunks, d.n_chunks, (int64_t)(d.ny + 1) * d.n_chunks};
  const bool have[4] = {true, hx, hy, hx && hy};
  const bool nxt = ch + 1 < d.n_chunks;
  const uint64_t* s = signs + (int64_t)row * d.n_chunks + ch;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    c.a[k] = have[k] ? s[o[k]] : 0ull;
    const uint64_t b = (have[k] && nxt) ? s[o[k] + 1] : 0ull;
    c.t[k] = (c.a[k] >> 1) | (b << 63);
  }
  const int32_t left = d.nz - ch * 64;                    // samples from this chunk's first to the row's end
  c.in_z = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
  c.in_z1 = left - 1 >= 64 ? ~0ull : ((1ull << (left - 1)) - 1ull);
  c.cells = (hx && hy) ? c.in_z1 : 0ull;
  return c;
}

__device__ __forceinline__ int mc_case_of_lane(const McCellWords& c, int lane) {
  int v = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v |= (int)((c.a[k] >> lane) & 1ull) << k;
    v |= (int)((c.t[k] >> lane) & 1ull) << (4 + k);
  }
  return ((c.cells >> lane) & 1ull) ? v : 0;
}

// The case table staged in LDS (4 KB): 256 threads copy one 16-byte row each.  Table reads then cost an LDS
// round trip instead of a global one -- they sit in the dependent chains of both kernels below.
__device__ __forceinline__ void mc_stage_table(int8_t (*tab)[16]) {
  reinterpret_cast<int4*>(&tab[0][0])[threadIdx.x] = reinterpret_cast<const int4*>(&kMcTable[0][0])[threadIdx.x];
  __syncthreads();
}

// counts: [0, 3W) vertices per vertex word in (row, axis, chunk) order, [3W, 4W) triangles per (row, chunk),
// [4W] / [4W+1] lengths of the two work lists; W = n_rows * n_chunks.
// offsets: exclusive prefix sums of the first two segments, same layout.
// Appends `value` of every active thread of the block to list[*length ...]: ranks inside the wave by ballot,
// inside the block through an LDS counter, and ONE global atomic per block (all waves of a launch hitting one
// address cost ~5 ns each).  slot: LDS scratch {block count, block base}.  Call from all 256 threads.
__device__ __forceinline__ void mc_append(bool active, int32_t* __restrict__ length, int32_t* __restrict__ list,
                                          int32_t value, int32_t* slot) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x == 0) slot[0] = 0;
  __syncthreads();
  const uint64_t vote = __ballot(active);
  int32_t wave_base = 0;
  if (vote != 0ull) {
    const int leader = __builtin_ctzll(vote);
    if (lane == leader) wave_base = atomicAdd(&slot[0], (int32_t)__popcll(vote));
    wave_base = __shfl(wave_base, leader, 64);
  }
  __syncthreads();
  if (threadIdx.x == 0) slot[1] = slot[0] ? atomicAdd(length, slot[0]) : 0;
  __syncthreads();
  if (active) list[slot[1] + wave_base + __popcll(vote & ((1ull << lane) - 1ull))] = value;
}

__global__ __launch_bounds__(256) void mc_edges_kernel(McDims d, McWorkspace ws, int32_t* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) int8_t tab[256][16];
  mc_stage_table(tab);
  const int32_t n_words = d.n_rows * d.n_chunks;          // < 2^31 / 64
  const int32_t w = blockIdx.x * 256 + threadIdx.x;
  int n_tri = 0;
  bool any_vertex = false;
  if (w < n_words) {
    const int32_t row = w / d.n_chunks;
    const int ch = w - row * d.n_chunks;
    const int32_t x = row / d.ny, y = row - x * d.ny;
    const bool hx = x + 1 < d.nx, hy = y + 1 < d.ny;
    const McCellWords c = mc_cell_words(d, ws.signs, row, hx, hy, ch);
    const uint64_t live = d.any_cells ? ~0ull : 0ull;
    const uint64_t wx = hx ? ((c.a[0] ^ c.a[1]) & c.in_z & live) : 0ull;
    const uint64_t wy = hy ? ((c.a[0] ^ c.a[2]) & c.in_z & live) : 0ull;
    const uint64_t wz = (c.a[0] ^ c.t[0]) & c.in_z1 & live;
    // the surface touches cell z of this chunk only if some pair of its corners differs
    uint64_t mixed = ((c.a[0] ^ c.a[1]) | (c.a[0] ^ c.a[2]) | (c.a[0] ^ c.a[3]) | (c.a[0] ^ c.t[0]) |
                      (c.a[0] ^ c.t[1]) | (c.a[0] ^ c.t[2]) | (c.a[0] ^ c.t[3])) & c.cells;
    while (mixed != 0ull) {
      n_tri += tab[mc_case_of_lane(c, __builtin_ctzll(mixed))][15];
      mixed &= mixed - 1ull;
    }
    const int64_t v = ((int64_t)row * 3) * d.n_chunks + ch;
    ws.words[v] = wx;
    ws.words[v + d.n_chunks] = wy;
    ws.words[v + 2 * d.n_chunks] = wz;
    counts[v] = __popcll(wx);
    counts[v + d.n_chunks] = __popcll(wy);
    counts[v + 2 * d.n_chunks] = __popcll(wz);
    counts[3 * (int64_t)n_words + w] = n_tri;
    any_vertex = (wx | wy | wz) != 0ull;
  }
  __shared__ int32_t slot[4];
  mc_append(n_tri > 0, counts + 4 * (int64_t)n_words, ws.tri_list, w, slot);
  mc_append(any_vertex, counts + 4 * (int64_t)n_words + 1, ws.vert_list, w, slot + 2);
}

// index of the vertex on the edge out of sample (row, z) along `axis`
__device__ __forceinline__ int64_t mc_vertex_index(const McDims& d, const McWorkspace& ws,
                                                   const int64_t* __restrict__ offsets, int32_t row, int axis,
                                                   int32_t z) {
  const int64_t wi = ((int64_t)row * 3 + axis) * d.n_chunks + (z >> 6);
  const uint64_t below = ws.words[wi] & ((1ull << (z & 63)) - 1ull);
  return offsets[wi] + __popcll(below);
}

__global__ __launch_bounds__(256) void mc_emit_kernel(McDims d, McWorkspace ws, const int64_t* __restrict__ offsets,
                                                      int32_t n_listed, int64_t capacity, int64_t* __restrict__ faces) {
  __shared__ __attribute__((aligned(16))) int8_t tab[256][16];
  __shared__ uint16_t owner[4][5 * 64];     // per wave: triangle j of the chunk -> cell lane | its t << 6
  __shared__ uint8_t case_of[4][64];
  mc_stage_table(tab);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n_words = (int64_t)d.n_rows * d.n_chunks;
  const int32_t item = blockIdx.x * 4 + wave;
  const bool live = item < n_listed;
  const int32_t w = ws.tri_list[live ? item : 0];
  const int32_t row = w / d.n_chunks;
  const int ch = w - row * d.n_chunks;
  const McCellWords cw = mc_cell_words(d, ws.signs, row, true, true, ch);   // triangles => the cells exist
  const int c = mc_case_of_lane(cw, lane);
  const int n = tab[c][15];
  int incl = n;
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  const int total = __shfl(incl, 63, 64);
  // hand the chunk's triangles out one per lane: the cells that own them are few and far between, their
  // triangles are consecutive in the output
  case_of[wave][lane] = (uint8_t)c;
  for (int t = 0; t < n; ++t) owner[wave][incl - n + t] = (uint16_t)(lane | (t << 6));
  __syncthreads();
  if (!live) return;
  const int64_t tri0 = offsets[3 * n_words + w];
  for (int j = lane; j < total; j += 64) {
    const int64_t tri = tri0 + j;
    if (tri >= capacity) break;
    const int own = owner[wave][j];
    const int cell = own & 63, t = own >> 6;
    const int cc = case_of[wave][cell];
    const int32_t z = ch * 64 + cell;
    int64_t idx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = tab[cc][3 * t + k];
      const int axis = e >> 2, a = e & 1, b = (e >> 1) & 1;
      // low sample of the edge: the two coordinates other than `axis`, in increasing axis order, take (a, b)
      const int dx = axis == 0 ? 0 : a;
      const int dy = axis == 0 ? a : (axis == 1 ? 0 : b);
      const int dz = axis == 2 ? 0 : b;
      idx[k] = mc_vertex_index(d, ws, offsets, row + dx * d.ny + dy, axis, z + dz);
    }
    faces[3 * tri + 0] = idx[0];
    faces[3 * tri + 1] = idx[1];
    faces[3 * tri + 2] = idx[2];
  }
}

__global__ __launch_bounds__(256) void mc_vertices_kernel(const float* __restrict__ u, McDims d, float iso,
                                                          McWorkspace ws, const int64_t* __restrict__ offsets,
                                                          int32_t n_listed, int64_t capacity,
                                                          float* __restrict__ verts) {
  const int lane = threadIdx.x & 63;
  const int32_t item = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= n_listed) return;
  const int32_t w = ws.vert_list[item];
  const int32_t row = w / d.n_chunks;
  const int ch = w - row * d.n_chunks;
  const int32_t x = row / d.ny, y = row - x * d.ny, z = ch * 64 + lane;
  const int64_t lin = (int64_t)row * d.nz + z;
#pragma unroll
  for (int axis = 0; axis < 3; ++axis) {
    const int64_t wi = ((int64_t)row * 3 + axis) * d.n_chunks + ch;
    const uint64_t word = ws.words[wi];
    if (!((word >> lane) & 1ull)) continue;
    const int64_t v = offsets[wi] + __popcll(word & ((1ull << lane) - 1ull));
    if (v >= capacity) continue;
    const int64_t step = axis == 0 ? (int64_t)d.ny * d.nz : (axis == 1 ? d.nz : 1);
    const float ua = u[lin], ub = u[lin + step];
    const float t = (iso - ua) / (ub - ua);
    verts[3 * v + 0] = (float)x + (axis == 0 ? t : 0.0f);
    verts[3 * v + 1] = (float)y + (axis == 1 ? t : 0.0f);
    verts[3 * v + 2] = (float)z + (axis == 2 ? t : 0.0f);
  }
}

McDims make_dims(int32_t nx, int32_t ny, int32_t nz) {
  McDims d;
  d.nx = nx; d.ny = ny; d.nz = nz;
  d.n_rows = nx * ny;
  d.n_chunks = (nz + 63) / 64;
  d.any_cells = nx > 1 && ny > 1 && nz > 1;
  return d;
}

}  // namespace

int64_t mc_words(int32_t nx, int32_t ny, int32_t nz) {
  return (int64_t)nx * ny * ((nz + 63) / 64);
}

int64_t mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz) {
  return mc_words(nx, ny, nz) * (8 + 3 * 8 + 2 * 4);
}

hipError_t launch_mc_classify(const float* u, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                              int32_t* counts, hipStream_t s) {
  const McDims d = make_dims(nx, ny, nz);
  const McWorkspace ws = mc_carve(workspace, d);
  const int64_t n_words = mc_words(nx, ny, nz);
  mc_signs_kernel<<<(uint32_t)((d.n_rows + 3) / 4), 256, 0, s>>>(u, d, iso, ws.signs, counts);
  mc_edges_kernel<<<(uint32_t)((n_words + 255) / 256), 256, 0, s>>>(d, ws, counts);
  return hipGetLastError();
}

hipError_t launch_mc_emit(int32_t nx, int32_t ny, int32_t nz, void* workspace, const int64_t* offsets,
                          int32_t n_listed, int64_t capacity, int64_t* faces, hipStream_t s) {
  if (capacity == 0 || n_listed == 0) return hipSuccess;
  const McDims d = make_dims(nx, ny, nz);
  mc_emit_kernel<<<(uint32_t)((n_listed + 3) / 4), 256, 0, s>>>(d, mc_carve(workspace, d), offsets, n_listed, capacity,
                                                                faces);
  return hipGetLastError();
}

hipError_t launch_mc_vertices(const float* u, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                              const int64_t* offsets, int32_t n_listed, int64_t capacity, float* verts,
                              hipStream_t s) {
  if (capacity == 0 || n_listed == 0) return hipSuccess;
  const McDims d = make_dims(nx, ny, nz);
  mc_vertices_kernel<<<(uint32_t)((n_listed + 3) / 4), 256, 0, s>>>(u, d, iso, mc_carve(workspace, d), offsets,
                                                                    n_listed, capacity, verts);
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
