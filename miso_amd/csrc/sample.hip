// SDF training samples from posed depth frames, generated where the trainer reads them.
//
// Reference: PosedSdfRgbd.getitem_sdf, grid_opt/datasets/sdf_rgbd.py:381-483, which chains
//   sample_points (:221-293) -> utils_sample.get_batch_data (grid_opt/utils/utils_sample.py:142-192: depth
//   lookup, `depth != 0` / NaN-normal ray filter) -> sample_along_rays (:249-302: stratified bins :195-246,
//   surface sample + Gaussian near-surface samples clamped to [min_depth, depth + dist_behind_surf]) ->
//   bounds_ray (sdf_rgbd.py:525-534: sdf = |dir_C| (depth - z)) -> NaN-ray filter (:416-424) -> a Python loop
//   over keyframes moving every sample from the world into its keyframe (:436-445) -> truncation mask and
//   free/occupied signs (:447-455).
// That is ~40 tensor ops, two boolean compactions and one host loop with a `nonzero` per keyframe, every
// iteration.  Here: three launches (flag, check, emit), no host round trip; the compactions are the two-level
// counting scheme of sort.hip (per-block counts, every block re-derives its base from the count table).
// Random draws are INPUTS (pixel indices, uniforms, Gaussian offsets) so that the arithmetic can be pinned
// against the reference; they are indexed exactly as the reference indexes them: by the ray's position after
// the first filter.
//
// Output rows are ray-major, [surface, near x (n_surf-1), stratified x n_strat] inside a ray
// (torch.cat order, utils_sample.py:296-297).  Rows past the last valid ray are neutral padding
// (weight 0, valid 0, sign 0, sdf 0, at the position of some live sample): they add nothing to any loss or
// gradient, so a fixed-capacity batch can flow through a captured step with the live count kept on the device.
#include <algorithm>

#include "common.hpp"

namespace miso {

constexpr int RAY_BLOCK = 256;

struct RayK {
  const float* depth;      // (B,H,W)
  const float* normals;    // (B,H,W,3) or nullptr
  const float* T_WC;       // (B,4,4)
  const float* R_wk;       // (B,3,3)
  const float* t_inv;      // (B,3)   -R_wk^T t_wk, formed by the launcher's helper kernel
  const int64_t* frame_ids;
  const int64_t *pix_b, *pix_h, *pix_w;
  const float *u, *g;
  int64_t n_rays;
  int32_t B, H, W, rays_per_frame;
  float fx, fy, cx, cy;
  float min_depth, behind, trunc;
  int32_t n_strat, n_surf, S;
  float lin[MISO_RAY_MAX_BINS + 1];
  int32_t *cnt1, *cnt2, *slot;   // workspace: per-block counts of the two filters, per-ray slot after filter 1
  float* coords;
  int64_t* ids;
  float4* aux;
  float* pc_world;
  float* z_vals;
  int32_t* counts;
};

struct Ray {
  float o[3], dw[3];
  float depth, dnorm;
  int32_t b;
};

__device__ __forceinline__ int ray_frame(const RayK& k, int64_t r) {
  return k.pix_b ? (int)k.pix_b[r] : (int)(r / k.rays_per_frame);
}

// depth lookup + first filter (utils_sample.py:156-166)
__device__ __forceinline__ bool ray_first_filter(const RayK& k, int64_t r, int& b, float& depth, int& h, int& w) {
  b = ray_frame(k, r);
  h = (int)k.pix_h[r];
  w = (int)k.pix_w[r];
  const int64_t pix = ((int64_t)b * k.H + h) * k.W + w;
  depth = k.depth[pix];
  bool ok = depth != 0.0f;
  if (k.normals) { const float nx = k.normals[pix * 3]; ok = ok && !(nx != nx); }
  return ok;
}

__device__ __forceinline__ void ray_setup(const RayK& k, int b, int h, int w, float depth, Ray& ray) {
#pragma clang fp contract(off)
  // ray_dirs_C, depth_type 'z' (utils_sample.py:10-30); origin_dirs_W (:33-38)
  const float dc0 = ((float)w - k.cx) / k.fx, dc1 = ((float)h - k.cy) / k.fy, dc2 = 1.0f;
  const float* T = k.T_WC + (int64_t)b * 16;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    ray.dw[i] = (T[i * 4 + 0] * dc0 + T[i * 4 + 1] * dc1) + T[i * 4 + 2] * dc2;
    ray.o[i] = T[i * 4 + 3];
  }
  ray.depth = depth;
  ray.dnorm = sqrtf((dc0 * dc0 + dc1 * dc1) + dc2 * dc2);   // z_to_euclidean_depth, sdf_rgbd.py:527
  ray.b = b;
}

// depth of sample j along the ray (utils_sample.py:210-246, :276-297)
__device__ __forceinline__ float ray_z(const RayK& k, const Ray& ray, int64_t slot, int j) {
#pragma clang fp contract(off)
  const float max_d = ray.depth + k.behind;
  if (j < k.n_surf) {
    if (j == 0) return ray.depth;
    float z = ray.depth + k.g[slot * (k.n_surf - 1) + (j - 1)];
    if (z == z) { z = z < k.min_depth ? k.min_depth : z; z = z > max_d ? max_d : z; }   // torch.clamp keeps NaN
    return z;
  }
  const int jj = j - k.n_surf;
  const float range = max_d - k.min_depth;
  const float lower = k.lin[jj] * range + k.min_depth;
  const float bin = range / (float)k.n_strat;
  return lower + k.u[slot * k.n_strat + jj] * bin;
}

__device__ __forceinline__ void ray_point(const Ray& ray, float z, float p[3]) {
#pragma clang fp contract(off)
#pragma unroll
  for (int i = 0; i < 3; ++i) p[i] = ray.o[i] + ray.dw[i] * z;
}

// exclusive scan of a per-thread 0/1 flag over the 256-thread block; returns the block total in `total`
__device__ __forceinline__ int block_scan_flag(bool flag, int* wave_tot, int& total) {
  const unsigned long long m = __ballot(flag);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wave_tot[wave] = __popcll(m);
  __syncthreads();
  int base = 0;
  total = 0;
#pragma unroll
  for (int i = 0; i < RAY_BLOCK / 64; ++i) {
    const int c = wave_tot[i];
    if (i < wave) base += c;
    total += c;
  }
  __syncthreads();
  return base + before;
}

// sum of tab[0 .. n) by the whole block; every thread gets the result
__device__ __forceinline__ int block_sum_prefix(const int32_t* tab, int n, int* red) {
  int v = 0;
  for (int i = threadIdx.x; i < n; i += RAY_BLOCK) v += tab[i];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const int s = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return s;
}

__global__ __launch_bounds__(RAY_BLOCK) void ray_flag_kernel(RayK k) {
  __shared__ int wt[RAY_BLOCK / 64];
  const int64_t r = (int64_t)blockIdx.x * RAY_BLOCK + threadIdx.x;
  bool ok = false;
  if (r < k.n_rays) { int b, h, w; float d; ok = ray_first_filter(k, r, b, d, h, w); }
  int total;
  block_scan_flag(ok, wt, total);
  if (threadIdx.x == 0) k.cnt1[blockIdx.x] = total;
}

__global__ __launch_bounds__(RAY_BLOCK) void ray_check_kernel(RayK k) {
  __shared__ int wt[RAY_BLOCK / 64];
  __shared__ int red[RAY_BLOCK / 64];
  const int base1 = block_sum_prefix(k.cnt1, blockIdx.x, red);
  const int64_t r = (int64_t)blockIdx.x * RAY_BLOCK + threadIdx.x;
  bool ok = false;
  int b = 0, h = 0, w = 0;
  float depth = 0.0f;
  if (r < k.n_rays) ok = ray_first_filter(k, r, b, depth, h, w);
  int total1;
  const int slot = base1 + block_scan_flag(ok, wt, total1);
  bool keep = ok;
  if (ok) {   // second filter: any NaN coordinate among the ray's samples drops the ray (sdf_rgbd.py:416-419)
    Ray ray;
    ray_setup(k, b, h, w, depth, ray);
    for (int j = 0; j < k.S; ++j) {
      float p[3];
      ray_point(ray, ray_z(k, ray, slot, j), p);
      keep = keep && !(p[0] != p[0] || p[1] != p[1] || p[2] != p[2]);
    }
  }
  if (r < k.n_rays) k.slot[r] = keep ? slot : -1;
  int total2;
  block_scan_flag(keep, wt, total2);
  if (threadIdx.x == 0) {
    k.cnt2[blockIdx.x] = total2;
    if (blockIdx.x == gridDim.x - 1) k.counts[0] = base1 + total1;
  }
}

__global__ __launch_bounds__(RAY_BLOCK) void ray_emit_kernel(RayK k) {
  __shared__ int wt[RAY_BLOCK / 64];
  __shared__ int red[RAY_BLOCK / 64];
  __shared__ Ray rays[RAY_BLOCK];
  __shared__ int slots[RAY_BLOCK];
  const int base2 = block_sum_prefix(k.cnt2, blockIdx.x, red);
  const int all2 = base2 + block_sum_prefix(k.cnt2 + blockIdx.x, (int)gridDim.x - (int)blockIdx.x, red);
  const int64_t first = (int64_t)blockIdx.x * RAY_BLOCK;
  const int64_t r = first + threadIdx.x;
  const int slot = r < k.n_rays ? k.slot[r] : -1;
  int nv;
  const int local = block_scan_flag(slot >= 0, wt, nv);
  if (slot >= 0) {
    int b, h, w;
    float depth;
    ray_first_filter(k, r, b, depth, h, w);
    ray_setup(k, b, h, w, depth, rays[local]);
    slots[local] = slot;
  }
  __syncthreads();
  // blockIdx.y: a slice of the S samples of this block's rays (and of its padding rows).  With one block per 256 rays
  // and a serial loop over all their samples a keyframe-window batch (2 000 rays x 27) ran on 8 workgroups: 32 us of
  // dependent loads; sliced, every workgroup has a few rows per thread
  const int S = k.S;
  const int sj = (S + (int)gridDim.y - 1) / (int)gridDim.y, j0 = (int)blockIdx.y * sj, nj = min(sj, S - j0);
  for (int row = threadIdx.x; row < nv * nj; row += RAY_BLOCK) {
#pragma clang fp contract(off)
    const int lr = row / nj, j = j0 + (row - lr * nj);
    const Ray& ray = rays[lr];
    const float z = ray_z(k, ray, slots[lr], j);
    float p[3];
    ray_point(ray, z, p);
    const float sdf = ray.dnorm * (ray.depth - z);                 // bounds_ray, sdf_rgbd.py:526-528
    const float* R = k.R_wk + (int64_t)ray.b * 9;
    const float* ti = k.t_inv + (int64_t)ray.b * 3;
    const int64_t out = (int64_t)(base2 + lr) * S + j;
#pragma unroll
    for (int i = 0; i < 3; ++i)                                    // x R + (-R^T t)^T, utils_geometry.py:227-240
      k.coords[out * 3 + i] = ((p[0] * R[i] + p[1] * R[3 + i]) + p[2] * R[6 + i]) + ti[i];
    k.ids[out] = k.frame_ids ? k.frame_ids[ray.b] : (int64_t)ray.b;
    const float a = fabsf(sdf);
    const float sign = sdf < -k.trunc ? -1.0f : (sdf > k.trunc ? 1.0f : 0.0f);   // sdf_rgbd.py:452-455
    k.aux[out] = make_float4(sdf, a < k.trunc ? 1.0f : 0.0f, sign, 1.0f);
    if (k.pc_world) { k.pc_world[out * 3] = p[0]; k.pc_world[out * 3 + 1] = p[1]; k.pc_world[out * 3 + 2] = p[2]; }
    if (k.z_vals) k.z_vals[out] = z;
  }
  // Padding for this block's dropped rays: zero labels (no loss, no gradient) at the POSITION of a live sample --
  // a padded batch still goes through the binned trainer step, and thousands of rows parked on one point
  // would all land in one tile, which a single wave drains serially.  Borrow a ray of this block, or, when the
  // whole block is dead, probe the slot table for any live ray; only with no live ray at all do zeros remain.
  const int64_t in_block = (k.n_rays - first) < RAY_BLOCK ? (k.n_rays - first) : RAY_BLOCK;
  const int64_t pad0 = ((int64_t)all2 + (first - base2)) * S;
  const int64_t npad = (in_block - nv) * S;
  for (int64_t i = threadIdx.x + (int64_t)blockIdx.y * RAY_BLOCK; i < npad; i += (int64_t)RAY_BLOCK * gridDim.y) {
#pragma clang fp contract(off)
    const int64_t out = pad0 + i;
    Ray ray;
    int sl = -1;
    if (nv > 0) {
      const int lr = (int)(i % nv);
      ray = rays[lr];
      sl = slots[lr];
    } else if (all2 > 0) {
      uint64_t q = ((uint64_t)(first + i) * 0x9E3779B97F4A7C15ull) % (uint64_t)k.n_rays;
      for (int tries = 0; tries < 64 && sl < 0; ++tries, q = (q + 1) % (uint64_t)k.n_rays) {
        sl = k.slot[q];
        if (sl >= 0) {
          int b, h, w;
          float depth;
          ray_first_filter(k, (int64_t)q, b, depth, h, w);
          ray_setup(k, b, h, w, depth, ray);
        }
      }
    }
    float c[3] = {0.0f, 0.0f, 0.0f}, p[3] = {0.0f, 0.0f, 0.0f}, z = 0.0f;
    int64_t id = k.frame_ids ? k.frame_ids[0] : 0;
    if (sl >= 0) {
      z = ray_z(k, ray, sl, (int)((i / (nv > 0 ? nv : 1)) % S));
      ray_point(ray, z, p);
      const float* R = k.R_wk + (int64_t)ray.b * 9;
      const float* ti = k.t_inv + (int64_t)ray.b * 3;
#pragma unroll
      for (int a = 0; a < 3; ++a) c[a] = ((p[0] * R[a] + p[1] * R[3 + a]) + p[2] * R[6 + a]) + ti[a];
      id = k.frame_ids ? k.frame_ids[ray.b] : (int64_t)ray.b;
    }
    k.coords[out * 3] = c[0]; k.coords[out * 3 + 1] = c[1]; k.coords[out * 3 + 2] = c[2];
    k.ids[out] = id;
    k.aux[out] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (k.pc_world) { k.pc_world[out * 3] = p[0]; k.pc_world[out * 3 + 1] = p[1]; k.pc_world[out * 3 + 2] = p[2]; }
    if (k.z_vals) k.z_vals[out] = z;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { k.counts[1] = all2; k.counts[2] = all2 * S; k.counts[3] = 0; }
}

// t_inv[b] = -(R_wk[b]^T t_wk[b])   (utils_geometry.py:238)
__global__ void ray_pose_inverse_kernel(const float* R, const float* t, float* t_inv, int B) {
#pragma clang fp contract(off)
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float* Rb = R + (int64_t)b * 9;
  const float* tb = t + (int64_t)b * 3;
#pragma unroll
  for (int i = 0; i < 3; ++i) t_inv[b * 3 + i] = -((Rb[i] * tb[0] + Rb[3 + i] * tb[1]) + Rb[6 + i] * tb[2]);
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t sample_rays_workspace_bytes(int64_t n_rays, int32_t n_frames) {
  const size_t blocks = (size_t)((n_rays + RAY_BLOCK - 1) / RAY_BLOCK);
  return align256(blocks * 4) * 2 + align256((size_t)n_rays * 4) + align256((size_t)n_frames * 12);
}

hipError_t launch_sample_rays(const miso_ray_frames_t& f, const miso_ray_sampling_t& c, const float* lin,
                              int64_t n_rays, const int64_t* pix_b, const int64_t* pix_h, const int64_t* pix_w,
                              const float* u, const float* g, void* workspace, float* coords, int64_t* ids,
                              float* aux, float* pc_world, float* z_vals, int32_t* counts, hipStream_t s) {
  if (n_rays == 0) return launch_zero_words(counts, 4, s);
  const unsigned blocks = (unsigned)((n_rays + RAY_BLOCK - 1) / RAY_BLOCK);
  RayK k;
  k.depth = f.depth; k.normals = f.normals; k.T_WC = f.T_WC; k.R_wk = f.R_wk; k.frame_ids = f.frame_ids;
  k.pix_b = pix_b; k.pix_h = pix_h; k.pix_w = pix_w; k.u = u; k.g = g;
  k.n_rays = n_rays; k.B = f.n_frames; k.H = f.H; k.W = f.W; k.rays_per_frame = c.rays_per_frame;
  k.fx = f.fx; k.fy = f.fy; k.cx = f.cx; k.cy = f.cy;
  k.min_depth = c.min_depth; k.behind = c.dist_behind_surf; k.trunc = c.trunc_dist;
  k.n_strat = c.n_strat; k.n_surf = c.n_surf; k.S = c.n_strat + c.n_surf;
  for (int i = 0; i <= c.n_strat; ++i) k.lin[i] = lin[i];
  char* ws = static_cast<char*>(workspace);
  k.cnt1 = reinterpret_cast<int32_t*>(ws); ws += align256((size_t)blocks * 4);
  k.cnt2 = reinterpret_cast<int32_t*>(ws); ws += align256((size_t)blocks * 4);
  k.slot = reinterpret_cast<int32_t*>(ws); ws += align256((size_t)n_rays * 4);
  float* t_inv = reinterpret_cast<float*>(ws);
  k.t_inv = t_inv;
  k.coords = coords; k.ids = ids; k.aux = reinterpret_cast<float4*>(aux); k.pc_world = pc_world; k.z_vals = z_vals;
  k.counts = counts;
  ray_pose_inverse_kernel<<<(f.n_frames + 63) / 64, 64, 0, s>>>(f.R_wk, f.t_wk, t_inv, f.n_frames);
  ray_flag_kernel<<<blocks, RAY_BLOCK, 0, s>>>(k);
  ray_check_kernel<<<blocks, RAY_BLOCK, 0, s>>>(k);
  const unsigned ny = (unsigned)std::max(1, std::min(k.S, (int)((512 + blocks - 1) / blocks)));
  ray_emit_kernel<<<dim3(blocks, ny), RAY_BLOCK, 0, s>>>(k);
  return hipGetLastError();
}

}  // namespace miso
