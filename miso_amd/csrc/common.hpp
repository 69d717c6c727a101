// Shared device helpers for libmiso_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "../../include/miso_hip.h"

namespace miso {

// Kernel-side view of one level (no grad/data split: the launcher picks).
struct LevelK {
  const float* data;
  float* grad;
  const float* gg;  // cotangent-of-grad grid (second order), or nullptr
  unsigned char* touched;  // miso_level_t.grad_touched: one byte per MISO_ADAM_CHUNK floats of `grad`, or nullptr
  int32_t C, Z, Y, X;
  int32_t sC, sZ, sY, sX;  // element strides, validated < 2^31 on the host
  int32_t foff;            // first output column of this level
};

// internal GridK::flags bit (not part of the ABI): the batch is binned, there is no perm[] and the original index of
// sorted point p is the integer in xn[p].w (miso_sort_points with perm == NULL; sdf_train_kernel)
constexpr uint32_t MISO_F_INDEX_IN_XN = 1u << 20;

// GridK::tune bit 31: sdf_train_kernel<SCAT> has a second block of cell records in LDS and requests the next chunk's
// gathers in front of this chunk's atomics (set by its launcher when the shape leaves room)
constexpr uint32_t MISO_TUNE_ROTATE = 1u << 31;
constexpr int MISO_ROTATE_MAX_F = 12;        // ... for feature rows up to 12 floats (wider rows: the rotated loop spills)
constexpr size_t MISO_LDS_LIMIT = 160 * 1024;      // LDS of a CDNA4 compute unit = the most one workgroup can have

struct GridK {
  int32_t n_levels;
  uint32_t ignore_mask;
  uint32_t flags;
  int32_t F;
  float bmin[3], bmax[3];
  // sorted batches hand the kernels pre-normalised float4 points (sort.hip): xstride = 4,
  // flags carry COORDS_NORMALIZED and gscale = 2/len restores d xn / d x for the pose gradient
  float gscale[3];
  int32_t xstride;
  uint32_t tune;   // dev-only ablation bits from $MISO_TUNE (0 in production), and MISO_TUNE_ROTATE (set by launch_train_t)
  LevelK lv[MISO_MAX_LEVELS];
};

// One fused atlas query (atlas.hip; miso_atlas_sdf_fwd)
struct AtlasK {
  const GridK* submaps;      // device, n_submaps entries (miso_atlas_plan_build)
  const float* poses;        // device, n_submaps x 12: R_submap_world row-major (9), t_submap_world (3)
  int32_t n_submaps;
  const float* x;            // (N,3) world points, or nullptr: lattice
  const float* ax[3];        // lattice axes (device): point p = (i ny + j) nz + k sits at (ax[0][i], ax[1][j], ax[2][k])
  int32_t dim[3];
  int64_t n;
  float* sdf;                // (N) or nullptr
  float* feats;              // (N, ld) mean features or nullptr
  int64_t ld;
  int32_t no_bound;          // MISO_F_ATLAS_NO_BOUND: no coords_in_bound test (one submap queried as GridNet.forward does)
};

__device__ __forceinline__ void load_point(const GridK& g, const float* __restrict__ x, int64_t p, float& px,
                                           float& py, float& pz) {
  if (g.xstride == 4) {
    const float4 v = reinterpret_cast<const float4*>(x)[p];
    px = v.x; py = v.y; pz = v.z;
  } else {
    px = x[p * 3 + 0]; py = x[p * 3 + 1]; pz = x[p * 3 + 2];
  }
}

// miso_sorted_t.tiles_per_axis: a plain count (1..16: cubic binning) or MISO_TILES_XYZ(tx, ty, tz), 1..32 per axis
inline bool tiles_xyz(int32_t code, int T[3]) {
  if (code >= 1 && code <= 16) { T[0] = T[1] = T[2] = code; return true; }
  if (code < 256 || (code >> 24) != 0) return false;
  T[0] = code & 255; T[1] = (code >> 8) & 255; T[2] = (code >> 16) & 255;
  for (int a = 0; a < 3; ++a)
    if (T[a] < 1 || T[a] > 32) return false;
  return true;
}
inline bool tiles_cubic16(int32_t code) { return code >= 1 && code <= 16; }

// clears n_words 32-bit words with a kernel (loss.hip; see there why not hipMemsetAsync)
hipError_t launch_zero_words(void* p, int n_words, hipStream_t s);

struct MlpK {
  const float* w[MISO_MAX_LINEAR];
  const float* b[MISO_MAX_LINEAR];
};

// Per-axis sample position.  Mirrors the reference op by op (no FMA
// contraction) so coordinates are bit-identical to the PyTorch path:
//   utils.normalize_coordinates (grid_opt/utils/utils.py:49):
//       xn = 2 * (x - bmin) / (bmax - bmin) - 1
//   ATen grid_sampler_unnormalize (align_corners=False): ix = ((xn + 1) * size - 1) / 2
// `mult` is d(ix)/d(x) as autograd forms it.
struct Axis {
  float pos;   // continuous index ix
  float mult;  // d ix / d x  (0 where border padding clipped the coordinate)
};

__device__ __forceinline__ Axis axis_coord(float x, float bmin, float bmax, int size, uint32_t flags) {
  float xn = x;
  float m = 1.0f;
  if (!(flags & MISO_F_COORDS_NORMALIZED)) {
    float len = __fsub_rn(bmax, bmin);
    xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(x, bmin)), len), 1.0f);
    m = __fdiv_rn(2.0f, len);
  }
  float fs = (float)size;
  float ix;
  if (flags & MISO_F_ALIGN_CORNERS) {
    ix = __fmul_rn(__fmul_rn(__fadd_rn(xn, 1.0f), 0.5f), (float)(size - 1));   // x / 2 == x * 0.5 exactly
    m *= 0.5f * (float)(size - 1);
  } else {
    ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xn, 1.0f), fs), 1.0f), 0.5f);   // (an IEEE divide is ~10 instructions)
    m *= 0.5f * fs;
  }
  if (flags & MISO_F_PAD_BORDER) {
    float hi = (float)(size - 1);
    if (ix <= 0.0f) { ix = 0.0f; m = 0.0f; }
    else if (ix >= hi) { ix = hi; m = 0.0f; }
  }
  Axis a; a.pos = ix; a.mult = m;
  return a;
}

// The two halves of axis_coord, for a kernel that evaluates several levels at one point: the normalised coordinate and
// the factor 2 / len do not depend on the level (same operations, same order, same bits -- formed once instead of per level:
// two IEEE divisions per axis and level otherwise).
__device__ __forceinline__ void axis_norm(float x, float bmin, float bmax, uint32_t flags, float& xn, float& m) {
  xn = x; m = 1.0f;
  if (!(flags & MISO_F_COORDS_NORMALIZED)) {
    const float len = __fsub_rn(bmax, bmin);
    xn = __fsub_rn(__fdiv_rn(__fmul_rn(2.0f, __fsub_rn(x, bmin)), len), 1.0f);
    m = __fdiv_rn(2.0f, len);
  }
}
__device__ __forceinline__ Axis axis_from_norm(float xn, float m, int size, uint32_t flags) {
  const float fs = (float)size;
  float ix;
  if (flags & MISO_F_ALIGN_CORNERS) {
    ix = __fmul_rn(__fmul_rn(__fadd_rn(xn, 1.0f), 0.5f), (float)(size - 1));
    m *= 0.5f * (float)(size - 1);
  } else {
    ix = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xn, 1.0f), fs), 1.0f), 0.5f);
    m *= 0.5f * fs;
  }
  if (flags & MISO_F_PAD_BORDER) {
    const float hi = (float)(size - 1);
    if (ix <= 0.0f) { ix = 0.0f; m = 0.0f; }
    else if (ix >= hi) { ix = hi; m = 0.0f; }
  }
  Axis a; a.pos = ix; a.mult = m;
  return a;
}

// Trilinear cell: base corner, the two 1-D weights per axis (ATen forms them
// as (i0+1-ix) and (ix-i0), gridsample_cuda.cu:335-342) and in-range flags.
struct Cell {
  int i0, j0, k0;
  float wx[2], wy[2], wz[2];
  bool inx[2], iny[2], inz[2];
};

__device__ __forceinline__ void axis_cell(float pos, int size, int& i0, float w[2], bool in[2]) {
  float f = floorf(pos);
  // clamp before the int conversion so far-away points cannot overflow
  f = fminf(fmaxf(f, -2.0f), (float)size + 1.0f);
  i0 = (int)f;
  if (pos != pos) { i0 = -2; }  // NaN coordinate: every corner out of range
  w[0] = __fsub_rn(f + 1.0f, pos);
  w[1] = __fsub_rn(pos, f);
  in[0] = (i0 >= 0) && (i0 < size);
  in[1] = (i0 + 1 >= 0) && (i0 + 1 < size);
}

__device__ __forceinline__ Cell make_cell(const Axis& ax, const Axis& ay, const Axis& az,
                                          const LevelK& lv) {
  Cell c;
  axis_cell(ax.pos, lv.X, c.i0, c.wx, c.inx);
  axis_cell(ay.pos, lv.Y, c.j0, c.wy, c.iny);
  axis_cell(az.pos, lv.Z, c.k0, c.wz, c.inz);
  return c;
}

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
  // hardware fp32 add at L2 (global_atomic_add_f32, no return value)
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One Levenberg-Marquardt step of the tracker on the device (lm.hip)
struct LmTrackK {
  const float* x;            // (N,3) samples in the keyframe frame
  const float* gt;           // (N) measured SDF, element stride s_gt
  const void* valid;         // (N) validity (float or bool), stride s_valid, or nullptr
  const int64_t* frame_ids;  // (N), stride s_fid, or nullptr
  int64_t s_gt, s_valid, s_fid;
  int valid_is_bool;
  int64_t n, kf;
  float trunc;
  const float* Rwk;          // 9: base rotation of the keyframe
  const float* twk;          // 3
  float* dr;                 // 3: rotation correction (axis-angle), updated by the solve
  float* dt;                 // 3
  float* pose;               // scratch 12: R (row-major), t
  float* xw;                 // scratch (N,3): samples in the submap frame
  float* sums;               // scratch 36: 32 of lm_normal_eq + {rows kept, in bound, wrong frame id, invalid}
  float* info;               // out 8: |dR| (rad), |dt|, |g|, in bound, kept, wrong frame id, invalid, 0
  float* clean;              // nullptr, or scratch 5N: torch.nan_to_num of x (3N), gt (N), valid (N, as floats) -- written
                             // by the transform kernel, read by everything after it (the caller repoints x / gt / valid)
  float bmin[3], bmax[3];
  float lm_lambda;
};

// miso_level_t.grad_touched: a non-zero went into grad[off] (element offset from lv.grad)
constexpr int ADAM_CHUNK_SHIFT = 6;
static_assert(MISO_ADAM_CHUNK == (1 << ADAM_CHUNK_SHIFT), "touch_chunk shifts by log2 of the chunk");
__device__ __forceinline__ void touch_chunk(const LevelK& lv, int64_t off) {
  if (lv.touched) lv.touched[off >> ADAM_CHUNK_SHIFT] = 1;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device, size): the call is ~5 us of host time, and a
// training step that issues its six launches on the stream makes it three times (the host's lead over the device is
// what protects such a step against a busy host).  The attribute only ever grows.
inline hipError_t allow_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 48 * 1024) return hipSuccess;
  static std::mutex mu;
  static std::unordered_map<uint64_t, size_t> granted;      // (kernel address ^ device) -> bytes
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t key = (uint64_t)(uintptr_t)kernel ^ ((uint64_t)(dev + 1) << 56);
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return hipSuccess;
  }
  e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  size_t& g = granted[key];
  if (g < bytes) g = bytes;
  return hipSuccess;
}

// One (src, dst) pair of a fused alignment iteration, as the batched kernels of pair_latent.hip read it from the
// device-resident plan (built on the host by miso_align_plan_build from miso_align_pair_t).
struct AlignPairK {
  GridK g;              // destination levels 0..level (data only) and bound
  const float* p;       // (n,3) source vertices in the source frame
  const float* fsrc;    // (n, ld) source features
  int64_t ld, n;
  const float* gate_p;  // (gate_n,3) finest-level source vertices for the overlap gate, or nullptr: always on
  int64_t gate_n;
  // the same vertices as a lattice (preferred: nothing to read but three short tables): vertex (i,j,k), index
  // (k ny + j) nx + i, sits at (gate_ax[0][i], gate_ax[1][j], gate_ax[2][k]); gate_n = nx ny nz
  const float* gate_ax[3];
  int32_t gate_dim[3];
  int32_t src, dst;
  float n_ch;           // channels compared (the L2 mean divides by count * n_ch)
  // axis-aligned boxes {min xyz, max xyz} (source frame) of every run of ALIGN_BOX_VERTS consecutive source vertices, or
  // nullptr: a run that cannot reach the destination bound under the current poses is not read
  const float* boxes;
};
constexpr int ALIGN_BOX_VERTS = 64;

// One torch.optim.Adam update (amsgrad=False, weight_decay=0) as torch forms it op by op; shared by adam.hip
// (dense grids) and align.hip (the 6(S-1) pose numbers of the alignment loop).
struct AdamScalars {
  float one_minus_b1, b2, one_minus_b2, neg_step_size, bc2_sqrt, eps;
};

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, const AdamScalars& a) {
#pragma clang fp contract(off)   // one rounding per torch op, and the same bits from every kernel that inlines this
  m = m + a.one_minus_b1 * (g - m);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * a.b2 + (a.one_minus_b2 * g) * g;          // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
  float denom = sqrtf(v) / a.bc2_sqrt + a.eps;      // (sqrt / bias_correction2_sqrt).add_(eps)
  p = p + (a.neg_step_size * m) / denom;            // addcdiv_(exp_avg, denom, value=-step_size)
}

// One Adam iteration of the tracker's pose refinement on the device (lm.hip: miso_track_adam_step)
struct TrackAdamK {
  LmTrackK s;
  int loss_type;             // 1 L1, 2 L2, 3 GM
  float weight_sdf, gm_scale;
  float* gpred;              // scratch (N): d loss / d sdf
  const float* gx;           // (N,3): d loss / d x_world, from the fused coordinate backward
  const float* sdf;          // (N)
  const AdamScalars* table;  // step scalars of steps 1..table_len (miso_adam_scalars_table)
  int table_len;
  float* state;              // 12 floats m[6] v[6], then int32 {steps taken, skipped, iterations}
  float* ring;               // loss of iteration i at ring[i], i < ring_len
  int ring_len;
};



// Pointwise mapping loss (grid_opt/loss.py:594-635, :668-700) shared by loss.hip and the fused
// backward.  g / gf: d(w_sdf * sdf term) / ds and d(w_fs * free-space term) / ds BEFORE the 1/N of
// the mean; s_sdf / s_fs accumulate the unweighted term sums.
struct MapLossK {
  int loss_type;  // 1 = L1, 2 = L2
  float w_sdf, w_fs, trunc;
};

__device__ __forceinline__ void map_loss_one(const MapLossK& p, float s, float t, float w, bool v, bool fs,
                                             float& g, float& gf, float& s_sdf, float& s_fs) {
  g = 0.f; gf = 0.f;
  const float d = s - t;
  if (v) {
    if (p.loss_type == 1) {
      s_sdf += w * fabsf(d);
      g = p.w_sdf * w * ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f));
    } else {
      s_sdf += w * d * d;
      g = p.w_sdf * w * 2.f * d;
    }
  }
  if (fs) {
    const float up = fmaxf(d, 0.f), lo = fmaxf(p.trunc - s, 0.f);
    s_fs += fmaxf(up, lo);
    // d/ds max(relu(s-t), relu(trunc-s)); ties carry zero slope on both sides
    if (up > lo) gf = p.w_fs;
    else if (lo > up) gf = -p.w_fs;
  }
}


// The mapping loss folded into the fused forward (sdf_fwd_kernel): with loss_type != 0 the kernel
// forms d loss / d sdf for every point right after its SDF and writes it in the binned order.
struct LossInK {
  MapLossK p;          // p.loss_type 0 = not fused
  const float4* aux;   // (N) {target, valid, sign, weight} per point, caller order
  float* gsdf_sorted;  // (N) out: d loss / d sdf, binned order
  float* loss_out;     // (MISO_LOSS_SLOTS, 2): every slot written by the launch
  float inv_n;
  const int32_t* n_live;   // device count of live rows (the rest is neutral padding), or nullptr: the means divide by it
};

}  // namespace miso
