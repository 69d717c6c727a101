// extern "C" entry points of libmiso_hip.so (see include/miso_hip.h).
// Argument validation + conversion to kernel-side structs + dispatch.
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "align.hpp"
#include "version.inc"

namespace miso {
hipError_t launch_encode_fwd(const GridK&, bool, const float*, int64_t, float*, int64_t, const int*, hipStream_t);
hipError_t launch_encode_bwd(const GridK&, bool, const float*, int64_t, const float*, int64_t, float*, const int*,
                             hipStream_t);
hipError_t launch_encode_bwd2(const GridK&, bool, const float*, int64_t, const float*, int64_t, const float*, float*,
                              int64_t, float*, const int*, hipStream_t);
bool fused_shape_supported(int C, int L, int H, int NH);
int64_t sdf_train_lds_bytes(int C, int L, int H, int NH, bool scat);
hipError_t launch_sdf_fwd(int, int, int, int, const GridK&, const float*, const float*, int64_t, float*,
                          uint32_t*, const int*, const LossInK&, hipStream_t);
hipError_t launch_sdf_bwd(int, int, int, int, const GridK&, const float*, const float*, int64_t,
                          const float*, const uint32_t*, float*, bool, const int*, float*, uint32_t, bool,
                          hipStream_t);
hipError_t launch_sdf_train(int, int, int, int, const GridK&, const float*, const float*, int64_t, float*, const int*,
                            const LossInK&, float*, uint32_t, bool, hipStream_t);
int64_t sort_workspace_bytes(int64_t n, int T);
hipError_t launch_sort(const GridK&, const float*, int64_t, int, void*, float*, float*, int*, int*,
                       hipStream_t);
hipError_t launch_pair_latent(const GridK&, bool, const float*, const float*, const float*, int64_t, int64_t, int,
                              double*, hipStream_t);
hipError_t launch_zero_fill(float*, int64_t, hipStream_t);
hipError_t launch_adam_touched(float*, float*, float*, float*, unsigned char*, unsigned char*, int64_t, double, double,
                               double, double, int, int, const float*, hipStream_t, const float*, const int32_t*, int);
hipError_t launch_adam_bump(int32_t*, const float*, hipStream_t);
hipError_t launch_adam_active_multi(const miso_adam_tensor_t*, int, double, double, double, double, int, const float*, int,
                                    const int32_t*, const float*, hipStream_t);
hipError_t launch_loss_total_bump(const float*, int, float*, int32_t*, float*, int, hipStream_t);
hipError_t launch_lm_track_head(const LmTrackK&, hipStream_t);
hipError_t launch_track_loss(const TrackAdamK&, hipStream_t);
hipError_t launch_track_tail(const TrackAdamK&, hipStream_t);
hipError_t launch_lm_track_tail(const LmTrackK&, const float*, const float*, int, float, hipStream_t);
void adam_scalars_table(double, double, double, double, int, int, float*);
hipError_t launch_mapping_batch(const float*, const float*, int32_t, const int64_t*, int64_t, const int64_t*,
                                const float*, const float*, const void*, const float*, const float*, int64_t, float*,
                                float*, const int64_t*, int, int, hipStream_t);
hipError_t launch_mapping_loss_rows(int, float, float, float, const float*, const float*, int64_t, float*, float*,
                                    hipStream_t);
uint32_t plan_grad_pull(const GridK&, int);
hipError_t launch_grad_pull(const GridK&, int, int, const int*, const float*, const float*, int64_t, const int*,
                            uint32_t, int, const float*, int32_t*, int64_t, hipStream_t, uint32_t push_mask, int64_t n);
uint32_t plan_push(const GridK&, int, int64_t, uint32_t);
bool mc_pull_ok(const GridK& g, int C, const int T[3], uint32_t level_mask, int64_t n, int64_t ld);
int64_t pull_queue_ints(int64_t);
hipError_t launch_rigid_by_index(const float*, const float*, const int64_t*, const float*, int64_t, int32_t, int, float*,
                                 hipStream_t);
int64_t mc_words(int32_t, int32_t, int32_t);
int64_t mc_workspace_bytes(int32_t, int32_t, int32_t);
hipError_t launch_mc_classify(const float*, int32_t, int32_t, int32_t, float, void*, int32_t*, hipStream_t);
hipError_t launch_mc_emit(int32_t, int32_t, int32_t, void*, const int64_t*, int32_t, int64_t, int64_t*, hipStream_t);
hipError_t launch_mc_vertices(const float*, int32_t, int32_t, int32_t, float, void*, const int64_t*, int32_t, int64_t,
                              float*, hipStream_t);
void mc_copy_table(int8_t*);
hipError_t launch_overlap_count(const float*, const float*, int64_t, const float*, const float*, float*, hipStream_t);
hipError_t launch_src_boxes(const float*, int64_t, float*, hipStream_t);
hipError_t launch_lm_normal_eq(const float*, const float*, const float*, const float*, const float*, int64_t, int,
                               float, float*, hipStream_t);
size_t sample_rays_workspace_bytes(int64_t, int32_t);
hipError_t launch_sample_rays(const miso_ray_frames_t&, const miso_ray_sampling_t&, const float*, int64_t,
                              const int64_t*, const int64_t*, const int64_t*, const float*, const float*, void*,
                              float*, int64_t*, float*, float*, float*, int32_t*, hipStream_t);
hipError_t launch_grid_pool_avg(const float*, const float*, int64_t, int32_t, int64_t, const float*, float, int32_t, int32_t,
                                int32_t, float*, int32_t*, hipStream_t);
hipError_t launch_atlas_sdf(int C, int L, int H, int NH, const AtlasK& a, const float* packed, bool exact, hipStream_t s);
hipError_t launch_mlp_pack(const MlpK&, int, int, int, float*, hipStream_t);
int64_t mlp_packed_floats(int F, int H, int NH);
hipError_t launch_adam(float*, float*, float*, float*, int64_t, double, double, double, double, int, int,
                       hipStream_t);
hipError_t launch_adam_active(float*, float*, float*, float*, unsigned char*, int64_t, double, double, double, double, int,
                              int, const float*, hipStream_t, const float*, const int32_t*, int);
hipError_t launch_mapping_loss(int, float, float, float, const float*, const float*, const float*,
                               const float*, const float*, int64_t, float*, float*, float*, hipStream_t);
hipError_t launch_align_a(const AlignK&, int64_t, int64_t, int64_t, bool, bool, hipStream_t);
hipError_t launch_align_b(const AlignK&, hipStream_t);
}  // namespace miso

using namespace miso;


namespace {

// which pointer of a level a call needs
enum Need { NEED_DATA = 1, NEED_GRAD_OPT = 2 };

int convert_grid(const miso_grid_t* in, GridK* out, bool need_data, bool* vec4) {
  if (!in || in->n_levels < 1 || in->n_levels > MISO_MAX_LEVELS) return MISO_E_BADARG;
  if (in->flags & ~(MISO_F_ALIGN_CORNERS | MISO_F_PAD_BORDER | MISO_F_COORDS_NORMALIZED | MISO_F_GRAD_OVERWRITE |
                    MISO_F_GRAD_SDF_SORTED | MISO_F_GRAD_ZEROED | MISO_F_CROWDED | MISO_F_EXACT_F32 | MISO_F_FULL_TRIPS))
    return MISO_E_BADARG;
  memset(out, 0, sizeof(*out));
  out->n_levels = in->n_levels;
  out->ignore_mask = in->ignore_mask;
  out->flags = in->flags & ~(MISO_F_GRAD_OVERWRITE | MISO_F_GRAD_SDF_SORTED | MISO_F_GRAD_ZEROED);   // host-side flags
  for (int a = 0; a < 3; ++a) { out->bmin[a] = in->bound_min[a]; out->bmax[a] = in->bound_max[a]; out->gscale[a] = 1.0f; }
  out->xstride = 3;
  static const uint32_t tune = [] { const char* e = getenv("MISO_TUNE"); return e ? (uint32_t)atoi(e) : 0u; }();
  out->tune = tune;
  bool v4 = true;
  int foff = 0;
  for (int l = 0; l < in->n_levels; ++l) {
    const miso_level_t& s = in->level[l];
    if (s.C < 1 || s.X < 1 || s.Y < 1 || s.Z < 1) return MISO_E_BADARG;
    if (need_data && !s.data) return MISO_E_BADARG;
    if (s.sC < 0 || s.sX < 0 || s.sY < 0 || s.sZ < 0) return MISO_E_BADARG;
    int64_t span = (int64_t)(s.C - 1) * s.sC + (int64_t)(s.X - 1) * s.sX + (int64_t)(s.Y - 1) * s.sY +
                   (int64_t)(s.Z - 1) * s.sZ + 1;
    if (span >= ((int64_t)1 << 31)) return MISO_E_TOOLARGE;
    LevelK& d = out->lv[l];
    d.data = s.data; d.grad = s.grad; d.gg = nullptr;
    d.touched = s.grad ? s.grad_touched : nullptr;
    d.C = s.C; d.X = s.X; d.Y = s.Y; d.Z = s.Z;
    d.sC = (int32_t)s.sC; d.sX = (int32_t)s.sX; d.sY = (int32_t)s.sY; d.sZ = (int32_t)s.sZ;
    d.foff = foff;
    foff += s.C;
    bool ok = (s.C % 4 == 0) && (s.sC == 1 || s.C == 1) && (s.sX % 4 == 0) && (s.sY % 4 == 0) &&
              (s.sZ % 4 == 0) && (((uintptr_t)s.data & 15u) == 0) && (((uintptr_t)s.grad & 15u) == 0);
    v4 = v4 && ok;
  }
  out->F = foff;
  if (vec4) *vec4 = v4;
  return MISO_OK;
}

int fused_shape(const GridK& g, bool vec4, const miso_mlp_t* m, int* C, int* L, int* H, int* NH) {
  if (!m || !vec4) return MISO_E_UNSUPPORTED;
  if (m->n_linear < 2 || m->n_linear > MISO_MAX_LINEAR) return MISO_E_UNSUPPORTED;
  int c = g.lv[0].C;
  for (int l = 0; l < g.n_levels; ++l)
    if (g.lv[l].C != c) return MISO_E_UNSUPPORTED;
  if (m->in_dim != g.F || m->out_dim != 1) return MISO_E_UNSUPPORTED;
  int nh = m->n_linear - 2;
  if (!fused_shape_supported(c, g.n_levels, m->hidden_dim, nh)) return MISO_E_UNSUPPORTED;
  *C = c; *L = g.n_levels; *H = m->hidden_dim; *NH = nh;
  return MISO_OK;
}

}  // namespace

extern "C" {

const char* miso_version(void) { return "miso_hip 0.2 (gfx950) src=" MISO_SOURCE_HASH; }

const char* miso_error_string(int code) {
  switch (code) {
    case MISO_OK: return "ok";
    case MISO_E_BADARG: return "bad argument";
    case MISO_E_UNSUPPORTED: return "shape not covered by the fused kernels";
    case MISO_E_TOOLARGE: return "a grid level spans >= 2^31 elements";
    default: return hipGetErrorString((hipError_t)code);
  }
}

int miso_encode_fwd(const miso_grid_t* grid, const float* x, int64_t n, float* feats, int64_t ld_out,
                    void* stream) {
  if (n < 0 || (n > 0 && (!x || !feats))) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, true, &v4);
  if (rc) return rc;
  if (ld_out < g.F) return MISO_E_BADARG;
  return (int)launch_encode_fwd(g, v4, x, n, feats, ld_out, nullptr, (hipStream_t)stream);
}

static const float* sorted_points(GridK* g, const miso_sorted_t* sorted);
static int check_sorted(const miso_sorted_t* s, int64_t n, bool perm_optional = false);

static int encode_bwd_impl(const miso_grid_t* grid, const float* x, int64_t n, const float* grad_feats,
                           int64_t ld_g, float* grad_x, const miso_sorted_t* sorted, void* stream) {
  if (n < 0 || (n > 0 && !grad_feats)) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, grad_x != nullptr, &v4);
  if (rc) return rc;
  if (ld_g < g.F) return MISO_E_BADARG;
  const int* perm = nullptr;
  if (sorted) { x = sorted_points(&g, sorted); perm = sorted->perm; }
  if (n > 0 && !x) return MISO_E_BADARG;
  return (int)launch_encode_bwd(g, v4, x, n, grad_feats, ld_g, grad_x, perm, (hipStream_t)stream);
}

int miso_encode_bwd(const miso_grid_t* grid, const float* x, int64_t n, const float* grad_feats,
                    int64_t ld_g, float* grad_x, void* stream) {
  return encode_bwd_impl(grid, x, n, grad_feats, ld_g, grad_x, nullptr, stream);
}

int miso_encode_bwd_sorted(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n,
                           const float* grad_feats, int64_t ld_g, float* grad_x, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  return encode_bwd_impl(grid, nullptr, n, grad_feats, ld_g, grad_x, sorted, stream);
}

static int encode_bwd2_impl(const miso_grid_t* grid, const miso_grid_t* gg_grid, const float* x, int64_t n,
                            const float* grad_feats, int64_t ld_g, const float* gg_x, float* gg_out,
                            int64_t ld_gg, float* g_x, const miso_sorted_t* sorted, void* stream) {
  if (n < 0 || (n > 0 && (!grad_feats || !gg_out))) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, true, &v4);
  if (rc) return rc;
  if (ld_g < g.F || ld_gg < g.F) return MISO_E_BADARG;
  if (gg_grid) {
    if (gg_grid->n_levels != grid->n_levels) return MISO_E_BADARG;
    for (int l = 0; l < g.n_levels; ++l) {
      const miso_level_t& a = grid->level[l];
      const miso_level_t& b = gg_grid->level[l];
      if (!b.data) continue;
      if (a.C != b.C || a.X != b.X || a.Y != b.Y || a.Z != b.Z || a.sC != b.sC || a.sX != b.sX ||
          a.sY != b.sY || a.sZ != b.sZ)
        return MISO_E_BADARG;  // cotangent must share the layout of the grid
      if (((uintptr_t)b.data & 15u) != 0) v4 = false;
      g.lv[l].gg = b.data;
    }
  }
  const int* perm = nullptr;
  if (sorted) { x = sorted_points(&g, sorted); perm = sorted->perm; }
  if (n > 0 && !x) return MISO_E_BADARG;
  return (int)launch_encode_bwd2(g, v4, x, n, grad_feats, ld_g, gg_x, gg_out, ld_gg, g_x, perm,
                                 (hipStream_t)stream);
}

int miso_encode_bwd2(const miso_grid_t* grid, const miso_grid_t* gg_grid, const float* x, int64_t n,
                     const float* grad_feats, int64_t ld_g, const float* gg_x, float* gg_out,
                     int64_t ld_gg, float* g_x, void* stream) {
  return encode_bwd2_impl(grid, gg_grid, x, n, grad_feats, ld_g, gg_x, gg_out, ld_gg, g_x, nullptr, stream);
}

int miso_encode_bwd2_sorted(const miso_grid_t* grid, const miso_grid_t* gg_grid, const miso_sorted_t* sorted,
                            int64_t n, const float* grad_feats, int64_t ld_g, const float* gg_x, float* gg_out,
                            int64_t ld_gg, float* g_x, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  return encode_bwd2_impl(grid, gg_grid, nullptr, n, grad_feats, ld_g, gg_x, gg_out, ld_gg, g_x, sorted, stream);
}

int64_t miso_mlp_packed_floats(const miso_mlp_t* mlp) {
  if (!mlp || mlp->n_linear < 2 || mlp->n_linear > MISO_MAX_LINEAR) return 0;
  if (mlp->out_dim != 1 || mlp->in_dim < 1 || mlp->in_dim > 32) return 0;
  if (mlp->hidden_dim != 32 && mlp->hidden_dim != 64) return 0;
  return mlp_packed_floats(mlp->in_dim, mlp->hidden_dim, mlp->n_linear - 2);
}

int64_t miso_sdf_mask_words(const miso_mlp_t* mlp) {
  if (!mlp) return 0;
  return (int64_t)(mlp->n_linear - 1) * (mlp->hidden_dim / 32);
}

int miso_mlp_pack(const miso_mlp_t* mlp, float* packed, void* stream) {
  if (!packed || miso_mlp_packed_floats(mlp) == 0) return MISO_E_UNSUPPORTED;
  MlpK k;
  memset(&k, 0, sizeof(k));
  for (int i = 0; i < mlp->n_linear; ++i) {
    if (!mlp->weight[i]) return MISO_E_BADARG;
    k.w[i] = mlp->weight[i];
    k.b[i] = mlp->bias[i];
  }
  return (int)launch_mlp_pack(k, mlp->in_dim, mlp->hidden_dim, mlp->n_linear - 2, packed,
                              (hipStream_t)stream);
}

int64_t miso_sdf_train_lds_bytes(const miso_grid_t* grid, const miso_mlp_t* mlp, int32_t scattering) {
  GridK g; bool v4;
  if (convert_grid(grid, &g, false, &v4)) return 0;
  int C, L, H, NH;
  if (fused_shape(g, v4, mlp, &C, &L, &H, &NH) != MISO_OK) return 0;
  return sdf_train_lds_bytes(C, L, H, NH, scattering != 0);
}

int miso_sdf_supported(const miso_grid_t* grid, const miso_mlp_t* mlp) {
  GridK g; bool v4;
  if (convert_grid(grid, &g, false, &v4)) return 0;
  int C, L, H, NH;
  return fused_shape(g, v4, mlp, &C, &L, &H, &NH) == MISO_OK ? 1 : 0;
}

// Sorted batches: read the pre-normalised float4 points instead of the metric ones.
static const float* sorted_points(GridK* g, const miso_sorted_t* sorted) {
  if (!sorted->xn_sorted) return sorted->x_sorted;
  if (!(g->flags & MISO_F_COORDS_NORMALIZED)) {
    for (int a = 0; a < 3; ++a) g->gscale[a] = 2.0f / (g->bmax[a] - g->bmin[a]);   // axis_coord's m
    g->flags |= MISO_F_COORDS_NORMALIZED;
  }
  g->xstride = 4;
  return sorted->xn_sorted;
}

static int sdf_fwd_impl(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const float* x, int64_t n, float* sdf, uint32_t* relu_mask,
                        const miso_sorted_t* sorted, void* stream, const LossInK* loss = nullptr) {
  LossInK lin;
  memset(&lin, 0, sizeof(lin));
  if (loss) lin = *loss;
  if (n < 0 || !packed || (n > 0 && !sdf && !loss)) return MISO_E_BADARG;
  if (((uintptr_t)packed & 15u) != 0) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, true, &v4);
  if (rc) return rc;
  int C, L, H, NH;
  rc = fused_shape(g, v4, mlp, &C, &L, &H, &NH);
  if (rc) return rc;
  const int* perm = nullptr;
  if (sorted) { x = sorted_points(&g, sorted); perm = sorted->perm; }
  if (n > 0 && !x) return MISO_E_BADARG;
  return (int)launch_sdf_fwd(C, L, H, NH, g, packed, x, n, sdf, relu_mask, perm, lin, (hipStream_t)stream);
}

// A per-axis (or finer than 16) binning is served by the matrix-core pull alone, and that kernel addresses the d-feat
// rows with 32-bit byte offsets: a batch whose rows pass 2 GB cannot be pulled under such a binning.  Said here, BEFORE
// the step's first kernel is launched (the launch used to fail after the backward had already deferred the
// levels to the pull).
static bool pull_serviceable(const miso_sorted_t* sorted, int64_t n, int64_t ld) {
  if (!sorted || tiles_cubic16(sorted->tiles_per_axis)) return true;
  return n * ld * 4 < ((int64_t)1 << 31);
}

static int sdf_bwd_impl(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const float* x, int64_t n, const float* grad_sdf, const uint32_t* relu_mask,
                        float* grad_x, const miso_sorted_t* sorted, float* workspace, void* stream) {
  if (n < 0 || !packed || (n > 0 && (!grad_sdf || !relu_mask))) return MISO_E_BADARG;
  if (((uintptr_t)packed & 15u) != 0) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, grad_x != nullptr, &v4);
  if (rc) return rc;
  GridK gp = g;                                   // sample positions as the kernels read them
  if (sorted) x = sorted_points(&gp, sorted);
  if (n > 0 && !x) return MISO_E_BADARG;
  int C, L, H, NH;
  rc = fused_shape(g, v4, mlp, &C, &L, &H, &NH);
  if (rc) return rc;
  bool want_grid = false;
  for (int l = 0; l < g.n_levels; ++l) want_grid = want_grid || (g.lv[l].grad != nullptr);
  if (!want_grid && !grad_x) return MISO_OK;
  const bool overwrite = (grid->flags & MISO_F_GRAD_OVERWRITE) != 0;
  const int* perm = sorted ? sorted->perm : nullptr;
  uint32_t pull = 0;
  if (sorted && sorted->xn_sorted && workspace && want_grid && ((uintptr_t)workspace & 15u) == 0)
    pull = plan_grad_pull(g, sorted->tiles_per_axis);
  // coarse levels under a crowd: the matrix-core push (grad_pull.hip; its d-feat rows go through the workspace like a
  // pulled level's, so it stays in `pull`)
  const uint32_t push = (pull && sorted) ? plan_push(g, sorted->tiles_per_axis, n, pull) : 0u;
  if (pull && !pull_serviceable(sorted, n, g.F)) return MISO_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (overwrite && !(grid->flags & MISO_F_GRAD_ZEROED)) {
    // levels that are scattered or pushed with atomics start from zero; pulled levels need no fill
    for (int l = 0; l < g.n_levels; ++l) {
      const LevelK& lv = g.lv[l];
      if (!lv.grad || (((pull & ~push) >> l) & 1u)) continue;
      size_t span = (size_t)(lv.C - 1) * lv.sC + (size_t)(lv.X - 1) * lv.sX + (size_t)(lv.Y - 1) * lv.sY +
                    (size_t)(lv.Z - 1) * lv.sZ + 1;
      // (a kernel, not hipMemsetAsync: memset nodes of a captured graph that is replayed back to back with other
      // launches in between now and then fill with garbage on ROCm 7.2 -- see loss.hip:zero_words_kernel)
      hipError_t e = launch_zero_fill(lv.grad, (int64_t)span, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (n == 0 && !pull) return MISO_OK;
  if (n > 0) {
    rc = (int)launch_sdf_bwd(C, L, H, NH, gp, packed, x, n, grad_sdf, relu_mask, grad_x, want_grid, perm,
                             pull ? workspace : nullptr, pull,
                             sorted && (grid->flags & MISO_F_GRAD_SDF_SORTED), st);
    if (rc) return rc;
  }
  if (!pull) return MISO_OK;
  return (int)launch_grad_pull(g, C, sorted->tiles_per_axis, sorted->tile_offsets, sorted->xn_sorted, workspace,
                               g.F, nullptr, pull, overwrite ? 1 : 0, nullptr, sorted->pull_queue,
                               sorted->pull_queue_ints, st, push, n);
}

int miso_sdf_fwd(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x,
                 int64_t n, float* sdf, uint32_t* relu_mask, void* stream) {
  return sdf_fwd_impl(grid, mlp, packed, x, n, sdf, relu_mask, nullptr, stream);
}

int miso_sdf_bwd(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x,
                 int64_t n, const float* grad_sdf, const uint32_t* relu_mask, float* grad_x,
                 void* stream) {
  return sdf_bwd_impl(grid, mlp, packed, x, n, grad_sdf, relu_mask, grad_x, nullptr, nullptr, stream);
}

// The first backward with its d-feat rows handed out: what the SECOND backward (create_graph=True: eikonal / smoothness
// terms, loss_isdf.py:96-152,367-377) differentiates -- d sdf / d x = J_E(x; G)^T rows, and the decoder, piecewise linear,
// contributes nothing else (miso_encode_bwd2 on these rows is the whole double backward).
int miso_sdf_bwd_rows(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                      const float* grad_sdf, const uint32_t* relu_mask, float* grad_x, float* dfeat_rows, void* stream) {
  if (!dfeat_rows) return sdf_bwd_impl(grid, mlp, packed, x, n, grad_sdf, relu_mask, grad_x, nullptr, nullptr, stream);
  if (n < 0 || !packed || (n > 0 && (!grad_sdf || !relu_mask || !x))) return MISO_E_BADARG;
  if ((((uintptr_t)packed) & 15u) != 0 || (((uintptr_t)dfeat_rows) & 15u) != 0) return MISO_E_BADARG;
  if (grid->flags & (MISO_F_GRAD_OVERWRITE | MISO_F_GRAD_SDF_SORTED | MISO_F_GRAD_ZEROED)) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, grad_x != nullptr, &v4);
  if (rc) return rc;
  int C, L, H, NH;
  rc = fused_shape(g, v4, mlp, &C, &L, &H, &NH);
  if (rc) return rc;
  if (n == 0) return MISO_OK;
  // want_grid = true selects the kernel form that stages the d-feat tile; levels without a gradient buffer are not scattered
  return (int)launch_sdf_bwd(C, L, H, NH, g, packed, x, n, grad_sdf, relu_mask, grad_x, true, nullptr, dfeat_rows, 0u, false,
                             (hipStream_t)stream);
}

// perm_optional: the entry point can take the original index from xn_sorted[p].w (miso_sdf_train_sorted)
static int check_sorted(const miso_sorted_t* s, int64_t n, bool perm_optional) {
  if (!s || !s->tile_offsets) return MISO_E_BADARG;
  if (n > 0 && (!s->x_sorted && !s->xn_sorted)) return MISO_E_BADARG;   // an empty batch has no buffers
  if (n > 0 && !s->perm && !(perm_optional && s->xn_sorted)) return MISO_E_BADARG;
  if (((uintptr_t)s->xn_sorted & 15u) != 0) return MISO_E_BADARG;
  { int t3_[3]; if (!tiles_xyz(s->tiles_per_axis, t3_)) return MISO_E_BADARG; }
  return MISO_OK;
}

int64_t miso_sort_workspace_bytes(int64_t n, int32_t tiles_per_axis) {
  if (n < 0) return 0;
  return sort_workspace_bytes(n, tiles_per_axis);
}

int miso_sort_points(const miso_grid_t* grid, const float* x, int64_t n, int32_t tiles_per_axis,
                     void* workspace, float* x_sorted, float* xn_sorted, int32_t* perm,
                     int32_t* tile_offsets, void* stream) {
  { int t3_[3]; if (n < 0 || n >= ((int64_t)1 << 31) || !tiles_xyz(tiles_per_axis, t3_)) return MISO_E_BADARG; }
  if (!workspace || !tile_offsets || (n > 0 && (!x || (!x_sorted && !xn_sorted)))) return MISO_E_BADARG;
  if (n > 0 && !perm && !xn_sorted) return MISO_E_BADARG;      // without perm the index lives in xn_sorted[p].w only
  if (((uintptr_t)xn_sorted & 15u) != 0) return MISO_E_BADARG;
  GridK g;
  int rc = convert_grid(grid, &g, false, nullptr);
  if (rc) return rc;
  return (int)launch_sort(g, x, n, tiles_per_axis, workspace, x_sorted, xn_sorted, perm, tile_offsets,
                          (hipStream_t)stream);
}

int miso_sdf_fwd_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const miso_sorted_t* sorted, int64_t n, float* sdf, uint32_t* relu_mask,
                        void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  return sdf_fwd_impl(grid, mlp, packed, nullptr, n, sdf, relu_mask, sorted, stream);
}

int miso_encode_fwd_sorted(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, float* feats,
                           int64_t ld_out, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  if (n < 0 || (n > 0 && !feats)) return MISO_E_BADARG;
  GridK g; bool v4;
  rc = convert_grid(grid, &g, true, &v4);
  if (rc) return rc;
  if (ld_out < g.F) return MISO_E_BADARG;
  const float* x = sorted_points(&g, sorted);
  return (int)launch_encode_fwd(g, v4, x, n, feats, ld_out, sorted->perm, (hipStream_t)stream);
}

int miso_sdf_fwd_sorted_loss(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                             const miso_sorted_t* sorted, int64_t n, int loss_type, float weight_sdf,
                             float weight_fs, float trunc_dist, const float* loss_inputs, float* sdf,
                             uint32_t* relu_mask, float* grad_sdf_sorted, float* loss_slots,
                             const int32_t* n_live, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  if ((loss_type != 1 && loss_type != 2) || !loss_slots || (n > 0 && (!loss_inputs || !grad_sdf_sorted)))
    return MISO_E_BADARG;
  if (((uintptr_t)loss_inputs & 15u) != 0) return MISO_E_BADARG;
  if (n == 0) return (int)launch_zero_words(loss_slots, MISO_LOSS_SLOTS * 2, (hipStream_t)stream);
  LossInK lin;
  memset(&lin, 0, sizeof(lin));
  lin.p.loss_type = loss_type; lin.p.w_sdf = weight_sdf; lin.p.w_fs = weight_fs; lin.p.trunc = trunc_dist;
  lin.aux = reinterpret_cast<const float4*>(loss_inputs);
  lin.gsdf_sorted = grad_sdf_sorted; lin.loss_out = loss_slots;
  lin.inv_n = n > 0 ? 1.0f / (float)n : 0.0f;
  lin.n_live = n_live;
  return sdf_fwd_impl(grid, mlp, packed, nullptr, n, sdf, relu_mask, sorted, stream, &lin);
}

int miso_sdf_fwd_loss(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                      int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* loss_inputs,
                      float* sdf, uint32_t* relu_mask, float* grad_sdf, float* loss_slots, void* stream) {
  if ((loss_type != 1 && loss_type != 2) || !loss_slots || n < 0 || (n > 0 && (!x || !loss_inputs || !grad_sdf)))
    return MISO_E_BADARG;
  if (((uintptr_t)loss_inputs & 15u) != 0) return MISO_E_BADARG;
  if (n == 0) return (int)launch_zero_words(loss_slots, MISO_LOSS_SLOTS * 2, (hipStream_t)stream);
  LossInK lin;
  memset(&lin, 0, sizeof(lin));
  lin.p.loss_type = loss_type; lin.p.w_sdf = weight_sdf; lin.p.w_fs = weight_fs; lin.p.trunc = trunc_dist;
  lin.aux = reinterpret_cast<const float4*>(loss_inputs);
  lin.gsdf_sorted = grad_sdf; lin.loss_out = loss_slots;
  lin.inv_n = 1.0f / (float)n;
  return sdf_fwd_impl(grid, mlp, packed, x, n, sdf, relu_mask, nullptr, stream, &lin);
}

// levels (with a gradient requested) the owner-computes pull covers for this grid
static int pull_plan(const miso_grid_t* grid, int32_t tiles_per_axis, GridK* g, int* C, uint32_t* mask) {
  bool v4;
  int rc = convert_grid(grid, g, false, &v4);
  if (rc) return rc;
  *mask = 0;
  *C = g->lv[0].C;
  { int t3_[3]; if (!tiles_xyz(tiles_per_axis, t3_)) return MISO_E_BADARG; }
  if (!v4 || (*C != 4 && *C != 8)) return MISO_OK;
  for (int l = 0; l < g->n_levels; ++l)
    if (g->lv[l].C != *C) return MISO_OK;
  *mask = plan_grad_pull(*g, tiles_per_axis);
  return MISO_OK;
}

uint32_t miso_sdf_bwd_scattered_levels(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n) {
  GridK g; int C; uint32_t mask;
  if (pull_plan(grid, tiles_per_axis, &g, &C, &mask)) return 0;
  const uint32_t owned = mask & ~plan_push(g, tiles_per_axis, n, mask);
  uint32_t out = 0;
  for (int l = 0; l < g.n_levels; ++l)
    if (g.lv[l].grad && !((owned >> l) & 1u)) out |= 1u << l;
  return out;
}

uint32_t miso_sdf_bwd_push_levels(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n) {
  GridK g; int C; uint32_t mask;
  if (pull_plan(grid, tiles_per_axis, &g, &C, &mask)) return 0;
  return plan_push(g, tiles_per_axis, n, mask);
}

int miso_grad_pull_on_matrix_cores(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n, int64_t ld_d) {
  GridK g; int C; uint32_t mask;
  if (pull_plan(grid, tiles_per_axis, &g, &C, &mask) || !mask) return 0;
  int T3[3];
  if (!tiles_xyz(tiles_per_axis, T3)) return 0;
  return mc_pull_ok(g, C, T3, mask & ~plan_push(g, tiles_per_axis, n, mask), n, ld_d) ? 1 : 0;
}

uint32_t miso_grad_pull_levels(const miso_grid_t* grid, int32_t tiles_per_axis) {
  GridK g; int C; uint32_t mask;
  if (pull_plan(grid, tiles_per_axis, &g, &C, &mask)) return 0;
  return mask;
}

static int grad_pull_impl(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, const float* dfeat,
                          int64_t ld_d, int32_t rows_in_caller_order, const float* gg_x, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  if (n < 0 || (n > 0 && (!sorted->xn_sorted || !dfeat)) || ((uintptr_t)dfeat & 15u) != 0 || (ld_d & 3) != 0)
    return MISO_E_BADARG;
  GridK g; int C; uint32_t pull;
  rc = pull_plan(grid, sorted->tiles_per_axis, &g, &C, &pull);
  if (rc) return rc;
  if (ld_d < g.F) return MISO_E_BADARG;
  if (pull && !gg_x && !pull_serviceable(sorted, n, ld_d)) return MISO_E_UNSUPPORTED;
  for (int l = 0; l < g.n_levels; ++l)
    if (g.lv[l].grad && !((pull >> l) & 1u)) return MISO_E_UNSUPPORTED;   // every requested level must be pullable
  if (gg_x && !(g.flags & MISO_F_COORDS_NORMALIZED))
    for (int a = 0; a < 3; ++a) g.gscale[a] = 2.0f / (g.bmax[a] - g.bmin[a]);   // d xn / d x (axis_coord's m)
  return (int)launch_grad_pull(g, C, sorted->tiles_per_axis, sorted->tile_offsets, sorted->xn_sorted, dfeat, ld_d,
                               rows_in_caller_order ? sorted->perm : nullptr, pull,
                               (grid->flags & MISO_F_GRAD_OVERWRITE) ? 1 : 0, gg_x, sorted->pull_queue,
                               sorted->pull_queue_ints, (hipStream_t)stream, 0u, n);
}

int miso_grad_pull(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, const float* dfeat,
                   int64_t ld_d, int32_t rows_in_caller_order, void* stream) {
  return grad_pull_impl(grid, sorted, n, dfeat, ld_d, rows_in_caller_order, nullptr, stream);
}

int miso_grad_pull_dx(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, const float* grad_feats,
                      int64_t ld_g, const float* gg_x, void* stream) {
  if (!gg_x) return MISO_E_BADARG;
  return grad_pull_impl(grid, sorted, n, grad_feats, ld_g, 1, gg_x, stream);
}

int64_t miso_sdf_bwd_workspace_floats(const miso_grid_t* grid, int64_t n) {
  GridK g;
  if (n < 0 || convert_grid(grid, &g, false, nullptr)) return 0;
  return n * g.F;
}

int miso_sdf_bwd_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const miso_sorted_t* sorted, int64_t n, const float* grad_sdf,
                        const uint32_t* relu_mask, float* grad_x, float* workspace, void* stream) {
  int rc = check_sorted(sorted, n);
  if (rc) return rc;
  return sdf_bwd_impl(grid, mlp, packed, nullptr, n, grad_sdf, relu_mask, grad_x, sorted, workspace,
                      stream);
}

// One training iteration of a frozen-decoder submap: forward + mapping loss + decoder backward in ONE launch
// (sdf_train_kernel).  sorted != nullptr: a binned batch -- the levels the pull / push can form get their gradient from
// the d-feat rows left in `workspace`, the others (bricks beyond the pull's reach) are scattered with float atomics from
// the kernel itself.  sorted == nullptr: an unbinned batch (x in the caller's order), every level scattered.
static int sdf_train_impl(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x,
                          const miso_sorted_t* sorted, int64_t n, int loss_type, float weight_sdf, float weight_fs,
                          float trunc_dist, const float* loss_inputs, float* sdf, float* loss_slots,
                          const int32_t* n_live, float* workspace, void* stream) {
  if ((loss_type != 1 && loss_type != 2) || !loss_slots || !packed || n < 0 || (n > 0 && !loss_inputs))
    return MISO_E_BADARG;
  if ((((uintptr_t)loss_inputs | (uintptr_t)packed | (uintptr_t)workspace) & 15u) != 0) return MISO_E_BADARG;
  if (sorted && !sorted->xn_sorted && n > 0) return MISO_E_BADARG;
  if (!sorted && n > 0 && !x) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(grid, &g, true, &v4);
  if (rc) return rc;
  int C, L, H, NH;
  rc = fused_shape(g, v4, mlp, &C, &L, &H, &NH);
  if (rc) return rc;
  uint32_t want = 0;
  for (int l = 0; l < g.n_levels; ++l)
    if (g.lv[l].grad && !((g.ignore_mask >> l) & 1u)) want |= 1u << l;
  if (!want) return MISO_E_UNSUPPORTED;
  // levels formed from the d-feat rows (pull or push); the rest is scattered from the kernel
  // (an ignored level with a gradient buffer stays in `pull`: the pull writes its zeros, as miso_sdf_bwd_sorted does)
  const uint32_t pull = (sorted && workspace) ? plan_grad_pull(g, sorted->tiles_per_axis) : 0u;
  const uint32_t push = pull ? plan_push(g, sorted->tiles_per_axis, n, pull) : 0u;
  if (pull && !pull_serviceable(sorted, n, g.F)) return MISO_E_UNSUPPORTED;
  const uint32_t scat = want & ~pull;
  hipStream_t st = (hipStream_t)stream;
  const bool overwrite = sorted && (grid->flags & MISO_F_GRAD_OVERWRITE) != 0;
  if (overwrite && !(grid->flags & MISO_F_GRAD_ZEROED)) {
    for (int l = 0; l < g.n_levels; ++l) {      // pushed and scattered levels are added to with atomics: they start from zero
      const LevelK& lv = g.lv[l];
      if (!lv.grad || !(((push | scat) >> l) & 1u)) continue;
      size_t span = (size_t)(lv.C - 1) * lv.sC + (size_t)(lv.X - 1) * lv.sX + (size_t)(lv.Y - 1) * lv.sY +
                    (size_t)(lv.Z - 1) * lv.sZ + 1;
      hipError_t e = launch_zero_fill(lv.grad, (int64_t)span, st);
      if (e != hipSuccess) return (int)e;
    }
  }
  if (n == 0) {
    hipError_t e = launch_zero_words(loss_slots, MISO_LOSS_SLOTS * 2, st);
    if (e != hipSuccess) return (int)e;
  } else {
    LossInK lin;
    memset(&lin, 0, sizeof(lin));
    lin.p.loss_type = loss_type; lin.p.w_sdf = weight_sdf; lin.p.w_fs = weight_fs; lin.p.trunc = trunc_dist;
    lin.aux = reinterpret_cast<const float4*>(loss_inputs);
    lin.loss_out = loss_slots;
    lin.inv_n = 1.0f / (float)n;
    lin.n_live = n_live;
    GridK gp = g;
    if (sorted) x = sorted_points(&gp, sorted);
    if (sorted && !sorted->perm) gp.flags |= MISO_F_INDEX_IN_XN;
    rc = (int)launch_sdf_train(C, L, H, NH, gp, packed, x, n, sdf, sorted ? sorted->perm : nullptr, lin,
                               pull ? workspace : nullptr, pull, scat != 0, st);
    if (rc) return rc;
  }
  if (!pull) return MISO_OK;
  return (int)launch_grad_pull(g, C, sorted->tiles_per_axis, sorted->tile_offsets, sorted->xn_sorted, workspace, g.F,
                               nullptr, pull, overwrite ? 1 : 0, nullptr, sorted->pull_queue, sorted->pull_queue_ints,
                               st, push, n);
}

int miso_sdf_train_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                          const miso_sorted_t* sorted, int64_t n, int loss_type, float weight_sdf, float weight_fs,
                          float trunc_dist, const float* loss_inputs, float* sdf, float* loss_slots,
                          const int32_t* n_live, float* workspace, void* stream) {
  int rc = check_sorted(sorted, n, true);
  if (rc) return rc;
  if (n > 0 && !workspace) return MISO_E_BADARG;
  return sdf_train_impl(grid, mlp, packed, nullptr, sorted, n, loss_type, weight_sdf, weight_fs, trunc_dist,
                        loss_inputs, sdf, loss_slots, n_live, workspace, stream);
}

int miso_sdf_train(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                   int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* loss_inputs,
                   float* sdf, float* loss_slots, void* stream) {
  return sdf_train_impl(grid, mlp, packed, x, nullptr, n, loss_type, weight_sdf, weight_fs, trunc_dist, loss_inputs,
                        sdf, loss_slots, nullptr, nullptr, stream);
}

int miso_pair_latent(const miso_grid_t* dst_grid, const float* pose, const float* coords_src,
                     const float* feats_src, int64_t ld_feats, int64_t n, int loss_type, double* out,
                     void* stream) {
  if (n < 0 || !pose || !out || (loss_type != 1 && loss_type != 2)) return MISO_E_BADARG;
  if (reinterpret_cast<uintptr_t>(out) & 7u) return MISO_E_BADARG;      // fp64 atomics
  if (n > 0 && (!coords_src || !feats_src)) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(dst_grid, &g, true, &v4);
  if (rc) return rc;
  if (g.flags & (MISO_F_COORDS_NORMALIZED | MISO_F_ALIGN_CORNERS | MISO_F_PAD_BORDER)) return MISO_E_UNSUPPORTED;
  if (ld_feats < g.F) return MISO_E_BADARG;
  return (int)launch_pair_latent(g, v4, pose, coords_src, feats_src, ld_feats, n, loss_type, out,
                                 (hipStream_t)stream);
}

int miso_overlap_count(const float* pose, const float* coords_src, int64_t n, const float* bound_min,
                       const float* bound_max, float* count_out, void* stream) {
  if (n < 0 || !pose || !bound_min || !bound_max || !count_out || (n > 0 && !coords_src)) return MISO_E_BADARG;
  if (n >= ((int64_t)1 << 24)) return MISO_E_TOOLARGE;     // the fp32 count stays exact
  return (int)launch_overlap_count(pose, coords_src, n, bound_min, bound_max, count_out, (hipStream_t)stream);
}

int miso_align_src_boxes(const float* coords, int64_t n, float* boxes, void* stream) {
  if (n < 0 || (n > 0 && (!coords || !boxes))) return MISO_E_BADARG;
  return (int)launch_src_boxes(coords, n, boxes, (hipStream_t)stream);
}

int64_t miso_align_plan_bytes(int32_t n_pairs) {
  return n_pairs < 0 ? 0 : (int64_t)(n_pairs > 0 ? n_pairs : 1) * (int64_t)sizeof(AlignPairK);
}

int miso_align_plan_build(const miso_align_pair_t* pairs, miso_align_t* cfg, void* plan_host) {
  if (!cfg || cfg->n_pairs < 0 || cfg->n_submaps < 1 || cfg->n_submaps > 64) return MISO_E_BADARG;
  if (cfg->n_pairs > 0 && (!pairs || !plan_host)) return MISO_E_BADARG;
  AlignPairK* out = reinterpret_cast<AlignPairK*>(plan_host);
  bool v4_all = true;
  int64_t max_n = 0, max_gate = 0, max_rows = 0;
  for (int p = 0; p < cfg->n_pairs; ++p) {
    const miso_align_pair_t& in = pairs[p];
    if (in.src < 0 || in.src >= cfg->n_submaps || in.dst < 0 || in.dst >= cfg->n_submaps || in.src == in.dst)
      return MISO_E_BADARG;
    if (in.n < 0 || in.gate_n < 0 || (in.n > 0 && (!in.coords_src || !in.feats_src))) return MISO_E_BADARG;
    if (in.gate_n >= ((int64_t)1 << 24)) return MISO_E_TOOLARGE;     // the fp32 count stays exact
    AlignPairK& d = out[p];
    memset(&d, 0, sizeof(d));
    bool v4;
    int rc = convert_grid(&in.dst_grid, &d.g, true, &v4);
    if (rc) return rc;
    if (d.g.flags & (MISO_F_COORDS_NORMALIZED | MISO_F_ALIGN_CORNERS | MISO_F_PAD_BORDER)) return MISO_E_UNSUPPORTED;
    if (in.ld_feats < d.g.F) return MISO_E_BADARG;
    v4_all = v4_all && v4;
    d.p = in.coords_src; d.fsrc = in.feats_src; d.ld = in.ld_feats; d.n = in.n;
    d.gate_p = in.gate_n > 0 ? in.gate_coords : nullptr; d.gate_n = in.gate_n;
    if (in.gate_n > 0 && in.gate_axis[0]) {
      if (!in.gate_axis[1] || !in.gate_axis[2]) return MISO_E_BADARG;
      int64_t prod = 1;
      for (int a = 0; a < 3; ++a) {
        if (in.gate_dims[a] < 1) return MISO_E_BADARG;
        d.gate_ax[a] = in.gate_axis[a]; d.gate_dim[a] = in.gate_dims[a];
        prod *= in.gate_dims[a];
      }
      if (prod != in.gate_n) return MISO_E_BADARG;
      d.gate_p = in.gate_axis[0];      // non-null marks the gate as on; the lattice path does not read it as points
      if ((int64_t)in.gate_dims[1] * in.gate_dims[2] > max_rows) max_rows = (int64_t)in.gate_dims[1] * in.gate_dims[2];
    } else if (in.gate_n > 0 && !in.gate_coords) {
      return MISO_E_BADARG;
    }
    d.src = in.src; d.dst = in.dst; d.n_ch = (float)d.g.F;
    d.boxes = in.n > 0 ? in.src_boxes : nullptr;
    if (in.n > max_n) max_n = in.n;
    if (d.gate_p && !d.gate_ax[0] && in.gate_n > max_gate) max_gate = in.gate_n;      // point-list gates only (grid sizing)
  }
  cfg->vec4 = v4_all ? 1 : 0;
  cfg->max_n = max_n;
  cfg->max_gate_n = max_gate;
  cfg->max_gate_rows = max_rows;
  return MISO_OK;
}

int64_t miso_align_state_layout(int32_t n_submaps, int32_t n_pairs, int32_t ring_iters, int32_t save_poses,
                                int64_t* offsets) {
  if (n_submaps < 1 || n_submaps > 64 || n_pairs < 0 || ring_iters < 0) return 0;
  const AlignLayout L = align_layout(n_submaps, n_pairs, ring_iters, save_poses);
  if (offsets) {
    const int64_t o[12] = {L.params, L.pose, L.out, L.cnt, L.pair_loss, L.flat, L.adam_m, L.adam_v, L.ctrl, L.ring,
                           L.ring_row, L.adam_t};
    for (int i = 0; i < 12; ++i) offsets[i] = o[i];
  }
  return L.total;
}

static int align_k(const miso_align_t* c, AlignK* k) {
  if (!c || c->n_submaps < 1 || c->n_submaps > 64 || c->n_pairs < 0 || c->ring_iters < 0) return MISO_E_BADARG;
  if (!c->R0 || !c->t0 || !c->state || (c->n_pairs > 0 && !c->plan)) return MISO_E_BADARG;
  if (c->loss_type != 1 && c->loss_type != 2) return MISO_E_BADARG;
  if (((uintptr_t)c->state & 15u) != 0) return MISO_E_BADARG;
  memset(k, 0, sizeof(*k));
  k->S = c->n_submaps; k->P = c->n_pairs; k->loss_type = c->loss_type; k->ring_iters = c->ring_iters;
  k->save_poses = c->save_poses;
  k->align_weight = c->align_weight; k->overlap_thresh = c->overlap_thresh; k->reg_weight = c->reg_weight;
  k->reg_rad = c->reg_thresh_rad; k->reg_m = c->reg_thresh_m; k->rel_thresh = c->rel_change_thresh;
  k->lr = c->lr; k->b1 = c->beta1; k->b2 = c->beta2; k->eps = c->eps;
  k->R0 = c->R0; k->t0 = c->t0; k->plan = reinterpret_cast<const AlignPairK*>(c->plan); k->state = c->state;
  k->L = align_layout(k->S, k->P, k->ring_iters, k->save_poses);
  return MISO_OK;
}

int miso_align_iteration_a(const miso_align_t* cfg, void* stream) {
  AlignK k;
  int rc = align_k(cfg, &k);
  if (rc) return rc;
  return (int)launch_align_a(k, cfg->max_n, cfg->max_gate_n, cfg->max_gate_rows, cfg->vec4 != 0, cfg->poses_ready != 0,
                             (hipStream_t)stream);
}

int miso_align_iteration_b(const miso_align_t* cfg, void* stream) {
  AlignK k;
  int rc = align_k(cfg, &k);
  if (rc) return rc;
  return (int)launch_align_b(k, (hipStream_t)stream);
}

int miso_lm_normal_eq(const float* coords_frame, const float* R_frame, const float* grad_sdf_x,
                      const float* sdf, const float* target, int64_t n, int loss_type, float gm_scale,
                      float* out, void* stream) {
  if (n < 0 || !out || !R_frame) return MISO_E_BADARG;
  if (n > 0 && (!coords_frame || !grad_sdf_x || !sdf || !target)) return MISO_E_BADARG;
  if (loss_type != 2 && loss_type != 3) return MISO_E_UNSUPPORTED;
  if (loss_type == 3 && !(gm_scale > 0.0f)) return MISO_E_BADARG;
  return (int)launch_lm_normal_eq(coords_frame, R_frame, grad_sdf_x, sdf, target, n, loss_type, gm_scale, out,
                                  (hipStream_t)stream);
}

static int lm_track_args(const miso_grid_t* grid, const miso_lm_track_t* a, LmTrackK* out) {
  LmTrackK& k = *out;
  memset(&k, 0, sizeof(k));
  k.x = a->coords_frame; k.gt = a->target; k.valid = a->valid; k.frame_ids = a->frame_ids;
  k.s_gt = a->stride_target; k.s_valid = a->stride_valid; k.s_fid = a->stride_frame_ids;
  k.valid_is_bool = a->valid_is_bool; k.n = a->n; k.kf = a->keyframe_id; k.trunc = a->trunc_dist;
  k.Rwk = a->R_base; k.twk = a->t_base; k.dr = a->rot_correction; k.dt = a->trans_correction;
  k.pose = a->pose; k.xw = a->coords_world; k.sums = a->sums; k.info = a->info; k.clean = a->sanitized;
  for (int i = 0; i < 3; ++i) { k.bmin[i] = grid->bound_min[i]; k.bmax[i] = grid->bound_max[i]; }
  k.lm_lambda = a->lm_lambda;
  return MISO_OK;
}

// after the head (pose + transform) has run: the later kernels read the sanitised copies
static void lm_track_use_clean(LmTrackK* k) {
  if (!k->clean) return;
  k->x = k->clean; k->gt = k->clean + 3 * k->n; k->s_gt = 1;
  k->valid = k->clean + 4 * k->n; k->s_valid = 1; k->valid_is_bool = 0;
  k->clean = nullptr;
}

int miso_track_adam_step(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const miso_track_adam_t* t,
                         void* stream) {
  if (!t || !grid) return MISO_E_BADARG;
  const miso_lm_track_t* a = &t->s;
  if (a->n < 0 || !a->R_base || !a->t_base || !a->rot_correction || !a->trans_correction || !a->pose || !a->sums ||
      !a->info || !t->adam_table || t->adam_table_len < 1 || !t->state || t->ring_len < 0 || (t->ring_len > 0 && !t->loss_ring))
    return MISO_E_BADARG;
  if (a->n > 0 && (!a->coords_frame || !a->target || !a->coords_world || !a->sdf || !a->grad || !a->relu_mask || !t->grad_pred))
    return MISO_E_BADARG;
  if (a->stride_target < 0 || a->stride_valid < 0 || a->stride_frame_ids < 0) return MISO_E_BADARG;
  if (t->loss_type < 1 || t->loss_type > 3) return MISO_E_UNSUPPORTED;
  if (t->loss_type == 3 && !(t->gm_scale > 0.0f)) return MISO_E_BADARG;
  for (int l = 0; l < grid->n_levels && l < MISO_MAX_LEVELS; ++l)
    if (grid->level[l].grad) return MISO_E_BADARG;
  TrackAdamK k;
  memset(&k, 0, sizeof(k));
  lm_track_args(grid, a, &k.s);
  k.loss_type = t->loss_type; k.weight_sdf = t->weight_sdf; k.gm_scale = t->gm_scale;
  k.gpred = t->grad_pred; k.gx = a->grad; k.sdf = a->sdf;
  k.table = reinterpret_cast<const AdamScalars*>(t->adam_table); k.table_len = t->adam_table_len;
  k.state = t->state; k.ring = t->loss_ring; k.ring_len = t->ring_len;
  hipStream_t st = (hipStream_t)stream;
  int rc = (int)launch_lm_track_head(k.s, st);
  if (rc) return rc;
  lm_track_use_clean(&k.s);
  if (a->n > 0) {
    rc = sdf_fwd_impl(grid, mlp, packed, a->coords_world, a->n, a->sdf, a->relu_mask, nullptr, stream);
    if (rc) return rc;
    rc = (int)launch_track_loss(k, st);
    if (rc) return rc;
    rc = sdf_bwd_impl(grid, mlp, packed, a->coords_world, a->n, t->grad_pred, a->relu_mask, a->grad, nullptr, nullptr, stream);
    if (rc) return rc;
  }
  return (int)launch_track_tail(k, st);
}

int miso_lm_track_step(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const miso_lm_track_t* a,
                       void* stream) {
  if (!a || a->n < 0 || !a->R_base || !a->t_base || !a->rot_correction || !a->trans_correction || !a->pose || !a->sums ||
      !a->info)
    return MISO_E_BADARG;
  if (a->n > 0 && (!a->coords_frame || !a->target || !a->coords_world || !a->sdf || !a->grad || !a->ones || !a->relu_mask))
    return MISO_E_BADARG;
  if (a->stride_target < 0 || a->stride_valid < 0 || a->stride_frame_ids < 0) return MISO_E_BADARG;
  if (a->loss_type != 2 && a->loss_type != 3) return MISO_E_UNSUPPORTED;
  if (a->loss_type == 3 && !(a->gm_scale > 0.0f)) return MISO_E_BADARG;
  if (!grid) return MISO_E_BADARG;
  for (int l = 0; l < grid->n_levels && l < MISO_MAX_LEVELS; ++l)
    if (grid->level[l].grad) return MISO_E_BADARG;      // the step wants d sdf / d x only
  LmTrackK k;
  lm_track_args(grid, a, &k);
  hipStream_t st = (hipStream_t)stream;
  int rc = (int)launch_lm_track_head(k, st);
  if (rc) return rc;
  lm_track_use_clean(&k);
  if (a->n > 0) {
    rc = sdf_fwd_impl(grid, mlp, packed, a->coords_world, a->n, a->sdf, a->relu_mask, nullptr, stream);
    if (rc) return rc;
    rc = sdf_bwd_impl(grid, mlp, packed, a->coords_world, a->n, a->ones, a->relu_mask, a->grad, nullptr, nullptr, stream);
    if (rc) return rc;
  }
  return (int)launch_lm_track_tail(k, a->grad, a->sdf, a->loss_type, a->gm_scale, st);
}

int64_t miso_pull_queue_ints(int64_t n) { return n < 0 ? 0 : pull_queue_ints(n); }

int64_t miso_sample_rays_workspace_bytes(int64_t n_rays, int32_t n_frames) {
  if (n_rays < 0 || n_frames < 0) return 0;
  return (int64_t)sample_rays_workspace_bytes(n_rays, n_frames);
}

int miso_sample_rays(const miso_ray_frames_t* f, const miso_ray_sampling_t* c, int64_t n_rays, const int64_t* pix_b,
                     const int64_t* pix_h, const int64_t* pix_w, const float* u, const float* g, void* workspace,
                     float* coords_frame, int64_t* sample_frame_ids, float* aux, float* pc_world, float* z_vals,
                     int32_t* counts, void* stream) {
  if (!f || !c || n_rays < 0 || !counts) return MISO_E_BADARG;
  if (c->n_strat < 0 || c->n_surf < 0 || c->n_strat + c->n_surf < 1) return MISO_E_BADARG;
  if (c->n_strat > MISO_RAY_MAX_BINS) return MISO_E_UNSUPPORTED;
  const int64_t S = c->n_strat + c->n_surf;
  if (n_rays * S >= (int64_t(1) << 31)) return MISO_E_TOOLARGE;
  if (n_rays > 0) {
    if (!f->depth || !f->T_WC || !f->R_wk || !f->t_wk || f->n_frames < 1 || f->H < 1 || f->W < 1) return MISO_E_BADARG;
    if (!pix_h || !pix_w || !workspace || !coords_frame || !sample_frame_ids || !aux) return MISO_E_BADARG;
    if (!pix_b && c->rays_per_frame < 1) return MISO_E_BADARG;
    if ((c->n_strat > 0 && !u) || (c->n_surf > 1 && !g)) return MISO_E_BADARG;
    if (reinterpret_cast<uintptr_t>(aux) & 15u) return MISO_E_BADARG;
  }
  float lin[MISO_RAY_MAX_BINS + 1];
  if (c->bin_edges) {
    for (int i = 0; i <= c->n_strat; ++i) lin[i] = c->bin_edges[i];
  } else {   // at::linspace's two-sided float formula
    const int steps = c->n_strat + 1;
    const float step = steps > 1 ? 1.0f / (float)(steps - 1) : 0.0f;
    for (int i = 0; i < steps; ++i) lin[i] = i < steps / 2 ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
  }
  return (int)launch_sample_rays(*f, *c, lin, n_rays, pix_b, pix_h, pix_w, u, g, workspace, coords_frame,
                                 sample_frame_ids, aux, pc_world, z_vals, counts, (hipStream_t)stream);
}

int miso_mapping_loss(int loss_type, float weight_sdf, float weight_fs, float trunc_dist,
                      const float* pred, const float* target, const float* valid, const float* sign,
                      const float* weight, int64_t n, float* grad_pred, float* grad_pred_fs,
                      float* loss_out, void* stream) {
  if (n < 0 || (loss_type != 1 && loss_type != 2) || !loss_out) return MISO_E_BADARG;
  if (n > 0 && (!pred || !target || !grad_pred)) return MISO_E_BADARG;
  return (int)launch_mapping_loss(loss_type, weight_sdf, weight_fs, trunc_dist, pred, target, valid,
                                  sign, weight, n, grad_pred, grad_pred_fs, loss_out, (hipStream_t)stream);
}

int miso_adam_dense(float* param, float* grad, float* exp_avg, float* exp_avg_sq, int64_t numel,
                    double lr, double beta1, double beta2, double eps, int32_t step, int zero_grad,
                    void* stream) {
  if (numel < 0 || step < 1 || (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq)))
    return MISO_E_BADARG;
  return (int)launch_adam(param, grad, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step,
                          zero_grad, (hipStream_t)stream);
}

int miso_adam_active(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active, int64_t numel,
                     double lr, double beta1, double beta2, double eps, int32_t step, int zero_grad,
                     const float* guard, void* stream) {
  if (numel < 0 || step < 1 || (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !active)))
    return MISO_E_BADARG;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15u)
    return MISO_E_BADARG;
  return (int)launch_adam_active(param, grad, exp_avg, exp_avg_sq, active, numel, lr, beta1, beta2, eps, step,
                                 zero_grad, guard, (hipStream_t)stream, nullptr, nullptr, 0);
}

int miso_adam_touched(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active, uint8_t* touched,
                      int64_t numel, double lr, double beta1, double beta2, double eps, int32_t step, int zero_grad,
                      const float* guard, void* stream) {
  if (numel < 0 || step < 1 || (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !active || !touched)))
    return MISO_E_BADARG;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15u)
    return MISO_E_BADARG;
  if ((((uintptr_t)active) | ((uintptr_t)touched)) & 3u) return MISO_E_BADARG;      // read four flag bytes at a time
  return (int)launch_adam_touched(param, grad, exp_avg, exp_avg_sq, active, touched, numel, lr, beta1, beta2, eps,
                                  step, zero_grad, guard, (hipStream_t)stream, nullptr, nullptr, 0);
}

int miso_adam_scalars_table(double lr, double beta1, double beta2, double eps, int32_t first_step, int32_t count,
                            float* host_out) {
  if (first_step < 1 || count < 0 || (count > 0 && !host_out)) return MISO_E_BADARG;
  adam_scalars_table(lr, beta1, beta2, eps, first_step, count, host_out);
  return MISO_OK;
}

int miso_loss_total_bump(const float* loss_slots, int32_t n_floats, float* total, int32_t* step, void* stream) {
  if (!loss_slots || n_floats < 1 || !total) return MISO_E_BADARG;
  return (int)launch_loss_total_bump(loss_slots, n_floats, total, step, nullptr, 0, (hipStream_t)stream);
}

int miso_loss_total_bump_host(const float* loss_slots, int32_t n_floats, float* total, int32_t* step2, float* host_ring,
                              int32_t ring_len, void* stream) {
  if (!loss_slots || n_floats < 1 || !total || !step2 || !host_ring || ring_len < 1 || (ring_len & (ring_len - 1)))
    return MISO_E_BADARG;
  return (int)launch_loss_total_bump(loss_slots, n_floats, total, step2, host_ring, ring_len, (hipStream_t)stream);
}

int miso_adam_bump(int32_t* step, const float* guard, void* stream) {
  if (!step) return MISO_E_BADARG;
  return (int)launch_adam_bump(step, guard, (hipStream_t)stream);
}

int miso_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active, uint8_t* touched,
                       int64_t numel, const float* table, int32_t table_len, const int32_t* step, int zero_grad,
                       const float* guard, void* stream) {
  if (numel < 0 || table_len < 1 || !table || !step ||
      (numel > 0 && (!param || !grad || !exp_avg || !exp_avg_sq || !active)))
    return MISO_E_BADARG;
  if ((((uintptr_t)param) | ((uintptr_t)grad) | ((uintptr_t)exp_avg) | ((uintptr_t)exp_avg_sq)) & 15u)
    return MISO_E_BADARG;
  if (touched && ((((uintptr_t)active) | ((uintptr_t)touched)) & 3u)) return MISO_E_BADARG;
  if (touched)
    return (int)launch_adam_touched(param, grad, exp_avg, exp_avg_sq, active, touched, numel, 1e-3, 0.9, 0.999, 1e-8, 1,
                                    zero_grad, guard, (hipStream_t)stream, table, step, table_len);
  return (int)launch_adam_active(param, grad, exp_avg, exp_avg_sq, active, numel, 1e-3, 0.9, 0.999, 1e-8, 1, zero_grad,
                                 guard, (hipStream_t)stream, table, step, table_len);
}

static int adam_tensors_ok(const miso_adam_tensor_t* tensors, int32_t n_tensors) {
  if (!tensors || n_tensors < 1 || n_tensors > MISO_ADAM_MAX_TENSORS) return 0;
  for (int i = 0; i < n_tensors; ++i) {
    const miso_adam_tensor_t& t = tensors[i];
    if (t.numel < 0 || (t.numel > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq || !t.active))) return 0;
    if ((((uintptr_t)t.param) | ((uintptr_t)t.grad) | ((uintptr_t)t.exp_avg) | ((uintptr_t)t.exp_avg_sq)) & 15u) return 0;
    if (t.touched && ((((uintptr_t)t.active) | ((uintptr_t)t.touched)) & 3u)) return 0;      // four flag bytes at a time
  }
  return 1;
}

int miso_adam_active_multi(const miso_adam_tensor_t* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                           double eps, int32_t step, const float* guard, void* stream) {
  if (!adam_tensors_ok(tensors, n_tensors) || step < 1) return MISO_E_BADARG;
  return (int)launch_adam_active_multi(tensors, n_tensors, lr, beta1, beta2, eps, step, nullptr, 0, nullptr, guard,
                                       (hipStream_t)stream);
}

int miso_adam_step_dev_multi(const miso_adam_tensor_t* tensors, int32_t n_tensors, const float* table, int32_t table_len,
                             const int32_t* step, const float* guard, void* stream) {
  if (!adam_tensors_ok(tensors, n_tensors) || table_len < 1 || !table || !step) return MISO_E_BADARG;
  return (int)launch_adam_active_multi(tensors, n_tensors, 1e-3, 0.9, 0.999, 1e-8, 1, table, table_len, step, guard,
                                       (hipStream_t)stream);
}

int miso_mapping_batch(const float* R, const float* t, int32_t n_poses, const int64_t* table, int64_t table_len,
                       const int64_t* frame_ids, const float* coords_frame, const float* target, const void* valid,
                       const float* sign, const float* weight, const int64_t* col_strides, int valid_is_bool, int64_t n,
                       float* coords_world, float* loss_rows, int sanitize, void* stream) {
  if (n < 0 || n_poses < 1 || table_len < 1 || !R || !t || !table) return MISO_E_BADARG;
  if (col_strides)
    for (int i = 0; i < 4; ++i)
      if (col_strides[i] < 0) return MISO_E_BADARG;
  if (n > 0 && (!frame_ids || !coords_frame || !target || !coords_world || !loss_rows)) return MISO_E_BADARG;
  if (((uintptr_t)loss_rows & 15u) != 0) return MISO_E_BADARG;
  return (int)launch_mapping_batch(R, t, n_poses, table, table_len, frame_ids, coords_frame, target, valid, sign, weight,
                                   n, coords_world, loss_rows, col_strides, valid_is_bool, sanitize, (hipStream_t)stream);
}

int miso_mapping_loss_rows(int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* pred,
                           const float* loss_rows, int64_t n, float* grad_pred, float* loss_out, void* stream) {
  if ((loss_type != 1 && loss_type != 2) || n < 0 || !loss_out || (n > 0 && (!pred || !loss_rows || !grad_pred)))
    return MISO_E_BADARG;
  if (((uintptr_t)loss_rows & 15u) != 0) return MISO_E_BADARG;
  return (int)launch_mapping_loss_rows(loss_type, weight_sdf, weight_fs, trunc_dist, pred, loss_rows, n, grad_pred,
                                       loss_out, (hipStream_t)stream);
}

int miso_rigid_by_index(const float* R, const float* t, const int64_t* idx, const float* x, int64_t n, int32_t n_poses,
                        int transpose, float* y, void* stream) {
  if (n < 0 || n_poses < 1 || !R || (n > 0 && (!idx || !x || !y))) return MISO_E_BADARG;
  return (int)launch_rigid_by_index(R, t, idx, x, n, n_poses, transpose, y, (hipStream_t)stream);
}

static int mc_check_dims(int32_t nx, int32_t ny, int32_t nz) {
  if (nx < 1 || ny < 1 || nz < 1) return MISO_E_BADARG;
  // cell ids are 32-bit; sample indices (and keys = 3 * index + axis) are 64-bit
  if ((int64_t)nx * ny * nz >= ((int64_t)1 << 31)) return MISO_E_TOOLARGE;
  return 0;
}

int64_t miso_mc_words(int32_t nx, int32_t ny, int32_t nz) {
  if (mc_check_dims(nx, ny, nz)) return -1;
  return mc_words(nx, ny, nz);
}

int miso_grid_pool_avg(const float* coords, const float* features, int64_t n, int32_t d, int64_t ld_features,
                       const float* bound_min, float cell_size, int32_t nx, int32_t ny, int32_t nz, float* pooled,
                       int32_t* counts, void* stream) {
  if (n < 0 || d < 1 || nx < 1 || ny < 1 || nz < 1 || !bound_min || !pooled || !counts || !(cell_size > 0.0f)) return MISO_E_BADARG;
  if (n > 0 && (!coords || !features || ld_features < d)) return MISO_E_BADARG;
  if ((int64_t)nx * ny * nz * d >= ((int64_t)1 << 31)) return MISO_E_TOOLARGE;
  return (int)launch_grid_pool_avg(coords, features, n, d, ld_features, bound_min, cell_size, nx, ny, nz, pooled, counts,
                                   (hipStream_t)stream);
}

// ---- fused atlas query (atlas.hip) ------------------------------------------------------------------------------------
int64_t miso_atlas_plan_bytes(int32_t n_submaps) {
  return n_submaps < 1 ? 0 : (int64_t)n_submaps * (int64_t)sizeof(GridK);
}

int miso_atlas_plan_build(const miso_grid_t* grids, int32_t n_submaps, void* plan_host) {
  if (!grids || !plan_host || n_submaps < 1 || n_submaps > 4096) return MISO_E_BADARG;
  GridK* out = reinterpret_cast<GridK*>(plan_host);
  for (int s = 0; s < n_submaps; ++s) {
    bool v4;
    int rc = convert_grid(&grids[s], &out[s], true, &v4);
    if (rc) return rc;
    if (!v4 || (out[s].flags & MISO_F_COORDS_NORMALIZED)) return MISO_E_UNSUPPORTED;
    if (out[s].n_levels != out[0].n_levels) return MISO_E_UNSUPPORTED;
    for (int l = 0; l < out[s].n_levels; ++l)
      if (out[s].lv[l].C != out[0].lv[0].C) return MISO_E_UNSUPPORTED;
    // (ignore_mask is the caller's: GridAtlas.query_feature passes ignore_level=None, grid_atlas.py:385 -- its grids come
    // without one; a single GridNet queried through MISO_F_ATLAS_NO_BOUND keeps its own)
  }
  return MISO_OK;
}

int miso_atlas_sdf_fwd(const void* plan, int32_t n_submaps, const miso_grid_t* shape, const float* poses,
                       const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n, const float* axis_x,
                       const float* axis_y, const float* axis_z, int32_t nx, int32_t ny, int32_t nz, float* sdf,
                       float* feats, int64_t ld_feats, uint32_t flags, void* stream) {
  if (!plan || !poses || !shape || n_submaps < 1 || n < 0 || (!sdf && !feats)) return MISO_E_BADARG;
  if (flags & ~(MISO_F_EXACT_F32 | MISO_F_ATLAS_NO_BOUND)) return MISO_E_BADARG;
  GridK g; bool v4;
  int rc = convert_grid(shape, &g, false, &v4);
  if (rc) return rc;
  int C = g.lv[0].C, L = g.n_levels, H = 64, NH = 1;
  if (sdf) {
    if (!mlp || !packed || (((uintptr_t)packed) & 15u)) return MISO_E_BADARG;
    rc = fused_shape(g, true, mlp, &C, &L, &H, &NH);
    if (rc) return rc;
  } else {
    for (int l = 0; l < L; ++l)
      if (g.lv[l].C != C) return MISO_E_UNSUPPORTED;
    if (!fused_shape_supported(C, L, H, NH)) return MISO_E_UNSUPPORTED;
  }
  if (feats && ld_feats < g.F) return MISO_E_BADARG;
  AtlasK a;
  memset(&a, 0, sizeof(a));
  a.submaps = reinterpret_cast<const GridK*>(plan);
  a.poses = poses; a.n_submaps = n_submaps; a.n = n; a.sdf = sdf; a.feats = feats; a.ld = ld_feats;
  a.no_bound = (flags & MISO_F_ATLAS_NO_BOUND) ? 1 : 0;
  if (x) {
    a.x = x;
  } else {
    if (!axis_x || !axis_y || !axis_z || nx < 1 || ny < 1 || nz < 1) return MISO_E_BADARG;
    if ((int64_t)nx * ny * nz != n || n >= ((int64_t)1 << 31)) return MISO_E_BADARG;
    a.ax[0] = axis_x; a.ax[1] = axis_y; a.ax[2] = axis_z; a.dim[0] = nx; a.dim[1] = ny; a.dim[2] = nz;
  }
  return (int)launch_atlas_sdf(C, L, H, NH, a, packed, (flags & MISO_F_EXACT_F32) != 0, (hipStream_t)stream);
}

int64_t miso_mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz) {
  if (mc_check_dims(nx, ny, nz)) return -1;
  return mc_workspace_bytes(nx, ny, nz);
}

int miso_mc_classify(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                     int32_t* counts, void* stream) {
  if (int rc = mc_check_dims(nx, ny, nz)) return rc;
  if (!vol || !workspace || !counts || (((uintptr_t)workspace) & 7u)) return MISO_E_BADARG;
  return (int)launch_mc_classify(vol, nx, ny, nz, iso, workspace, counts, (hipStream_t)stream);
}

int miso_mc_emit(int32_t nx, int32_t ny, int32_t nz, void* workspace, const int64_t* offsets, int32_t n_tri_chunks,
                 int64_t capacity_tris, int64_t* faces, void* stream) {
  if (int rc = mc_check_dims(nx, ny, nz)) return rc;
  if (capacity_tris < 0 || n_tri_chunks < 0 || n_tri_chunks > mc_words(nx, ny, nz) || !workspace || !offsets ||
      (capacity_tris > 0 && !faces))
    return MISO_E_BADARG;
  return (int)launch_mc_emit(nx, ny, nz, workspace, offsets, n_tri_chunks, capacity_tris, faces, (hipStream_t)stream);
}

int miso_mc_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                     const int64_t* offsets, int32_t n_vert_chunks, int64_t capacity_verts, float* verts,
                     void* stream) {
  if (int rc = mc_check_dims(nx, ny, nz)) return rc;
  if (capacity_verts < 0 || n_vert_chunks < 0 || n_vert_chunks > mc_words(nx, ny, nz) || !vol || !workspace ||
      !offsets || (capacity_verts > 0 && !verts))
    return MISO_E_BADARG;
  return (int)launch_mc_vertices(vol, nx, ny, nz, iso, workspace, offsets, n_vert_chunks, capacity_verts, verts,
                                 (hipStream_t)stream);
}

int miso_mc_case_table(int8_t* table_host) {
  if (!table_host) return MISO_E_BADARG;
  mc_copy_table(table_host);
  return 0;
}

}  // extern "C"
