/*
 * miso_hip.h -- C ABI of libmiso_hip.so, the MI355X (gfx950) implementation of
 * MISO's encode/decode hot path.
 *
 * Every entry point takes raw DEVICE pointers, sizes and element strides plus a
 * hipStream_t (passed as void*), allocates nothing, launches asynchronously on
 * that stream and returns 0 on success or a non-zero code (MISO_E_* below, or a
 * hipError_t from the launch).  All arithmetic is fp32, index math is 32-bit
 * per level (grids up to 2^31-1 elements) and 64-bit across points.
 *
 * Paths are relative to the reference repository (ExistentialRobotics/MISO).
 *
 * What each entry replaces in the reference:
 *   miso_encode_fwd   ATen grid_sampler_3d forward as called per level by
 *                     grid_opt/models/grid_modules.py:72-95 (FeatureGrid.interpolate)
 *                     + utils.normalize_coordinates (grid_opt/utils/utils.py:22-51)
 *                     + the level loop / torch.cat of grid_opt/utils/utils.py:143-164;
 *                     also third_party/cuda_gridsample_grad2/cuda_gridsample.py:76-95.
 *   miso_encode_bwd   aten::grid_sampler_3d_backward as bound at
 *                     third_party/cuda_gridsample_grad2/cuda_gridsample.py:99-113
 *                     (output_mask = which of grad_grid / grad_x are non-NULL).
 *   miso_encode_bwd2  gridsample_grad2.grad2_3d, pybind at
 *                     third_party/cuda_gridsample_grad2/gridsample_cuda.cpp:39-56,
 *                     kernel gridsample_cuda.cu:212-533, launcher :601-666.
 *   miso_mlp_pack / miso_sdf_fwd / miso_sdf_bwd
 *                     GridNet.forward = query_feature + utils.grid_decode + MLPNet
 *                     (grid_opt/models/grid_net.py:288-325, grid_opt/utils/utils.py:194-208,
 *                     grid_opt/models/modules.py:11-32) and its autograd backward
 *                     with a frozen decoder (configs/rgbd/scannet.yaml:16).
 *   miso_pair_latent  pairwise_loss_latent, grid_opt/align/miso.py:116-211 (L2 / L1),
 *                     with the rigid maps of grid_opt/utils/utils_geometry.py:214-240.
 *   miso_overlap_count GridAtlas.check_submap_intersection, grid_opt/models/grid_atlas.py:405-420.
 *   miso_align_iteration_a / _b
 *                     one iteration of generic_align_multiple_submaps, grid_opt/align/base.py:127-160 (all pairs,
 *                     so3_exp_map backward, trust region, NaN guard, torch.optim.Adam on the pose corrections).
 *   miso_lm_normal_eq Tracker.lm_step, grid_opt/slam/tracker.py:148-212 (J, H = J^T W J,
 *                     g = J^T W r with the L2 / Geman-McClure weights of :139-146).
 *   miso_mapping_loss miso_loss_regression + miso_loss_free_space and their gradient
 *                     w.r.t. the prediction, grid_opt/loss.py:594-635, :668-700, as
 *                     combined by MisoLossMappingBase.compute (loss.py:776-806).
 *   miso_adam_dense / miso_adam_active
 *                     torch.optim.Adam.step on one dense tensor as used by
 *                     grid_opt/trainer.py:196-228 / :410-452.
 *   miso_rigid_by_index  the per-keyframe frame change of a sample batch (grid_opt/loss.py:763-774 around
 *                     transform_points_to, grid_opt/utils/utils_geometry.py:214-225).
 *   miso_mc_classify / miso_mc_emit / miso_mc_vertices
 *                     mcubes.marching_cubes as called by extract_geometry, grid_opt/utils/utils_sdf.py:89-101
 *                     (the step after the path: SDF volume -> triangle mesh).
 *   miso_sample_rays  PosedSdfRgbd.getitem_sdf, grid_opt/datasets/sdf_rgbd.py:381-483 (the step that
 *                     feeds the path): get_batch_data / sample_along_rays
 *                     (grid_opt/utils/utils_sample.py:142-302), bounds_ray (sdf_rgbd.py:525-534), the
 *                     per-keyframe world -> keyframe loop (:436-445) and the truncation labels (:447-455).
 */
#ifndef MISO_HIP_H
#define MISO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISO_MAX_LEVELS 8
#define MISO_MAX_LINEAR 4 /* Linear layers in the decoder: hidden_layers + 2 */

/* return codes (besides hipError_t values, which are > 0 and < 2000) */
#define MISO_OK 0
#define MISO_E_BADARG 2001      /* NULL pointer / negative size / bad flag */
#define MISO_E_UNSUPPORTED 2002 /* shape outside what the fused kernels cover */
#define MISO_E_TOOLARGE 2003    /* a level has >= 2^31 elements */

/* flags */
#define MISO_F_ALIGN_CORNERS 1u    /* grid_sample align_corners=True (MISO uses False) */
#define MISO_F_PAD_BORDER 2u       /* padding_mode='border' (MISO uses 'zeros') */
#define MISO_F_COORDS_NORMALIZED 4u /* x is already in [-1,1]: skip normalize_coordinates */
#define MISO_F_CROWDED 64u         /* hint: the batch crowds a few tiles (ray samples around surfaces and cameras), so
                                     miso_sdf_bwd_sorted pushes every eligible coarse level through the matrix
                                     cores whatever the average density per tile (default: from 100 samples per tile) */
#define MISO_F_EXACT_F32 128u      /* fused encode+decoder entries (miso_sdf_*): evaluate the decoder's products as exact fp32
                                     FMA chains on v_mfma_f32_32x32x2_f32 (rounds 1-5).  Default since round 6: every fp32 operand
                                     as three bf16 pieces, six piece products on v_mfma_f32_32x32x16_bf16 with fp32
                                     accumulation -- the same 2^-24 class of error against float64 (tests/test_split_precision.py),
                                     2.7 x fewer matrix clocks.  Results of the two forms differ in the last bits. */
#define MISO_F_ATLAS_NO_BOUND 512u /* miso_atlas_sdf_fwd: skip the coords_in_bound test -- ONE submap with an identity pose row is
                                     then queried exactly as GridNet.forward queries it (zeros padding decides what a point
                                     outside the grid sees; the level mask of the plan's grid is honoured): save_mesh(submap,
                                     submap.bound, ...) on a lattice generated in the kernel */
#define MISO_F_FULL_TRIPS 256u     /* miso_sdf_train on a small unbinned batch (<= 65 536 samples): 64-point trips per wavefront as
                                     for large batches, instead of the default 32-point trips (same arithmetic per point; the
                                     32-point form halves a wavefront's chain of matrix instructions where the batch is one
                                     chunk per wavefront anyway).  For tests and A/B runs. */
#define MISO_F_GRAD_SDF_SORTED 16u  /* miso_sdf_bwd_sorted: grad_sdf is in the binned order (what
                                       miso_sdf_fwd_sorted_loss writes), not the caller's */
#define MISO_F_GRAD_ZEROED 32u      /* with MISO_F_GRAD_OVERWRITE: the levels miso_sdf_bwd_sorted ADDS to with atomics
                                       (miso_sdf_bwd_scattered_levels) are zero on entry -- e.g. cleared by the Adam
                                       launch that consumed them (zero_grad) -- so the library skips its fill */
#define MISO_F_GRAD_OVERWRITE 8u    /* miso_sdf_bwd_sorted: level[l].grad = sum instead of += (the library
                                       clears what it still scatters; the caller never zero-fills) */

/* One feature-grid level: logical tensor (1,C,Z,Y,X) (grid_modules.py:54-57)
 * with arbitrary element strides, so both the reference's NCDHW layout and the
 * channels-last layout (sC==1) preferred on MI355X are accepted. */
typedef struct {
  const float* data; /* read-only values (may be NULL where only `grad` is used) */
  float* grad;       /* scatter-add target with the SAME strides, or NULL        */
  int32_t C, Z, Y, X;
  int64_t sC, sZ, sY, sX;
  /* NULL, or one byte per MISO_ADAM_CHUNK (64) consecutive floats of the DENSE storage that starts at `grad`:
   * every kernel of this library that adds or stores a non-zero into grad[off] also sets grad_touched[off / 64]
   * = 1 (nothing here ever clears a byte).  miso_adam_touched then steps a 0.6 G-byte level from its 2.3 M flag
   * bytes instead of reading the whole gradient to find the few chunks a small batch wrote. */
  uint8_t* grad_touched;
} miso_level_t;

typedef struct {
  int32_t n_levels;
  uint32_t ignore_mask;                /* bit l set => level l contributes zeros (utils.py:160-163) */
  float bound_min[3], bound_max[3];    /* x,y,z metres (unused with COORDS_NORMALIZED) */
  uint32_t flags;
  miso_level_t level[MISO_MAX_LEVELS];
} miso_grid_t;

/* Decoder MLP = MLPNet(input_dim, output_dim, hidden_dim, hidden_layers, bias)
 * (modules.py:11-21): n_linear = hidden_layers + 2 nn.Linear weights (out,in)
 * row-major, optional biases. */
typedef struct {
  int32_t in_dim, hidden_dim, out_dim, n_linear;
  const float* weight[MISO_MAX_LINEAR];
  const float* bias[MISO_MAX_LINEAR]; /* NULL = no bias */
} miso_mlp_t;

const char* miso_version(void);
const char* miso_error_string(int code);

/* feats[n, F] (row stride `ld_out` floats, F = sum of C over levels) =
 * concat_l trilinear(level l, x[n]) ; x is (N,3) row-major metres. */
int miso_encode_fwd(const miso_grid_t* grid, const float* x, int64_t n,
                    float* feats, int64_t ld_out, void* stream);

/* First backward.  grad_feats (N,F) row stride ld_g.  For each level with
 * level[l].grad != NULL: grad += scatter (caller zero-initialises).  grad_x (N,3)
 * may be NULL; when non-NULL level[l].data must be valid. */
int miso_encode_bwd(const miso_grid_t* grid, const float* x, int64_t n,
                    const float* grad_feats, int64_t ld_g, float* grad_x, void* stream);

/* Second backward (double backward of the encode).  gg_grid->level[l].data is
 * the cotangent of grad_grid (NULL per level = zero), gg_x (N,3) the cotangent of
 * grad_x (NULL = zero).  Outputs: gg_out (N,F) written; grid->level[l].grad +=
 * scatter where non-NULL; g_x (N,3) written where non-NULL. */
int miso_encode_bwd2(const miso_grid_t* grid, const miso_grid_t* gg_grid, const float* x,
                     int64_t n, const float* grad_feats, int64_t ld_g, const float* gg_x,
                     float* gg_out, int64_t ld_gg, float* g_x, void* stream);

/* --- fused encode + decoder (frozen decoder, out_dim == 1) ---------------- */
/* Size in floats of the packed-weight buffer for this decoder, or 0 if the fused
 * kernels do not cover the shape (then use miso_encode_* + a library GEMM). */
int64_t miso_mlp_packed_floats(const miso_mlp_t* mlp);
int miso_mlp_pack(const miso_mlp_t* mlp, float* packed, void* stream);
/* 1 if (grid, mlp) is covered by miso_sdf_fwd/bwd, else 0 */
int miso_sdf_supported(const miso_grid_t* grid, const miso_mlp_t* mlp);
/* Dynamic LDS (bytes) of the one-launch training kernel (miso_sdf_train / miso_sdf_train_sorted) for this (levels, C,
 * hidden width, hidden layers) shape -- scattering != 0: the form that scatters some level from the kernel (cell records
 * beside the d-feat tiles); 0: every level deferred to the pull.  0 = shape not covered.  A caller routes a shape whose
 * figure exceeds the device's LDS per workgroup (160 KB on gfx950) to the two-launch path instead of failing in the launch. */
int64_t miso_sdf_train_lds_bytes(const miso_grid_t* grid, const miso_mlp_t* mlp, int32_t scattering);

/* uint32 words of ReLU sign bits per point: (n_linear-1) * hidden_dim/32 */
int64_t miso_sdf_mask_words(const miso_mlp_t* mlp);

/* sdf[n] = MLP(encode(x[n])).  relu_mask: NULL (inference) or
 * ceil(n/64)*64*miso_sdf_mask_words(mlp) uint32 words that miso_sdf_bwd consumes. */
int miso_sdf_fwd(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                 const float* x, int64_t n, float* sdf, uint32_t* relu_mask, void* stream);

/* grad_sdf (N) -> scatter into level[l].grad (non-NULL levels) and/or grad_x. */
int miso_sdf_bwd(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                 const float* x, int64_t n, const float* grad_sdf, const uint32_t* relu_mask,
                 float* grad_x, void* stream);

/* miso_sdf_bwd that also hands out the d-feat rows (N, F) it forms on the way (row n = d loss / d feats of point n, F =
 * sum of C): the quantity the SECOND backward differentiates when the reference runs torch.autograd.grad(sdf, x,
 * create_graph=True) for its eikonal / smoothness terms (grid_opt/loss_isdf.py:96-152, :367-377; loss.py:638-665) --
 * d sdf / d x = J_E(x; G)^T rows, so miso_encode_bwd2 with grad_feats = these rows is the whole double backward (a ReLU
 * decoder is piecewise linear).  dfeat_rows NULL: exactly miso_sdf_bwd.  Caller-order points only. */
int miso_sdf_bwd_rows(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                      const float* grad_sdf, const uint32_t* relu_mask, float* grad_x, float* dfeat_rows, void* stream);

/* --- spatially binned batches ----------------------------------------------
 * Counting sort of a point batch by coarse tile (tiles_per_axis^3 tiles over the
 * bound).  No reference counterpart: the reference gathers every level with
 * independent random accesses (grid_modules.py:86-94); binning is what lets the
 * gathers of neighbouring lanes share cache lines and the backward form the grid
 * gradient owner-computes, without atomics (miso_grad_pull).  Results of the
 * *_sorted calls are identical to the unsorted ones up to fp32 summation order;
 * sdf, grad_sdf (unless MISO_F_GRAD_SDF_SORTED) and grad_x stay in the caller's
 * (original) point order. */
/* tiles_per_axis (here and wherever a binning is named): a plain count 1..16 (cubic binning), or per-axis counts of 1..32
 * packed with MISO_TILES_XYZ -- e.g. (25, 13, 25) puts at most 8 vertices of a 200 x 100 x 200 level (ScanNet's fine
 * level, configs/rgbd/scannet.yaml:23-24) on a tile and axis, which is what the owner-computes gradient can own.  A packed
 * binning is served by the matrix-core pull only: miso_grad_pull_levels reports which levels it takes (at least 2/3 as many
 * vertices as tiles on every axis, 4 or 8 channels); second-order gradients (miso_grad_pull with gg_x) need a plain count. */
#define MISO_TILES_XYZ(tx, ty, tz) ((int32_t)((tx) | ((ty) << 8) | ((tz) << 16)))
typedef struct {
  int32_t tiles_per_axis;      /* 1..16, or MISO_TILES_XYZ(tx, ty, tz) */
  const float* x_sorted;       /* (N,3) points grouped by tile                   */
  const float* xn_sorted;      /* (N,4) the same normalised to [-1,1] as {x, y, z, i}, 16-B aligned; NULL = absent.  i = the
                                  point's ORIGINAL index as a 32-bit integer bit pattern (what perm[] holds) */
  const int32_t* perm;         /* (N) sorted position -> original index.  NULL is accepted by miso_sdf_train_sorted only
                                  (it takes the index from xn_sorted[p].w): a mapping step that runs nothing else on the
                                  batch saves the sort one scattered 4-byte store per point */
  const int32_t* tile_offsets; /* (number of tiles + 1) start of every tile in x_sorted; tile (tx,ty,tz) = (tz Ty + ty) Tx + tx */
  /* Optional scratch of the owner-computes gradient (miso_grad_pull, miso_sdf_bwd_sorted):
   * miso_pull_queue_ints(n) int32, ZEROED ONCE by the caller when allocated (the library leaves it
   * zeroed after every call).  With it, tiles that hold far more points than the average are cut
   * into slices that run on separate wavefronts -- a batch hugging surfaces instead of filling the
   * bound would otherwise be limited by its heaviest tile.  NULL / 0: tiles are never cut. */
  int32_t* pull_queue;
  int64_t pull_queue_ints;
} miso_sorted_t;
int64_t miso_pull_queue_ints(int64_t n);

int64_t miso_sort_workspace_bytes(int64_t n, int32_t tiles_per_axis);
int miso_sort_points(const miso_grid_t* grid, const float* x, int64_t n, int32_t tiles_per_axis,
                     void* workspace, float* x_sorted /* may be NULL */, float* xn_sorted /* may be NULL */,
                     int32_t* perm /* may be NULL when xn_sorted is given */, int32_t* tile_offsets, void* stream);
/* miso_encode_fwd over a binned batch: the gathers of neighbouring lanes share cache lines
 * (105 -> 39 us at 262144 points); feats rows are written in the caller's order. */
int miso_encode_fwd_sorted(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, float* feats,
                           int64_t ld_out, void* stream);
int miso_sdf_fwd_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const miso_sorted_t* sorted, int64_t n, float* sdf, uint32_t* relu_mask,
                        void* stream);
/* miso_sdf_fwd_sorted with the mapping loss folded in (the trainer step of
 * grid_opt/trainer.py:196-228 with MisoLossMapping): right after a point's SDF the kernel
 * evaluates the loss terms of miso_mapping_loss for it and writes d loss / d sdf to
 * grad_sdf_sorted[p] (binned order: pass it to miso_sdf_bwd_sorted with
 * MISO_F_GRAD_SDF_SORTED).
 * loss_inputs (N,4), 16-B aligned, caller order: {target, valid, sign, weight} per point
 * (valid / sign compare against 1.0 as in miso_mapping_loss; use valid = weight = 1, sign = 0
 * for absent masks).
 * loss_slots (MISO_LOSS_SLOTS, 2) floats, device: per-workgroup partial sums, ALL written by the
 * launch (nothing to zero, no atomics: same-address atomics serialise at ~13 ns each and every
 * workgroup ends at about the same time).  The loss is the column sum: [.,0] weight_sdf * sdf
 * term, [.,1] weight_fs * free-space term.
 * sdf (caller order) may be NULL.
 * n_live (one int32 on the DEVICE, or NULL): number of live rows when the batch is a fixed-capacity
 * buffer whose other rows are neutral padding (valid = sign = weight = 0, as miso_sample_rays
 * leaves them).  The reference's means run over the rows of the batch (loss.py:627-635, :698-700),
 * so with padding they must divide by the live count, which only the device knows; NULL = n. */
#define MISO_LOSS_SLOTS 512
int miso_sdf_fwd_sorted_loss(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                             const miso_sorted_t* sorted, int64_t n, int loss_type, float weight_sdf,
                             float weight_fs, float trunc_dist, const float* loss_inputs, float* sdf,
                             uint32_t* relu_mask, float* grad_sdf_sorted, float* loss_slots,
                             const int32_t* n_live, void* stream);
/* The whole binned training step behind one call: forward + mapping loss + decoder backward in ONE launch
 * (sdf_train_kernel: the ReLU sign bits and d loss / d sdf never leave the registers), then the owner-computes pull /
 * matrix-core push of every level's gradient from the d-feat rows left in `workspace`
 * (miso_sdf_bwd_workspace_floats floats, 16-byte aligned).  Same results as miso_sdf_fwd_sorted_loss followed by
 * miso_sdf_bwd_sorted with MISO_F_GRAD_SDF_SORTED (same arithmetic, same order of operations per point).  Requires a
 * frozen decoder and at least one level with a gradient (MISO_E_UNSUPPORTED otherwise).  A level the pull cannot form
 * (bricks of more than 8 vertices per tile and axis: not in miso_grad_pull_levels) is scattered with float atomics from
 * the same launch, as miso_sdf_bwd_sorted scatters it.  grid->flags: MISO_F_GRAD_OVERWRITE / _ZEROED / MISO_F_CROWDED as
 * for miso_sdf_bwd_sorted.  sdf (caller order) may be NULL. */
int miso_sdf_train_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                          const miso_sorted_t* sorted, int64_t n, int loss_type, float weight_sdf, float weight_fs,
                          float trunc_dist, const float* loss_inputs, float* sdf, float* loss_slots,
                          const int32_t* n_live, float* workspace, void* stream);
/* The same step for an unbinned (small) batch, x (N,3) and loss_inputs (N,4) in the caller's order: forward + mapping
 * loss + decoder backward + the atomic scatter of every level's gradient in ONE launch (what miso_sdf_fwd_loss +
 * miso_sdf_bwd do in two).  The gradients are ADDED to `grad` (clear them first, or let miso_adam_active(zero_grad)). */
int miso_sdf_train(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                   int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* loss_inputs,
                   float* sdf /* may be NULL */, float* loss_slots, void* stream);
/* The forward half for an unbinned (small) batch: x (N,3) and loss_inputs in the caller's order, grad_sdf (N) in that order too
 * (feed it to miso_sdf_bwd).  Replaces miso_sdf_fwd + miso_mapping_loss_rows and the clear of their two sums. */
int miso_sdf_fwd_loss(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n,
                      int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* loss_inputs,
                      float* sdf /* may be NULL */, uint32_t* relu_mask, float* grad_sdf, float* loss_slots,
                      void* stream);
/* The owner-computes gradient on its own: rows of d loss / d feats (row pitch ld_d floats,
 * a multiple of 4; 16-B aligned base) -> level[l].grad for every level with a non-NULL grad,
 * written (MISO_F_GRAD_OVERWRITE) or accumulated, without atomics.  Rows are in the binned
 * order (rows_in_caller_order = 0: what miso_sdf_bwd_sorted leaves in its workspace) or in the
 * caller's order (1: row perm[p] belongs to sorted point p -- the grad_output of
 * F.grid_sample as autograd hands it over).  Every level with a grad must be pullable:
 * miso_grad_pull_levels returns that set as a bit mask (<= 8 vertices per tile and axis,
 * default sampling flags, channels-last C in {4,8}, at most 4 levels); the others go through
 * miso_encode_bwd.  Replaces the grad_input half of aten::grid_sampler_3d_backward
 * (third_party/cuda_gridsample_grad2/cuda_gridsample.py:102-105). */
uint32_t miso_grad_pull_levels(const miso_grid_t* grid, int32_t tiles_per_axis);
/* 1 when the pull of those levels over n points with d-feat rows of pitch ld_d runs as grad_pull_mc_kernel (the sums on the
 * fp32 matrix cores, grad_pull_mc.hip), 0 when the vector kernels of grad_pull.hip take it (second-order weights are always
 * theirs): what a profile of the call will show -- bench.py names its dominant kernel by this, not by assumption. */
int miso_grad_pull_on_matrix_cores(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n, int64_t ld_d);
int miso_grad_pull(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, const float* dfeat,
                   int64_t ld_d, int32_t rows_in_caller_order, void* stream);
/* miso_encode_bwd over a binned batch: grad_x (and any level the pull cannot own) with
 * tile-ordered gathers; grad_feats / grad_x rows stay in the caller's order. */
int miso_encode_bwd_sorted(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n,
                           const float* grad_feats, int64_t ld_g, float* grad_x, void* stream);
/* miso_encode_bwd2 over a binned batch (tile-ordered gathers); all per-point rows (grad_feats,
 * gg_x, gg_out, g_x) stay in the caller's order. */
int miso_encode_bwd2_sorted(const miso_grid_t* grid, const miso_grid_t* gg_grid, const miso_sorted_t* sorted,
                            int64_t n, const float* grad_feats, int64_t ld_g, const float* gg_x, float* gg_out,
                            int64_t ld_gg, float* g_x, void* stream);
/* The same for the SECOND backward: the g_input output of gridsample_grad2.grad2_3d
 * (gridsample_cuda.cu:462-481), level[l].grad (+)= sum_i (sum_a gg_x[i,a] * d w_corner / d x_a) *
 * grad_feats[i, channels of l] -- the scatter half of miso_encode_bwd2, which is then called with
 * grad == NULL for these levels.  grad_feats (N, ld_g) and gg_x (N,3) in the caller's order. */
int miso_grad_pull_dx(const miso_grid_t* grid, const miso_sorted_t* sorted, int64_t n, const float* grad_feats,
                      int64_t ld_g, const float* gg_x, void* stream);

/* workspace (16-B aligned, miso_sdf_bwd_workspace_floats floats, may be NULL): with it and
 * sorted->xn_sorted the grid gradient is formed owner-computes (grad_pull.hip): every tile
 * gathers the contributions to the vertices it owns and writes them once, without atomics.
 * Levels the pull cannot own (more than 8 vertices per tile and axis, non-default sampling
 * flags) keep the atomic scatter. */
/* Levels (bit l) whose gradient miso_sdf_bwd_sorted forms by the matrix-core push instead of the pull (grad_pull.hip:
 * per run of tile-sorted samples a (samples x 25)^T (samples x 5 C) product on v_mfma_f32_32x32x2_f32, added to the
 * zero-filled level with atomics): a batch that averages >= 100 samples per tile ($MISO_DENSE_MIN), 4 or 8 channels,
 * and every tile's samples within 5 vertices per axis (a tile owns <= 3).  n = batch size.  Informational: the entry
 * point decides by itself. */
uint32_t miso_sdf_bwd_push_levels(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n);
/* Levels (bit l, among those with a grad pointer) that miso_sdf_bwd_sorted forms by ADDING with atomics -- the decoder
 * pass's scatter for levels the pull cannot own, and the push -- as opposed to the pull's plain stores.  With
 * MISO_F_GRAD_OVERWRITE these are the levels the call zero-fills first, unless MISO_F_GRAD_ZEROED says they are
 * zero already. */
uint32_t miso_sdf_bwd_scattered_levels(const miso_grid_t* grid, int32_t tiles_per_axis, int64_t n);
int64_t miso_sdf_bwd_workspace_floats(const miso_grid_t* grid, int64_t n);
int miso_sdf_bwd_sorted(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed,
                        const miso_sorted_t* sorted, int64_t n, const float* grad_sdf,
                        const uint32_t* relu_mask, float* grad_x, float* workspace, void* stream);

/* --- latent alignment residual of a submap pair (pose-Jacobian path) --------
 * pairwise_loss_latent (grid_opt/align/miso.py:116-211) for the L2 / L1 variants.
 * dst_grid: the destination submap's levels 0..level (data only) with its bound.
 * pose (DEVICE pointer, 24 floats): R_src[9] t_src[3] R_dst[9] t_dst[3], row-major, the
 *   updated submap poses of grid_opt/models/grid_atlas.py:250-268.
 * coords_src (N,3): the source submap's cached voxel centres; feats_src (N, ld_feats >= F):
 *   its features there.  loss_type 1 = L1 (row-wise 2-norm), 2 = L2.
 * out (24 DOUBLES, device, 8-byte aligned): the lanes' fp32 terms are summed in fp64 (wave, workgroup and the atomics
 *   across workgroups) -- the rotation cotangents are differences of large sums, and an fp32 fan-in over ~1e4
 *   workgroups was less accurate than the reference's own fp32 tensor sum.
 *   [0] sum_i term_i over in-bound vertices, [1] their count,
 *   [2..4] sum g_i (g_i = d term_i / d q_i, q_i the point in the dst frame),
 *   [5..13] sum (w_i - t_dst) g_i^T, [14..22] sum (R_dst g_i) p_i^T  (row-major 3x3).
 * The caller forms loss = weight * out[0] / (count * F) (L2) or / count (L1) and the pose
 * cotangents dL/dt_dst = -R_dst sum g, dL/dR_dst = sum (w - t_dst) g^T, dL/dt_src = sum R_dst g,
 * dL/dR_src = sum (R_dst g) p^T, scaled alike. */
int miso_pair_latent(const miso_grid_t* dst_grid, const float* pose, const float* coords_src,
                     const float* feats_src, int64_t ld_feats, int64_t n, int loss_type, double* out,
                     void* stream);

/* --- submap overlap test ----------------------------------------------------
 * GridAtlas.check_submap_intersection (grid_opt/models/grid_atlas.py:405-420): count_out[0]
 * (device float) = number of coords_src (N,3; the source submap's finest-level voxel centres)
 * that fall inside [bound_min, bound_max] (host floats[3], inclusive) of the destination after
 * src -> world -> dst.  pose: as miso_pair_latent (24 floats, DEVICE).  No host sync: the
 * caller compares count / N with its threshold on the device or reads it back. */
int miso_overlap_count(const float* pose, const float* coords_src, int64_t n, const float* bound_min,
                       const float* bound_max, float* count_out, void* stream);

/* --- fused pose-Adam iteration of latent submap alignment ----------------------
 * generic_align_multiple_submaps (grid_opt/align/base.py:89-163) with pairwise_loss_latent (L2 / L1,
 * grid_opt/align/miso.py:116-211) as the pair loss: Adam over the pose corrections (dr_s, dt_s) of submaps
 * 1..S-1, submap 0 fixed.  One iteration = miso_align_iteration_a (poses from the corrections, overlap gate and
 * latent residual of EVERY pair in one launch each, pose cotangents pulled back through R0 Exp(dr)) followed by
 * miso_align_iteration_b (trust-region regulariser, NaN guard, Adam, relative-change early stop, bookkeeping).
 * Between the two the caller may all-reduce `flat` (7S + 1 floats: d loss / d (dr_s, dt_s) for every submap, the
 * summed pair loss, then per submap the number of its pairs that passed the overlap gate) over ranks that each hold a
 * share of the pair list (miso_amd/dist.py); nothing is read back by the host inside the loop.  A submap none of
 * whose pairs passed the gate has no gradient in the reference (its pair losses are not in the loss dict, base.py:134)
 * and torch.optim.Adam leaves such a parameter alone: so does _b -- value, moments and the submap's own step count.
 *
 * miso_align_pair_t   one (src, dst) pair: the destination's levels 0..level (data only) with its bound, the
 *                     source's cached voxel centres and its features there (as miso_pair_latent), and the
 *                     source's finest-level voxel centres for the overlap gate of base.py:134 /
 *                     GridAtlas.check_submap_intersection (NULL = pair always on).
 * miso_align_plan_build  converts n_pairs descriptors into the device layout the kernels read:
 *                     miso_align_plan_bytes(n_pairs) bytes of HOST memory, which the caller copies to the device
 *                     and passes as cfg->plan; fills cfg->vec4 / max_n / max_gate_n.
 * state               miso_align_state_layout(...) floats on the device, ZEROED by the caller before the first
 *                     iteration; offsets[12] (in floats) = {params (S,6: dr, dt), pose (S,12: R, t), out (P,24) DOUBLES = 48 P floats,
 *                     overlap counts (P), pair losses (P, weighted, gated), flat (7S+2), adam exp_avg (S,6),
 *                     exp_avg_sq (S,6), ctrl (8 x int32: iterations that stepped, stopped flag, iterations run,
 *                     NaN-skipped iterations), ring, ring row length, per-submap Adam step counts (S x int32)}.  Ring row k (ring_iters rows): {total loss, relative
 *                     pose change (inf at k = 0)} of iteration k [+ (S,4,4) poses BEFORE its step when save_poses:
 *                     iteration_results_helper, base.py:29-39].  The caller writes the initial corrections into
 *                     `params` and reads the final ones from there.  Behind the ring the library keeps P int32 of its
 *                     own (the order in which the next pair stage takes the pairs up: those with most in-bound
 *                     vertices first -- scheduling only, zero = the list's order).
 * Once the relative change falls below rel_change_thresh (base.py:157-158) the stopped flag is set and further
 * iterations change nothing. */
typedef struct {
  miso_grid_t dst_grid;
  const float* coords_src;  /* (n,3) */
  const float* feats_src;   /* (n, ld_feats) */
  int64_t ld_feats, n;
  const float* gate_coords; /* (gate_n,3) or NULL */
  int64_t gate_n;
  /* The gate vertices as a lattice instead (GridAtlas.check_submap_intersection tests ALL voxel centres of the source's
   * finest level, FeatureGrid.vertex_positions: a meshgrid of three per-axis tables): gate_axis[a] = the gate_dims[a]
   * coordinates along axis a (x, y, z), vertex index (k ny + j) nx + i.  When gate_axis[0] != NULL the kernel forms the
   * positions from the tables (same values, same arithmetic, same count) and gate_coords is not read: the gate of a
   * 4 M-vertex level costs no HBM traffic instead of 48 MB per pair and iteration.  gate_n must equal nx ny nz. */
  const float* gate_axis[3];
  int32_t gate_dims[3];
  int32_t src, dst;         /* submap indices */
  /* Optional: miso_align_src_boxes(coords_src, n, ...) -- one axis-aligned box per run of MISO_ALIGN_BOX_VERTS
   * consecutive source vertices.  With it the residual kernel maps a run's box into the destination frame first and
   * skips the run unread when it cannot touch the destination bound (conservatively: a skipped run holds no in-bound
   * vertex, so every sum is unchanged) -- the reference does the same per PAIR before any work
   * (check_submap_intersection, grid_opt/align/base.py:132-135); here per 64 vertices, every iteration, at the
   * current poses.  NULL: every vertex is read and tested. */
  const float* src_boxes;
} miso_align_pair_t;
#define MISO_ALIGN_BOX_VERTS 64
/* boxes: (ceil(n / MISO_ALIGN_BOX_VERTS), 6) floats {min x, y, z, max x, y, z} of coords[64 r .. 64 r + 63] (NaN
 * coordinates are ignored by min / max; they fail the bound test anyway).  Poses do not enter: build once per source list. */
int miso_align_src_boxes(const float* coords, int64_t n, float* boxes, void* stream);

typedef struct {
  int32_t n_submaps, n_pairs;
  int32_t loss_type;         /* 1 = L1 (row-wise 2-norm), 2 = L2 */
  int32_t ring_iters, save_poses;
  int32_t vec4;              /* set by miso_align_plan_build */
  int64_t max_n, max_gate_n, max_gate_rows; /* set by miso_align_plan_build: longest source list, longest point-list gate,
                                             * most lattice rows (ny nz) of a lattice gate */
  float align_weight, overlap_thresh;
  float reg_weight, reg_thresh_rad, reg_thresh_m; /* grid_atlas_pose_trust_region_loss, base.py:20-27; 0 = off */
  float rel_change_thresh;
  double lr, beta1, beta2, eps;
  const float* R0;           /* (S,9) base rotations, device */
  const float* t0;           /* (S,3) base translations, device */
  const void* plan;          /* device copy of the plan blob */
  float* state;              /* device */
  /* miso_align_iteration_a only.  Non-zero: the previous call on this state was miso_align_iteration_b, which leaves the
   * poses of the next iteration (R0 Exp(dr), t0 + dt of the stepped corrections), their snapshot in the ring and cleared
   * pair accumulators behind -- iteration_a then skips its prologue launch (four launches per iteration instead of five).
   * Zero: iteration_a forms them itself (first iteration; after the caller wrote the corrections; a repeated
   * iteration_a). */
  int32_t poses_ready;
} miso_align_t;

int64_t miso_align_plan_bytes(int32_t n_pairs);
int miso_align_plan_build(const miso_align_pair_t* pairs, miso_align_t* cfg, void* plan_host);
int64_t miso_align_state_layout(int32_t n_submaps, int32_t n_pairs, int32_t ring_iters, int32_t save_poses,
                                int64_t* offsets /* [11] or NULL */);
int miso_align_iteration_a(const miso_align_t* cfg, void* stream);
int miso_align_iteration_b(const miso_align_t* cfg, void* stream);

/* --- tracker: Gauss-Newton normal equations ------------------------------
 * coords_frame (N,3): samples in the keyframe frame; R_frame (9 floats, DEVICE, row-major):
 * rotation keyframe -> submap; grad_sdf_x (N,3): d sdf / d x in the submap frame (what
 * miso_sdf_bwd returns as grad_x for grad_sdf = 1); sdf / target (N).
 * loss_type 2 = L2 (w = 1), 3 = Geman-McClure (w = c / (c + r^2)^2, c = gm_scale).
 * out (32 floats, device): [0,21) upper triangle of H = sum w J^T J, row-major (H00..H05,
 * H11..H15, ..., H55), [21,27) g = sum w J^T r, [27] sum w r^2, [28] N; J = [c^T R, grad^T],
 * c = (R x) x grad.  The caller adds lambda I and solves the 6x6 system. */
int miso_lm_normal_eq(const float* coords_frame, const float* R_frame, const float* grad_sdf_x,
                      const float* sdf, const float* target, int64_t n, int loss_type, float gm_scale,
                      float* out, void* stream);

/* --- one whole Levenberg-Marquardt step of the keyframe tracker (Tracker.lm_step, grid_opt/slam/tracker.py:148-212)
 * R = R_base so3_exp_map(rot_correction), t = t_base + trans_correction (GridNet.updated_kf_pose); samples into the
 * submap frame; SDF and its spatial gradient by the fused forward / coordinate backward; J, H = J^T W J + lm_lambda I,
 * g = J^T W r as in miso_lm_normal_eq; delta = solve(H, -g) (LU with partial pivoting, fp32); the two corrections
 * += delta IN PLACE.  What the reference settles with host round trips is counted on the device instead and handed
 * back in info (8 floats, device): {|delta_R| rad, |delta_t|, |g|, rows inside the grid bound, rows kept by the
 * truncation filter |target| < trunc_dist (trunc_dist < 0: all), kept rows whose frame id != keyframe_id, kept rows
 * whose validity != 1, 0} -- fov_overlap = info[3] / info[4]; the reference asserts info[5] == info[6] == 0.
 * Filtered rows take no part in anything, as if removed.  loss_type 2 = L2, 3 = Geman-McClure (gm_scale).
 * The grid needs `data` only (no grad pointers).  Scratch, all device: pose 12, coords_world 3N, sdf N, grad 3N floats,
 * ones N floats holding 1.0, relu_mask N * miso_sdf_mask_words words, sums 36 floats. */
typedef struct {
  const float* coords_frame;       /* (N,3) samples in the keyframe frame */
  const float* target;             /* measured SDF, element stride stride_target */
  const void* valid;               /* float (== 1 means valid) or, with valid_is_bool, one byte per row; or NULL */
  const int64_t* frame_ids;        /* or NULL */
  int64_t stride_target, stride_valid, stride_frame_ids;
  int32_t valid_is_bool;
  int64_t n;
  int64_t keyframe_id;
  float trunc_dist;
  const float* R_base;             /* 9 */
  const float* t_base;             /* 3 */
  float* rot_correction;           /* 3, updated */
  float* trans_correction;         /* 3, updated */
  int32_t loss_type;
  float gm_scale, lm_lambda;
  float* pose; float* coords_world; float* sdf; float* grad; const float* ones; uint32_t* relu_mask; float* sums;
  float* info;
  float* sanitized;                /* NULL, or scratch 5N floats: torch.nan_to_num (NaN -> 0, inf -> +-FLT_MAX) of
                                      coords_frame, target and a float `valid` is taken first -- get_batch's
                                      prepare_batch (grid_opt/utils/utils.py:487-493) folded into the call */
} miso_lm_track_t;
int miso_lm_track_step(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const miso_lm_track_t* args,
                       void* stream);

/* --- one Adam iteration of the tracker's pose refinement (Tracker.track_window with MisoLossTracking: tracker.py:95-118,
 * loss.py:517-586, one Trainer step incl. its NaN guard) without a host round trip.  `s` as for miso_lm_track_step
 * (inputs, base pose, the two corrections -- updated by the Adam step --, scratch; s.ones and s.loss_type are not used;
 * s.info[0] receives the loss).  resid = valid && |target| < trunc ? sdf - target : 0; loss = weight_sdf * mean over
 * ALL rows of resid^2 (2) | |resid| (1) | w resid^2 with w = c / (c + resid^2)^2 held constant (3).  The gradient
 * reaches the keyframe's six correction numbers through x_world = R x + t and R = R_base so3_exp_map(dr); Adam with the
 * step scalars of adam_table (miso_adam_scalars_table) and the moments in `state` (device: 12 floats m, v, then int32
 * {steps taken, steps skipped for a NaN loss, iterations}; zero it when a window starts -- the reference builds a new
 * optimizer per window).  loss_ring[i] = loss of iteration i. */
typedef struct {
  miso_lm_track_t s;
  int32_t loss_type;
  float weight_sdf, gm_scale;
  float* grad_pred;              /* scratch N */
  const float* adam_table;
  int32_t adam_table_len;
  float* state;
  float* loss_ring;
  int32_t ring_len;
} miso_track_adam_t;
int miso_track_adam_step(const miso_grid_t* grid, const miso_mlp_t* mlp, const float* packed, const miso_track_adam_t* args,
                         void* stream);

/* --- mapping loss (value + d/d pred) --------------------------------------
 * loss_type 1 = L1, 2 = L2.  pred/target (N); valid/sign/weight (N) or NULL
 * (= all valid / no free-space rows / unit weights).  Writes grad_pred (N) =
 * d(weight_sdf*sdf_term + weight_fs*fs_term)/d pred, optionally grad_pred_fs (N)
 * = the free-space share of it (NULL to skip), and loss_out[2] =
 * {weight_sdf*sdf_term, weight_fs*fs_term} (means over all N rows). */
int miso_mapping_loss(int loss_type, float weight_sdf, float weight_fs, float trunc_dist,
                      const float* pred, const float* target, const float* valid,
                      const float* sign, const float* weight, int64_t n, float* grad_pred,
                      float* grad_pred_fs, float* loss_out, void* stream);

/* --- dense Adam (torch.optim.Adam defaults: amsgrad=False, weight_decay=0) -
 * One launch over one dense tensor; param/grad/exp_avg/exp_avg_sq share a
 * layout.  Scalars are doubles like torch's Python-side hyper-parameters.
 * zero_grad != 0 also clears grad in the same pass (the next backward then
 * needs no memset). */
int miso_adam_dense(float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                    int64_t numel, double lr, double beta1, double beta2, double eps,
                    int32_t step /* 1-based */, int zero_grad, void* stream);

/* --- sample generation: posed depth frames -> SDF training rows ------------
 * Frames: depth (B,H,W) metres, 0 = no return (DepthFilter, grid_opt/utils/utils_data.py:34-47);
 * normals (B,H,W,3) or NULL -- only `isnan(normals[...,0])` is used, as a ray filter
 * (utils_sample.py:164); T_WC (B,4,4) row-major camera -> world; R_wk (B,3,3), t_wk (B,3): the
 * keyframe pose the samples are expressed in (the reference passes the same pose twice,
 * sdf_rgbd.py:209-210,442); frame_ids (B) int64 written to sample_frame_ids, NULL = 0..B-1.
 * Pin-hole intrinsics as ray_dirs_C(depth_type='z') (utils_sample.py:10-30). */
#define MISO_RAY_MAX_BINS 64
typedef struct {
  const float* depth;
  const float* normals;
  const float* T_WC;
  const float* R_wk;
  const float* t_wk;
  const int64_t* frame_ids;
  int32_t n_frames, H, W;
  float fx, fy, cx, cy;
} miso_ray_frames_t;

/* PosedSdfRgbd's sampling knobs (sdf_rgbd.py:33-40).  bin_edges: HOST array of n_strat+1 floats =
 * torch.linspace(0,1,n_strat+1) (utils_sample.py:214-216), NULL = computed by the library.
 * rays_per_frame: ray r belongs to frame r / rays_per_frame when pix_b is NULL
 * (repeat_interleave, utils_sample.py:136-137). */
typedef struct {
  float min_depth, dist_behind_surf, trunc_dist;
  int32_t n_strat, n_surf, rays_per_frame;
  const float* bin_edges;
} miso_ray_sampling_t;

int64_t miso_sample_rays_workspace_bytes(int64_t n_rays, int32_t n_frames);

/* One batch of rays.  pix_b (or NULL) / pix_h / pix_w (n_rays) int64: the sampled pixels
 * (sample_pixels, utils_sample.py:129-139).  u (n_rays, n_strat) uniforms in [0,1) and
 * g (n_rays, n_surf-1) near-surface depth offsets (the reference draws N(0, 0.1^2),
 * utils_sample.py:284-286), both indexed by the ray's position AFTER the depth/normal filter, as
 * the reference draws them for the surviving rays only.  S = n_surf + n_strat rows per ray, in the
 * order [surface, near..., stratified...]; rays with depth 0, a NaN normal or a NaN sample are
 * dropped and the remaining rays packed to the front, ray order kept.
 * Outputs, capacity n_rays*S rows each: coords_frame (rows,3) in the keyframe frame,
 * sample_frame_ids (rows) int64, aux (rows,4) = {sdf, valid, sign, weight} -- the row layout
 * miso_sdf_fwd_sorted_loss reads -- with valid = |sdf| < trunc_dist, sign = -1 / 0 / +1 beyond the
 * truncation band, weight = 1; optional pc_world (rows,3) and z_vals (rows) (NULL to skip).
 * Rows past the packed ones are neutral padding: aux = 0 (no loss, no gradient), coordinates and frame id
 * borrowed from a live sample so that a padded batch keeps the spatial spread of the live one.
 * counts (4 x int32, device) = {rays after the first filter, rays kept, live rows = kept * S, 0};
 * &counts[2] is what miso_sdf_fwd_sorted_loss takes as n_live. */
int miso_sample_rays(const miso_ray_frames_t* frames, const miso_ray_sampling_t* sampling, int64_t n_rays,
                     const int64_t* pix_b, const int64_t* pix_h, const int64_t* pix_w, const float* u,
                     const float* g, void* workspace, float* coords_frame, int64_t* sample_frame_ids,
                     float* aux, float* pc_world, float* z_vals, int32_t* counts, void* stream);

/* The same step without the work that changes nothing: an element whose gradient has been zero in
 * every step so far has exp_avg = exp_avg_sq = 0 and an update of exactly 0.  `active` holds one byte
 * per MISO_ADAM_CHUNK consecutive floats of the storage ((numel + CHUNK-1)/CHUNK bytes, zeroed by the
 * caller together with the moments): a chunk is stepped if any of its gradients is non-zero or if it
 * has been stepped before (then the byte is 1).  Bit-identical to miso_adam_dense; reads 4 B per
 * element plus 28 B per active element.  All four arrays must be 16-B aligned.  zero_grad clears the
 * gradient of the stepped chunks -- the others are zero already.
 * guard (device float, or NULL): the loss of the step.  If it is NaN the launch changes no parameter, moment or
 * flag (the reference's NaN guard, grid_opt/trainer.py:213-219: "Loss is nan! Skip backward step") and only
 * clears the gradients it was asked to clear -- the host can launch without reading the loss back first. */
#define MISO_ADAM_CHUNK 64
int miso_adam_active(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active,
                     int64_t numel, double lr, double beta1, double beta2, double eps,
                     int32_t step /* 1-based */, int zero_grad, const float* guard, void* stream);

/* miso_adam_active driven by the flags the scatter kernels leave (miso_level_t.grad_touched) instead of by reading
 * the gradient: a chunk is stepped if active[c] or touched[c]; touched[c] is cleared.  Same arithmetic, same results
 * PROVIDED every writer of `grad` since the last call maintained the flags (the kernels of this library do; a caller
 * that adds to the gradient by other means must use miso_adam_active).  `active` and `touched` must be 4-byte aligned
 * (four flag bytes are read as one word) and hold the values 0 / 1 only.  Traffic: 2 B per chunk + 28 B per stepped
 * element.  zero_grad and guard as above (a NaN guard leaves parameters, moments and `active` alone, still clears
 * the flags and, if asked, the touched gradients). */
int miso_adam_touched(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active,
                      uint8_t* touched, int64_t numel, double lr, double beta1, double beta2, double eps,
                      int32_t step /* 1-based */, int zero_grad, const float* guard, void* stream);

/* Adam inside a captured HIP graph.  The bias corrections change every step but the arguments of a captured launch do
 * not, so a captured step reads them from the device: `table` (device, table_len rows of 6 floats) holds the step
 * scalars of steps 1..table_len exactly as miso_adam_active computes them on the host -- filled by
 * miso_adam_scalars_table (a HOST function writing a HOST buffer; upload it) -- and `step` (device int32) the 1-based
 * count of the step being taken.  Steps past the table use its last row (choose table_len so that beta^table_len
 * has left fp32: the rows no longer change).  miso_adam_bump, once per step and in front of the level launches:
 * step += 1 unless *guard is NaN (the reference's NaN guard skips optimizer.step(), grid_opt/trainer.py:213-219, so
 * the count must not move).  miso_adam_step_dev = miso_adam_active (touched == NULL) / miso_adam_touched with the
 * scalars of table[*step - 1]: bit-identical to the launch-by-launch step. */
int miso_adam_scalars_table(double lr, double beta1, double beta2, double eps, int32_t first_step, int32_t count,
                            float* host_out /* count x 6 */);
int miso_adam_bump(int32_t* step, const float* guard /* or NULL */, void* stream);
/* total[0] = sum of n_floats floats (the loss slots of miso_sdf_fwd_sorted_loss / miso_sdf_fwd_loss: 2 * MISO_LOSS_SLOTS,
 * i.e. both terms of the mapping loss) and, with step != NULL, miso_adam_bump(step, total) in the same launch. */
int miso_loss_total_bump(const float* loss_slots, int32_t n_floats, float* total, int32_t* step /* or NULL */,
                         void* stream);
/* The same with the total ALSO handed to the host: slot s = step2[1] % ring_len of host_ring receives
 * host_ring[2 s] = total and then, behind a system-scope fence, ((int32_t*)host_ring)[2 s + 1] = step2[1] + 1 (the
 * 1-based number of this launch); step2[1] += 1.  step2: device int32[2] = {Adam step count, launches so far};
 * host_ring: PINNED HOST memory mapped into the device, 2 * ring_len words, zeroed once; ring_len a power of two.
 * The reference reads the loss on the host after every step to decide whether to skip it (grid_opt/trainer.py:213-219);
 * here the host polls the slot for its launch's number -- no copy and no event on the stream.  The launch counter
 * moves on NaN too, so a skipped step does not alias the next one's slot. */
int miso_loss_total_bump_host(const float* loss_slots, int32_t n_floats, float* total, int32_t* step2,
                              float* host_ring, int32_t ring_len, void* stream);
int miso_adam_step_dev(float* param, float* grad, float* exp_avg, float* exp_avg_sq, uint8_t* active,
                       uint8_t* touched /* or NULL */, int64_t numel, const float* table, int32_t table_len,
                       const int32_t* step, int zero_grad, const float* guard, void* stream);
/* miso_adam_step_dev for up to MISO_ADAM_MAX_TENSORS tensors in ONE launch, each bit-identical to its
 * own call: the levels of a grid as one optimizer.step() (grid_opt/trainer.py:217), without a launch's ramp and tail per
 * level.  `tensors` is a HOST array (copied into the kernel arguments). */
#define MISO_ADAM_MAX_TENSORS 8
typedef struct {
  float* param; float* grad; float* exp_avg; float* exp_avg_sq;   /* DEVICE, 16-byte aligned, numel floats each */
  uint8_t* active;                                                /* miso_adam_flag_bytes(numel) */
  uint8_t* touched;                                               /* NULL: stepped by reading the gradient (miso_adam_active);
                                                                     else the scatter kernels' flags, as miso_adam_touched */
  int64_t numel;
  int32_t zero_grad;
  int32_t reserved;
} miso_adam_tensor_t;
int miso_adam_step_dev_multi(const miso_adam_tensor_t* tensors, int32_t n_tensors, const float* table, int32_t table_len,
                             const int32_t* step, const float* guard, void* stream);
/* the same for miso_adam_active (the step's scalars from the host: lr .. eps, the 1-based step) */
int miso_adam_active_multi(const miso_adam_tensor_t* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                           double eps, int32_t step, const float* guard, void* stream);

/* --- per-keyframe rigid map of a sample batch ---------------------------------
 * y[i] = R[idx[i]] x[i] + t[idx[i]] (transpose = 0) or R[idx[i]]^T x[i] (+ t if given; transpose = 1: the cotangent
 * of x).  R (n_poses x 9, row-major 3x3), t (n_poses x 3) or NULL, idx (n) int64 clamped into [0, n_poses),
 * x / y (n x 3), all DEVICE.  Replaces the per-keyframe Python loops of the losses (grid_opt/loss.py:763-774,
 * grid_opt/loss_isdf.py:52-61, grid_opt/align/miso.py:44-53) around transform_points_to
 * (grid_opt/utils/utils_geometry.py:214-225). */
int miso_rigid_by_index(const float* R, const float* t, const int64_t* idx, const float* x, int64_t n,
                        int32_t n_poses, int transpose, float* y, void* stream);

/* The input side of one mapping step in one launch (MisoLossMapping.world_coords, grid_opt/loss.py:763-774, plus
 * the label layout of miso_sdf_fwd_sorted_loss / miso_mapping_loss_rows): k = table[clamp(frame_ids[i], 0,
 * table_len - 1)] clamped into [0, n_poses), coords_world[i] = R[k] coords_frame[i] + t[k] (operation order of
 * miso_rigid_by_index), loss_rows[i] = {target[i], valid[i], sign[i], weight[i]} (valid / sign / weight may be NULL:
 * 1 / 0 / 1).  table: int64 keyframe-id -> pose-index (device).  loss_rows 16-B aligned.
 * col_strides (HOST, 4 element strides of target / valid / sign / weight, or NULL = unit): the columns may be views
 * of a row-major label block; valid_is_bool: `valid` holds one byte per row (a torch.bool mask) instead of floats.
 * sanitize: torch.nan_to_num on every float read (NaN -> 0, +-inf -> +-FLT_MAX): the trainer's prepare_batch
 * (grid_opt/utils/utils.py:487-493) folded into the launch. */
int miso_mapping_batch(const float* R, const float* t, int32_t n_poses, const int64_t* table, int64_t table_len,
                       const int64_t* frame_ids, const float* coords_frame, const float* target, const void* valid,
                       const float* sign, const float* weight, const int64_t* col_strides, int valid_is_bool, int64_t n,
                       float* coords_world, float* loss_rows, int sanitize, void* stream);

/* miso_mapping_loss over interleaved label rows {target, valid, sign, weight} (N,4), 16-B aligned. */
int miso_mapping_loss_rows(int loss_type, float weight_sdf, float weight_fs, float trunc_dist, const float* pred,
                           const float* loss_rows, int64_t n, float* grad_pred, float* loss_out, void* stream);

/* utils.grid_pool_3d_avg (grid_opt/utils/utils.py:239-291; called by models/encoder.py to pool residual signals onto a
 * level's cells): pooled[(ix ny + iy) nz + iz][c] = mean of features[i][c] over the points i with cell index
 * clamp(trunc((coords[i] - bound_min) / cell_size), 0, size - 1) per axis, 0 for empty cells.  coords (N,3), features (N,d)
 * with row stride ld_features, pooled (nx ny nz, d) and counts (nx ny nz) int32 are written by the call (no zero-fill by
 * the caller); bound_min: three host floats. */
int miso_grid_pool_avg(const float* coords, const float* features, int64_t n, int32_t d, int64_t ld_features,
                       const float* bound_min, float cell_size, int32_t nx, int32_t ny, int32_t nz, float* pooled,
                       int32_t* counts, void* stream);

/* --- fused atlas query: GridAtlas.query_feature / GridAtlas.forward in one launch (round 6) ---------------------------
 * Replaces the per-submap loop of grid_opt/models/grid_atlas.py:374-399 (for each active submap: transfrom_points_from,
 * coords_in_bound, grid_interp_regular of every point, mask * feats and mask added to running (N,F) / (N,1) tensors; then
 * count == 0 -> 1, sum / count, submap 0's decoder on the mean) and, with a lattice, the point generation of
 * utils_sdf.extract_fields (grid_opt/utils/utils_sdf.py:69-86: linspace per axis, meshgrid 'ij', 16^3-point chunks).
 *   miso_atlas_plan_bytes / miso_atlas_plan_build  the submaps' grids (levels, strides, bound; n_levels and C equal across
 *       submaps, channels-last) as the kernel reads them, written to HOST memory `plan_host`; the caller copies the bytes
 *       to the device once per set of feature tensors.  ignore_mask is not applied (the reference passes ignore_level=None).
 *   miso_atlas_sdf_fwd  `plan`: the device copy.  `shape`: any one of the submaps' grids (host struct: C and the level
 *       count are read from it).  `poses`: device (n_submaps, 12) floats -- per submap R_submap_world row-major (= R_world_
 *       submap^T) then t_submap_world (= -R^T t), formed by the caller as transfrom_points_from forms them.  Points: `x`
 *       (N,3) world coordinates, or x == NULL and a lattice: point (i j k), index (i ny + j) nz + k, sits at (axis_x[i],
 *       axis_y[j], axis_z[k]) (device arrays; n == nx ny nz < 2^31).  Outputs: `sdf` (N) and / or `feats` (N, ld_feats)
 *       mean features (either may be NULL; mlp / packed are needed for sdf only).  flags: MISO_F_EXACT_F32,
 *       MISO_F_ATLAS_NO_BOUND. */
int64_t miso_atlas_plan_bytes(int32_t n_submaps);
int miso_atlas_plan_build(const miso_grid_t* grids, int32_t n_submaps, void* plan_host);
int miso_atlas_sdf_fwd(const void* plan, int32_t n_submaps, const miso_grid_t* shape, const float* poses,
                       const miso_mlp_t* mlp, const float* packed, const float* x, int64_t n, const float* axis_x,
                       const float* axis_y, const float* axis_z, int32_t nx, int32_t ny, int32_t nz, float* sdf,
                       float* feats, int64_t ld_feats, uint32_t flags, void* stream);

/* --- marching cubes on the dense SDF volume ------------------------------------
 * Replaces mcubes.marching_cubes(u, threshold) as called by extract_geometry
 * (grid_opt/utils/utils_sdf.py:89-101; PyMCubes is a third-party dependency of the reference) with the
 * volume left in HBM, read once, and no sort.  vol: (nx, ny, nz) fp32, z fastest -- the layout of
 * extract_fields' u[x, y, z].  A corner is "inside" where vol < iso.
 * A sample ROW is the nz samples of one (x, y), row id = x*ny + y; a CHUNK is 64 consecutive z of a row.
 * W = nx*ny*ceil(nz/64) (row, chunk) pairs.  Vertices are numbered by the key ((x*ny + y)*3 + axis)*nz + z of
 * the lattice edge they sit on ((x,y,z) = the edge's low sample, axis 0/1/2 = x/y/z); triangles are listed
 * cell by cell in x-major order, a cell's triangles in table order, counter-clockwise seen from the
 * vol < iso side.
 *   miso_mc_words           W; -1 for a bad / too large shape (>= 2^31 samples).
 *   miso_mc_workspace_bytes device scratch shared by the three launches (sign bitmap, vertex bitmap, two work
 *                           lists: 5 bits per sample); 8-byte aligned.
 *   miso_mc_classify        one sweep over the volume: fills the workspace and counts (4W + 2 int32, device):
 *                           [0, 3W) vertices per (row, axis, chunk), [3W, 4W) triangles per (row, chunk),
 *                           [4W] = chunks with triangles, [4W + 1] = chunks with vertices (the lengths of the
 *                           work lists the next two launches run over: the caller reads them back with the totals).
 *   miso_mc_emit            offsets (4W int64, device) = exclusive prefix sums of counts[0, 3W) and of
 *                           counts[3W, 4W) (the caller's two cumsums), same layout; n_tri_chunks = counts[4W].
 *                           faces (capacity_tris x 3 int64) = vertex indices.
 *   miso_mc_vertices        n_vert_chunks = counts[4W + 1].  verts (capacity_verts x 3 fp32), index coordinates
 *                           (x, y, z): low sample + (iso - u_a) / (u_b - u_a) along the edge's axis.
 *                           Triangles / vertices past the capacities are dropped.
 *   miso_mc_case_table      copies the 256 x 16 table (15 edge ids, -1 padded, + triangle count; edge id =
 *                           4*axis + a + 2*b, (a, b) = the edge's other two corner coordinates in increasing axis
 *                           order; case bit c = corner x + 2y + 4z) to HOST memory. */
int64_t miso_mc_words(int32_t nx, int32_t ny, int32_t nz);
int64_t miso_mc_workspace_bytes(int32_t nx, int32_t ny, int32_t nz);
int miso_mc_classify(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                     int32_t* counts, void* stream);
int miso_mc_emit(int32_t nx, int32_t ny, int32_t nz, void* workspace, const int64_t* offsets,
                 int32_t n_tri_chunks, int64_t capacity_tris, int64_t* faces, void* stream);
int miso_mc_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, void* workspace,
                     const int64_t* offsets, int32_t n_vert_chunks, int64_t capacity_verts, float* verts,
                     void* stream);
int miso_mc_case_table(int8_t* table_host);

#ifdef __cplusplus
}
#endif
#endif /* MISO_HIP_H */
