// Fused pose-Adam iteration of latent submap alignment.
//
// Reference: generic_align_multiple_submaps, grid_opt/align/base.py:89-163, driving pairwise_loss_latent
// (grid_opt/align/miso.py:116-211).  One iteration there is, per pair, two so3_exp_map chains, two affine maps, a
// nonzero compaction, two multi-level grid_sample calls and their autograd backward, then torch.optim.Adam on the
// 2(S-1) pose tensors -- ~200 tiny launches and several host syncs per pair.  Here an iteration is three launches
// whatever the number of pairs (five in rounds 2-3: the prologue now runs in the tail of the previous iteration's
// epilogue B -- miso_align_t.poses_ready -- and the overlap gate in the pair kernel's launch, pair_stage_kernel):
//
//   prologue    (dr_s, dt_s) -> R_s = R0_s Exp(dr_s), t_s = t0_s + dt_s for every submap (GridAtlas.updated_submap_pose,
//               grid_atlas.py:250-268, utils_geometry.apply_pose_correction :78-99); clears the accumulators; writes
//               the (S,4,4) pose snapshot of iteration_results_helper (base.py:29-39) into the ring
//   overlap     check_submap_intersection of every pair (grid_atlas.py:405-420), pair_latent.hip, grid.y = pair
//   pairs       the masked latent residual + 23 pose-cotangent sums of every pair, pair_latent.hip, grid.y = pair
//   epilogue A  per pair: normalise, nan_to_num (base.py:138-141), overlap gate; per submap: sum the cotangents of
//               its pairs, pull them back through R0 Exp(.) -> d loss / d (dr_s, dt_s); writes `flat` = 6S pose
//               gradients + the summed pair loss + S flags "a pair of this submap passed the overlap gate".  With the
//               pair list sharded over ranks (miso_amd/dist.py) `flat` is what the ONE all-reduce of the iteration sums.
//   epilogue B  trust-region regulariser (base.py:20-27), NaN guard (:147-151), Adam on the poses of submaps 1..S-1
//               (:104-111), relative pose change + early stop (:152-158), loss / change into the ring.  As
//               torch.optim.Adam does with a parameter whose .grad is None, a submap NONE of whose pairs passed the
//               gate (and no regulariser) is left alone -- value, moments and its own step count (bias corrections
//               are per submap): it neither drifts on old momentum nor ages.
//
// Everything the host reads (losses, changes, snapshots, iteration count) stays on the device until the loop ends.
#include <math.h>
#include <string.h>

#include "common.hpp"
#include "align.hpp"

namespace miso {

AlignLayout align_layout(int S, int P, int ring_iters, int save_poses) {
  AlignLayout L;
  int64_t o = 0;
  L.params = o; o += up4(6 * S);
  L.pose = o; o += up4(12 * S);
  L.out = o; o += up4(48 * (int64_t)P);      // (P,24) DOUBLES (pair_latent.hip reduces in fp64)
  L.cnt = o; o += up4(P);
  L.pair_loss = o; o += up4(P);
  L.flat = o; o += up4(7 * S + 2);      // 6S pose gradients, the loss sum, S "this submap has a gradient" flags
  L.adam_m = o; o += up4(6 * S);
  L.adam_v = o; o += up4(6 * S);
  L.adam_t = o; o += up4(S);            // int32 per submap: Adam steps THIS submap has taken
  L.ctrl = o; o += 8;
  L.ring_row = 2 + (save_poses ? 16 * S : 0);
  L.ring = o; o += up4((int64_t)ring_iters * L.ring_row);
  // int32 per launch slot: (pair + 1) the slot's workgroups work on, heaviest pair first (epilogue A, from the in-bound
  // counts of the iteration before); 0 = not set yet: slot y works on pair y.  Behind everything else: the offsets the
  // ABI publishes (miso_align_state_layout) do not move.
  L.order = o; o += up4(P);
  L.total = o;
  return L;
}

// What an iteration starts from: every submap's pose from its corrections, the (S,4,4) snapshot of
// iteration_results_helper for iteration `it`, cleared pair accumulators.  Run by align_prologue_kernel, and (round 4) by
// the tail of align_epilogue_b_kernel for the NEXT iteration, so that a loop pays four launches per iteration, not five.
// full_clear: also the reduction buffer `flat` and the pair losses (epilogue A overwrites both: only for tidiness of a
// state nobody has run yet).
__device__ __forceinline__ void align_prologue_body(const AlignK& k, int it, bool full_clear) {
  for (int s = threadIdx.x; s < k.S; s += blockDim.x) {
    const float* prm = k.state + k.L.params + 6 * s;
    const float w[3] = {prm[0], prm[1], prm[2]};
    float E[9], R[9];
    so3_exp(w, E);
    const float* R0 = k.R0 + 9 * s;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i * 3 + j] = (R0[i * 3] * E[j] + R0[i * 3 + 1] * E[3 + j]) + R0[i * 3 + 2] * E[6 + j];
    float t[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = k.t0[3 * s + i] + prm[3 + i];
    float* pose = k.state + k.L.pose + 12 * s;
#pragma unroll
    for (int i = 0; i < 9; ++i) pose[i] = R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) pose[9 + i] = t[i];
    if (k.save_poses && it < k.ring_iters) {      // utils_geometry.pose_matrix(R, t), row-major 4x4
      float* m = k.state + k.L.ring + (int64_t)it * k.L.ring_row + 2 + 16 * s;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        m[i * 4] = R[i * 3]; m[i * 4 + 1] = R[i * 3 + 1]; m[i * 4 + 2] = R[i * 3 + 2]; m[i * 4 + 3] = t[i];
      }
      m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
    }
  }
  // accumulators of the pair stage (out, cnt: added to with atomics) [and pair_loss, flat, which follow them]
  float* z = k.state + k.L.out;
  const int64_t nz = full_clear ? (k.L.flat + up4(7 * k.S + 2)) - k.L.out : k.L.pair_loss - k.L.out;
  for (int64_t i = threadIdx.x; i < nz; i += blockDim.x) z[i] = 0.0f;
}

__global__ __launch_bounds__(256) void align_prologue_kernel(AlignK k) {
  int32_t* ctrl = reinterpret_cast<int32_t*>(k.state + k.L.ctrl);
  if (ctrl[CTRL_STOPPED]) return;
  align_prologue_body(k, ctrl[CTRL_ITER], true);
}

// torch.relu keeps a NaN (fmaxf would drop it, and with it the reference's "loss is nan" skip)
__device__ __forceinline__ float relu_nan(float v) { return v > 0.0f ? v : (v != v ? v : 0.0f); }

__device__ __forceinline__ float nan_to_num_f(float v) {
  if (v != v) return 0.0f;
  if (isinf(v)) return v > 0.f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
  return v;
}

constexpr int EPI_A_THREADS = 256;
constexpr int EPI_A_PAIRS = 128;      // pairs staged in LDS per pass (S <= 64 => at most 2016 pairs: 16 passes)

__global__ __launch_bounds__(EPI_A_THREADS) void align_epilogue_a_kernel(AlignK k) {
  const int32_t* ctrl = reinterpret_cast<const int32_t*>(k.state + k.L.ctrl);
  if (ctrl[CTRL_STOPPED]) return;
  const double* out = reinterpret_cast<const double*>(k.state + k.L.out);
  const float* cnt = k.state + k.L.cnt;
  const float* pose = k.state + k.L.pose;
  float* flat = k.state + k.L.flat;
  float* pl = k.state + k.L.pair_loss;
  // Pass structure: (1) one thread per pair normalises it -- loss value, overlap gate, the scale of its cotangents --
  // into LDS (the descriptors live in global memory: read in parallel, not P times in a row by every submap's
  // thread); (2) FOUR threads per submap sum the cotangents of its pairs, each every fourth pair of the list in list
  // order, and are then added in a fixed tree ((0 + 2) + (1 + 3)): deterministic; thread 0 of the four pulls the sum
  // back through R0 Exp(dr).  (One thread per submap walked all P pairs in double precision: 14 us at cfg-4's 28
  // pairs, the longest kernel of a level-0 iteration after the pair stage.)  The pairs' 24 sums are staged in LDS too:
  // read from global memory inside the per-submap loop they were seven dependent round trips per thread.
  __shared__ double s_out[EPI_A_PAIRS * 24];
  __shared__ float s_sc[EPI_A_PAIRS], s_loss[EPI_A_PAIRS];
  __shared__ int s_src[EPI_A_PAIRS], s_dst[EPI_A_PAIRS];
  __shared__ unsigned char s_gate[EPI_A_PAIRS];
  bool had = false;      // a pair of this thread's submap is in the loss (passed the gate): its poses get a gradient
  __shared__ float s_total;
  double gR[9] = {0., 0., 0., 0., 0., 0., 0., 0., 0.}, gt[3] = {0., 0., 0.};
  float total = 0.0f;
  const int s = threadIdx.x >> 2, part = threadIdx.x & 3;      // pass (2): submap and quarter of this thread (S <= 64)
  for (int p0 = 0; p0 < k.P; p0 += EPI_A_PAIRS) {
    const int np = min(EPI_A_PAIRS, k.P - p0);
    for (int i = threadIdx.x; i < np * 24; i += blockDim.x) s_out[i] = out[(int64_t)24 * p0 + i];
    __syncthreads();
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
      const int p = p0 + i;
      const AlignPairK& d = k.plan[p];
      const double* o = s_out + 24 * i;
      // (the count and n_ch are small integers: denom is exact in fp32, as the reference's clamp(min=1) * n_ch)
      const float denom = fmaxf((float)o[1], 1.0f) * (k.loss_type == 2 ? d.n_ch : 1.0f);
      const float val = (float)(o[0] / (double)denom);
      const bool finite = (val == val) && !isinf(val);          // nan_to_num passes no gradient otherwise
      const float gate = d.gate_p ? ((cnt[p] / (float)d.gate_n) > k.overlap_thresh ? 1.0f : 0.0f) : 1.0f;
      s_sc[i] = (finite && gate != 0.0f) ? k.align_weight / denom : 0.0f;
      const float v = gate != 0.0f ? nan_to_num_f(val) * k.align_weight : 0.0f;
      s_loss[i] = v;
      pl[p] = v;
      s_src[i] = d.src; s_dst[i] = d.dst;
      s_gate[i] = gate != 0.0f;
    }
    __syncthreads();
    if (k.P <= EPI_A_PAIRS) {
      // the next iteration's launch order: pairs by the number of in-bound vertices they had in this one, most first (a
      // pair's residual costs what it overlaps; with the list's own order a launch ended with its last pairs' workgroups
      // alone on the chip).  Scheduling only: every pair's sums land in its own slot of `out` whatever the order.
      int32_t* order = reinterpret_cast<int32_t*>(k.state + k.L.order);
      for (int i = threadIdx.x; i < np; i += blockDim.x) {
        const double ci = s_out[24 * i + 1];
        int rank = 0;
        for (int j = 0; j < np; ++j) {
          const double cj = s_out[24 * j + 1];
          rank += (cj > ci || (cj == ci && j < i)) ? 1 : 0;
        }
        order[rank] = i + 1;
      }
    }
    if (s < k.S) {
      for (int i = part; i < np; i += 4) {
        const bool is_src = s_src[i] == s, is_dst = s_dst[i] == s;
        const float scf = s_sc[i];
        if (!(is_src || is_dst)) continue;
        had = had || s_gate[i];      // (a non-finite pair loss is nan_to_num'ed: in the loss, with a zero gradient)
        if (scf == 0.0f) continue;
        const double sc = (double)scf;
        const double* o = s_out + 24 * i;
        const float* Rd = pose + 12 * s_dst[i];
        const double h[3] = {(double)Rd[0] * o[2] + (double)Rd[1] * o[3] + (double)Rd[2] * o[4],
                             (double)Rd[3] * o[2] + (double)Rd[4] * o[3] + (double)Rd[5] * o[4],
                             (double)Rd[6] * o[2] + (double)Rd[7] * o[3] + (double)Rd[8] * o[4]};
        if (is_src) {
          for (int q = 0; q < 9; ++q) gR[q] += o[14 + q] * sc;
          for (int q = 0; q < 3; ++q) gt[q] += h[q] * sc;
        } else {
          for (int q = 0; q < 9; ++q) gR[q] += o[5 + q] * sc;
          for (int q = 0; q < 3; ++q) gt[q] -= h[q] * sc;
        }
      }
    }
    if (threadIdx.x == 0)      // pair losses in list order (the reference sums its loss dict in insertion order)
      for (int i = 0; i < np; ++i) total += s_loss[i];
    __syncthreads();
  }
  // the four quarters of a submap: lanes 4 s .. 4 s + 3 of one wavefront
#pragma unroll
  for (int o = 2; o >= 1; o >>= 1) {
#pragma unroll
    for (int q = 0; q < 9; ++q) gR[q] += __shfl_xor(gR[q], o);
#pragma unroll
    for (int q = 0; q < 3; ++q) gt[q] += __shfl_xor(gt[q], o);
    const int had_o = __shfl_xor(had ? 1 : 0, o);      // (not inside the ||: every lane must take part in the exchange)
    had = had || had_o != 0;
  }
  if (s < k.S && part == 0) {
    const float* R0 = k.R0 + 9 * s;
    double G[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) G[i * 3 + j] = R0[i] * gR[j] + R0[3 + i] * gR[3 + j] + R0[6 + i] * gR[6 + j];
    const float* prm = k.state + k.L.params + 6 * s;
    const float w[3] = {prm[0], prm[1], prm[2]};
    double gw[3];
    so3_exp_backward(w, G, gw);
    for (int i = 0; i < 3; ++i) { flat[6 * s + i] = (float)gw[i]; flat[6 * s + 3 + i] = (float)gt[i]; }
    flat[6 * k.S + 1 + s] = had ? 1.0f : 0.0f;
  }
  if (threadIdx.x == 0) flat[6 * k.S] = total;
  (void)s_total;
}

__global__ __launch_bounds__(64) void align_epilogue_b_kernel(AlignK k) {
  int32_t* ctrl = reinterpret_cast<int32_t*>(k.state + k.L.ctrl);
  if (ctrl[CTRL_STOPPED]) return;
  __shared__ float s_total;
  __shared__ float s_nr[64], s_nt[64];           // |dr_s|, |dt_s| BEFORE the step (S <= 64, checked by the entry point)
  __shared__ double s_num[64], s_den[64];
  float* prm = k.state + k.L.params;
  const float* flat = k.state + k.L.flat;
  for (int s = threadIdx.x; s < k.S; s += blockDim.x) {
    const float* p = prm + 6 * s;
    s_nr[s] = sqrtf((p[0] * p[0] + p[1] * p[1]) + p[2] * p[2]);
    s_nt[s] = sqrtf((p[3] * p[3] + p[4] * p[4]) + p[5] * p[5]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = flat[6 * k.S];
    if (k.reg_weight > 0.0f) {      // grid_atlas_pose_trust_region_loss over ALL submaps, in its dict order
      for (int s = 0; s < k.S; ++s) {
        total += k.reg_weight * relu_nan(s_nr[s] - k.reg_rad);
        total += k.reg_weight * relu_nan(s_nt[s] - k.reg_m);
      }
    }
    s_total = total;
  }
  __syncthreads();
  const float total = s_total;
  const bool skip = total != total;      // "Loss at iter .. is nan! Skip backward step." (base.py:147-151)
  const int t = ctrl[CTRL_STEP] + 1;
  int32_t* adam_t = reinterpret_cast<int32_t*>(k.state + k.L.adam_t);
  double num = 0., den = 0.;
  for (int i = threadIdx.x; i < 6 * (k.S - 1); i += blockDim.x) {
    const int s = 1 + i / 6, c = i % 6;      // submap 0 stays fixed (base.py:104-107)
    const int j = 6 * s + c;
    const float old = prm[j];
    float g = flat[j];
    if (k.reg_weight > 0.0f) {               // d/dp of weight * relu(|p| - thresh)
      const float nrm = c < 3 ? s_nr[s] : s_nt[s];
      if (nrm - (c < 3 ? k.reg_rad : k.reg_m) > 0.0f) g += k.reg_weight * (old / nrm);
    }
    // torch.optim.Adam skips a parameter whose .grad is None: a submap none of whose pairs is in the loss this
    // iteration (all gated off) and no regulariser.  Its value, moments and step count stay.
    const bool has = k.reg_weight > 0.0f || flat[6 * k.S + 1 + s] > 0.0f;
    float pv = old;
    if (!skip && has) {
      const int ts = adam_t[s] + 1;          // this submap's own step count (read by its six threads, written below)
      const double bc1 = 1.0 - pow(k.b1, (double)ts), bc2 = 1.0 - pow(k.b2, (double)ts);
      AdamScalars a;
      a.one_minus_b1 = (float)(1.0 - k.b1); a.b2 = (float)k.b2; a.one_minus_b2 = (float)(1.0 - k.b2);
      a.neg_step_size = (float)(-(k.lr / bc1)); a.bc2_sqrt = (float)sqrt(bc2); a.eps = (float)k.eps;
      float m = k.state[k.L.adam_m + j], v = k.state[k.L.adam_v + j];
      adam_one(pv, g, m, v, a);
      k.state[k.L.adam_m + j] = m; k.state[k.L.adam_v + j] = v;
      prm[j] = pv;
    }
    num += ((double)pv - old) * ((double)pv - old);
    den += (double)old * old;
  }
  __syncthreads();                           // every thread has read adam_t
  if (!skip)
    for (int s = 1 + threadIdx.x; s < k.S; s += blockDim.x)
      if (k.reg_weight > 0.0f || flat[6 * k.S + 1 + s] > 0.0f) adam_t[s] += 1;
  __syncthreads();
  s_num[threadIdx.x] = num; s_den[threadIdx.x] = den;
  __syncthreads();
  if (threadIdx.x == 0) {
    double n = 0., d = 0.;
    for (int i = 0; i < 64; ++i) { n += s_num[i]; d += s_den[i]; }
    const int it = ctrl[CTRL_ITER];
    // relative_param_change (utils.py:507-516): after iteration k against after iteration k-1; inf at k = 0
    const float rel = (it == 0) ? INFINITY : (float)sqrt(n / d);
    if (it < k.ring_iters) {
      float* row = k.state + k.L.ring + (int64_t)it * k.L.ring_row;
      row[0] = total; row[1] = rel;
    }
    if (skip) ctrl[CTRL_SKIPPED] += 1; else ctrl[CTRL_STEP] = t;
    if (rel < k.rel_thresh) ctrl[CTRL_STOPPED] = 1;
    ctrl[CTRL_ITER] = it + 1;
  }
  // the next iteration's poses (from the corrections just stepped), their snapshot, cleared pair accumulators: what
  // align_prologue_kernel would do in a launch of its own (miso_align_t.poses_ready)
  __syncthreads();
  if (!ctrl[CTRL_STOPPED]) align_prologue_body(k, ctrl[CTRL_ITER], false);
}

hipError_t launch_pair_batch(const AlignPairK*, int, int64_t, int64_t, bool, const float*, int, double*, float*,
                             const int32_t*, int64_t, const int32_t*, hipStream_t);

hipError_t launch_align_a(const AlignK& k, int64_t max_n, int64_t max_gate_n, int64_t max_gate_rows, bool vec4,
                          bool poses_ready, hipStream_t s) {
  if (!poses_ready) align_prologue_kernel<<<1, 256, 0, s>>>(k);
  const int32_t* stopped = reinterpret_cast<const int32_t*>(k.state + k.L.ctrl) + CTRL_STOPPED;
  hipError_t e = launch_pair_batch(k.plan, k.P, max_n, max_gate_n, vec4, k.state + k.L.pose, k.loss_type,
                                   reinterpret_cast<double*>(k.state + k.L.out), k.state + k.L.cnt, stopped,
                                   max_gate_rows, reinterpret_cast<const int32_t*>(k.state + k.L.order), s);
  if (e != hipSuccess) return e;
  align_epilogue_a_kernel<<<1, EPI_A_THREADS, 0, s>>>(k);
  return hipGetLastError();
}

hipError_t launch_align_b(const AlignK& k, hipStream_t s) {
  align_epilogue_b_kernel<<<1, 64, 0, s>>>(k);
  return hipGetLastError();
}

}  // namespace miso
