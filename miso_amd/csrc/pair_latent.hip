// Fused latent alignment residual of one submap pair: value and pose cotangents in one pass.
//
// Reference: pairwise_loss_latent, grid_opt/align/miso.py:116-211 -- per iteration and pair it
// maps the source submap's cached voxel centres src -> world -> dst (two affine maps,
// grid_opt/utils/utils_geometry.py:214-240), masks them with dst's bound (:11-27), compacts with
// nonzero (host sync), samples ALL levels of both submaps, slices the channels, takes
// mean(diff^2) (L2) or mean(|diff|_2) (L1), and lets autograd walk back through
// grid_sampler_3d_backward (incl. an unused dense grad scatter into both grids), the affine
// maps and so3_exp_map.  Here one kernel does, per source vertex i:
//     w_i = R_s p_i + t_s ;  q_i = R_d^T (w_i - t_d) ;  m_i = q_i in bound(dst)
//     r_i = f_src,i - encode_dst(q_i)              (levels 0..level only)
//     term_i = sum_c r^2  (L2)  |  |r|_2  (L1)
//     g_i = d term_i / d q_i                        (corner-derivative of the same gathers)
// and reduces  sum term, sum m, sum g, sum (w - t_d) g^T, sum (R_d g) p^T  -- everything the
// chain rule needs for d/dR_s, d/dt_s, d/dR_d, d/dt_d (SURVEY App. B "pose chain").  The source
// features are read-only inputs (cached by the caller: they depend on no pose).
#include <stdlib.h>

#include "common.hpp"

namespace miso {

struct PairK {
  const float* pose;   // device: R_s[9] t_s[3] R_d[9] t_d[3], row-major
  const float* p;      // (N,3) source vertices in the source frame
  const float* fsrc;   // (N, ld) source features
  int64_t ld;
  int64_t n;
  int loss_type;       // 1 = L1 (vector norm), 2 = L2
  double* out;         // 24 doubles, zeroed by the launcher
  const float* boxes;  // per run of ALIGN_BOX_VERTS vertices {min xyz, max xyz}, or nullptr (AlignPairK::boxes)
};
constexpr float BOX_SLACK = 2e-3f;      // metres: the exact test rounds a mapped coordinate by ~1e-5 at 100 m

// hardware fp64 add at L2 (global_atomic_add_f64, no return value)
__device__ __forceinline__ void atomic_add_f64(double* p, double v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)(p), v);
}

// One workgroup's share (grid-stride from block bx of nbx) of one pair.  pose_s / pose_d: R[9] t[3] of the source
// and the destination submap.  Shared by the single-pair kernel and the batched one (one launch for all pairs of an
// alignment iteration, grid.y = pair, descriptors in device memory).
// src -> world -> dst of one source vertex (transform_points_to, then transfrom_points_from: utils_geometry.py:214-240)
// in ONE fixed arithmetic: d = (Rs p + ts) - td with the row sums as fma chains, q = Rd^T d likewise.  Every in-bound
// decision -- pass 1 and pass 2 of the pair stage, the point-list and the lattice form of the overlap gate -- is taken on
// these bits.  Written as plain products and sums, the copies of this expression came out of the compiler with different
// contractions, and the two forms of the gate disagreed on one vertex of 4 M for two submaps 0.35 degrees apart
// (test_lattice_overlap_gate_counts_equal_the_point_list_gate).
__device__ __forceinline__ void src_to_dst(const float (&Rs)[9], const float (&ts)[3], const float (&Rd)[9],
                                           const float (&td)[3], float px, float py, float pz, float (&d)[3],
                                           float (&q)[3]) {
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const float w = __fadd_rn(__fmaf_rn(Rs[3 * r + 2], pz, __fmaf_rn(Rs[3 * r + 1], py, __fmul_rn(Rs[3 * r], px))), ts[r]);
    d[r] = __fsub_rn(w, td[r]);
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) q[c] = __fmaf_rn(Rd[6 + c], d[2], __fmaf_rn(Rd[3 + c], d[1], __fmul_rn(Rd[c], d[0])));
}

template <bool VEC4>
__device__ __forceinline__ void pair_latent_body(const GridK& g, const float* __restrict__ pose_s,
                                                 const float* __restrict__ pose_d, const PairK& k, unsigned bx,
                                                 unsigned nbx) {
  float Rs[9], ts[3], Rd[9], td[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Rs[i] = pose_s[i]; Rd[i] = pose_d[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { ts[i] = pose_s[9 + i]; td[i] = pose_d[9 + i]; }
  // per lane: {term, in-bound count, G = sum g_q (3), S = sum p (x) g_q (9)}.  The two 3x3 sums the iteration needs --
  // sum d (x) g_q with d = Rs p + ts - td, and sum (Rd g_q) (x) p -- are linear in S and G: Rs S + (ts - td) (x) G and
  // Rd S^T, formed once per workgroup in double below.  (Accumulating both per vertex was 18 more multiply-adds, the
  // product Rd g_q, and 9 more live accumulators in a kernel at its 128-register budget.)
  constexpr int NACC = 14;
  float acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;

  // Two passes per chunk, per wavefront.  Under a perturbed pose only a fraction of a source submap lands inside the
  // destination (cfg-4: 21 % of the level-1 vertices), and the heavy part -- 8 corner gathers per level, the source
  // features, ~300 flops -- runs for the whole wavefront whenever one lane is inside.  So pass 1 only transforms and
  // tests PAIR_K x 64 consecutive vertices and compacts the in-bound ones into a per-wave LDS list (ballot + prefix:
  // a deterministic order, no block barrier); pass 2 walks that list with all lanes busy.
  constexpr int PAIR_K = 8;
  static_assert(ALIGN_BOX_VERTS == 64 && PAIR_K == 8, "one box per 64-vertex step, eight steps per wavefront and chunk");
  __shared__ uint16_t s_in[4][PAIR_K * 64];
  const int lane_ = threadIdx.x & 63, wave_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned long long lt_mask = (lane_ == 0) ? 0ull : (~0ull >> (64 - lane_));
  const int64_t chunk = (int64_t)blockDim.x * PAIR_K;
  // The composed map q = M p + q_0 (M = Rd^T Rs, q_0 = Rd^T (ts - td)) once per workgroup, in LDS.  It serves two coarse
  // filters, both followed by the exact test of pass 2 (the reference's operation order: w = Rs p + ts, w - td, Rd^T .):
  //  * a step's box in the destination frame: centre M c + q_0, half-extent along destination axis a sum_b |M[a][b]| e_b.
  //    Outside the bound by more than the slack on any axis: no vertex of the step can be in bound, the step is not read.
  //    Lane u < 8 tests the box of step u -- one test per 512 vertices of wavefront time;
  //  * pass 1 itself: a vertex is a candidate when M p + q_0 lies inside the bound widened by the slack -- 12 multiply-adds
  //    on 12 values read back from LDS per chunk, where forming w, w - td and Rd^T . per vertex kept 24 pose values live
  //    as scalar registers the compiler had to spill and reload inside every one of the eight unrolled steps.
  // The slack covers the rounding of either form (~1e-5 m at 100 m, growing with the size of the world coordinates): a
  // vertex the composed map puts inside the bound shrunk by the slack is in bound exactly, one outside the widened bound
  // is out, and the few in between take the exact test.  NaN poses compare false: no candidates, as the exact test finds.
  __shared__ float s_aff[16];
  if (threadIdx.x < 3) {
    const int a = threadIdx.x;
    s_aff[a * 4 + 0] = Rd[a] * Rs[0] + Rd[3 + a] * Rs[3] + Rd[6 + a] * Rs[6];
    s_aff[a * 4 + 1] = Rd[a] * Rs[1] + Rd[3 + a] * Rs[4] + Rd[6 + a] * Rs[7];
    s_aff[a * 4 + 2] = Rd[a] * Rs[2] + Rd[3 + a] * Rs[5] + Rd[6 + a] * Rs[8];
    s_aff[a * 4 + 3] = Rd[a] * (ts[0] - td[0]) + Rd[3 + a] * (ts[1] - td[1]) + Rd[6 + a] * (ts[2] - td[2]);
  }
  if (threadIdx.x == 3)
    s_aff[12] = BOX_SLACK + 4e-6f * (fabsf(ts[0]) + fabsf(ts[1]) + fabsf(ts[2]) + fabsf(td[0]) + fabsf(td[1]) + fabsf(td[2]));
  __syncthreads();
  for (int64_t c0 = (int64_t)bx * chunk; c0 < k.n; c0 += (int64_t)nbx * chunk) {
    const int64_t w0 = c0 + (int64_t)wave_ * (PAIR_K * 64);
    if (w0 >= k.n) continue;
    // (read per chunk, dead after pass 1: kept across the loop they would be 13 more live registers in pass 2, which sits
    // at the kernel's 128-register budget)
    const float4 m0 = *reinterpret_cast<const float4*>(s_aff), m1 = *reinterpret_cast<const float4*>(s_aff + 4),
                 m2 = *reinterpret_cast<const float4*>(s_aff + 8);
    // (wave-uniform: as scalar registers they cost pass 1 no vector registers)
    auto sc = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    const float slack = sc(s_aff[12]);
    const float M[3][4] = {{sc(m0.x), sc(m0.y), sc(m0.z), sc(m0.w)}, {sc(m1.x), sc(m1.y), sc(m1.z), sc(m1.w)},
                           {sc(m2.x), sc(m2.y), sc(m2.z), sc(m2.w)}};
    unsigned live_steps = 0xffu;
    if (k.boxes) {
      const int64_t s0 = w0 + (int64_t)(lane_ & 7) * 64;
      bool reach = false;
      if (s0 < k.n) {
        const float* bb = k.boxes + (s0 / ALIGN_BOX_VERTS) * 6;
        const float c[3] = {0.5f * (bb[0] + bb[3]), 0.5f * (bb[1] + bb[4]), 0.5f * (bb[2] + bb[5])};
        const float e[3] = {0.5f * (bb[3] - bb[0]), 0.5f * (bb[4] - bb[1]), 0.5f * (bb[5] - bb[2])};
        bool outside = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const float qc = M[a][0] * c[0] + M[a][1] * c[1] + M[a][2] * c[2] + M[a][3];
          const float r = fabsf(M[a][0]) * e[0] + fabsf(M[a][1]) * e[1] + fabsf(M[a][2]) * e[2] + slack +
                          1e-6f * (fabsf(qc) + fabsf(M[a][3]));
          outside = outside || qc - r > g.bmax[a] || qc + r < g.bmin[a];
        }
        reach = !outside;
      }
      live_steps = (unsigned)(__ballot(reach) & 0xffull);
      if (live_steps == 0u) continue;
    }
    const float blo[3] = {sc(g.bmin[0] - slack), sc(g.bmin[1] - slack), sc(g.bmin[2] - slack)};
    const float bhi[3] = {sc(g.bmax[0] + slack), sc(g.bmax[1] + slack), sc(g.bmax[2] + slack)};
    int n_in = 0;
    // (this wavefront's vertices through 32-bit offsets from wave-uniform bases: a 64-bit index per vertex is a
    // v_mad_u64_u32 -- quarter rate -- per address)
#pragma unroll
    for (int u = 0; u < PAIR_K; ++u) {
      const int64_t i = w0 + u * 64 + lane_;
      bool cand = false;
      if (i < k.n && ((live_steps >> u) & 1u)) {
        const float px = k.p[i * 3 + 0], py = k.p[i * 3 + 1], pz = k.p[i * 3 + 2];
        const float q0 = M[0][0] * px + M[0][1] * py + M[0][2] * pz + M[0][3];
        const float q1 = M[1][0] * px + M[1][1] * py + M[1][2] * pz + M[1][3];
        const float q2 = M[2][0] * px + M[2][1] * py + M[2][2] * pz + M[2][3];
        cand = q0 >= blo[0] && q0 <= bhi[0] && q1 >= blo[1] && q1 <= bhi[1] && q2 >= blo[2] && q2 <= bhi[2];
        // inside the bound SHRUNK by the slack: in bound by the exact arithmetic as well.  Between the two (within the slack
        // of a face: ~0.1 % of the candidates) the exact test decides, here -- a branch a wavefront takes in one step out of
        // ~15; in pass 2 (a `continue`, or selects at the end) it cost 0.93 - 0.97 ms per cfg-4 level-1 iteration against 0.69
        const bool sure = q0 >= blo[0] + 2.0f * slack && q0 <= bhi[0] - 2.0f * slack && q1 >= blo[1] + 2.0f * slack &&
                          q1 <= bhi[1] - 2.0f * slack && q2 >= blo[2] + 2.0f * slack && q2 <= bhi[2] - 2.0f * slack;
        if (cand && !sure) {
          float d[3], q[3];
          src_to_dst(Rs, ts, Rd, td, px, py, pz, d, q);
          cand = q[0] >= g.bmin[0] && q[0] <= g.bmax[0] && q[1] >= g.bmin[1] && q[1] <= g.bmax[1] &&
                 q[2] >= g.bmin[2] && q[2] <= g.bmax[2];
        }
      }
      const unsigned long long m = __ballot(cand);
      if (cand) s_in[wave_][n_in + (int)__popcll(m & lt_mask)] = (uint16_t)(u * 64 + lane_);
      n_in += (int)__popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int e = lane_; e < n_in; e += 64) {
    const int64_t idx = w0 + s_in[wave_][e];
    // the same arithmetic as pass 1 (three loads from L2 and 30 flops are cheaper than carrying q through LDS)
    const float px = k.p[idx * 3 + 0], py = k.p[idx * 3 + 1], pz = k.p[idx * 3 + 2];
    float d[3], q[3];
    src_to_dst(Rs, ts, Rd, td, px, py, pz, d, q);
    const float* fs = k.fsrc + idx * k.ld;
    // the normalised coordinate is the same for every level (axis_coord op for op, its first half formed once)
    float xn[3], mn[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) axis_norm(q[a], g.bmin[a], g.bmax[a], g.flags, xn[a], mn[a]);
    // pass 1: residual norm (needed by L1 before the derivative weights are known)
    float term = 0.0f, gq[3] = {0.f, 0.f, 0.f};
    float inv_norm = 1.0f;
    for (int pass = (k.loss_type == 1 ? 0 : 1); pass < 2; ++pass) {
      float ss = 0.0f;
      for (int l = 0; l < g.n_levels; ++l) {
        // the level's record by VALUE: one batch of scalar loads at the head of the level (the plan lives in device memory;
        // through a reference every field was fetched where it is used, a dependent scalar round trip each: 0.651 ->
        // 0.632 ms per cfg-4 level-1 iteration)
        const LevelK lv = g.lv[l];
        Axis ax = axis_from_norm(xn[0], mn[0], lv.X, g.flags);
        Axis ay = axis_from_norm(xn[1], mn[1], lv.Y, g.flags);
        Axis az = axis_from_norm(xn[2], mn[2], lv.Z, g.flags);
        Cell c = make_cell(ax, ay, az, lv);
        const bool ign = (g.ignore_mask >> l) & 1u;
        // Corner values with zeros outside the grid (padding_mode = zeros), then the trilinear value and its three
        // derivatives by a lerp tree along x, y, z -- the same polynomial as sum_k v_k w_k with w = (wx wy) wz and its
        // derivative weights, without 32 weight registers per level live next to the 32 corner values: the kernel is
        // bound by the latency of its gathers and ran at two wavefronts per SIMD (225 VGPRs) with the weight form.
        int off[8];
        bool inb8[8];
        // (the base corner's offset once, the others by additions: a 32-bit integer multiply is a quarter-rate instruction,
        // and three per corner were 24 of them per level and vertex)
        const int off0 = c.k0 * lv.sZ + c.j0 * lv.sY + c.i0 * lv.sX;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          const int dx = kk & 1, dy = (kk >> 1) & 1, dz = kk >> 2;
          inb8[kk] = c.inx[dx] && c.iny[dy] && c.inz[dz] && !ign;
          off[kk] = inb8[kk] ? off0 + (dz ? lv.sZ : 0) + (dy ? lv.sY : 0) + (dx ? lv.sX : 0) : 0;
        }
        const float fx = c.wx[1], fy = c.wy[1], fz = c.wz[1];      // weight of the upper corner along each axis
        float ax_ = 0.f, ay_ = 0.f, az_ = 0.f;
        for (int ch = 0; ch < lv.C; ch += (VEC4 ? 4 : 1)) {
          float v[8][4];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) {
            // (loaded whatever the flag -- an out-of-range corner's offset is 0, a valid address -- and zeroed afterwards: a
            // load under `if (in range)` is a branch per corner, and the eight gathers of a level then issue one by one)
            if (VEC4) {
              float4 t = *reinterpret_cast<const float4*>(lv.data + off[kk] + ch);
              if (!inb8[kk]) t = make_float4(0.f, 0.f, 0.f, 0.f);
              v[kk][0] = t.x; v[kk][1] = t.y; v[kk][2] = t.z; v[kk][3] = t.w;
            } else {
              const float t = lv.data[(int64_t)ch * lv.sC + off[kk]];
              v[kk][0] = inb8[kk] ? t : 0.0f;
            }
          }
#pragma unroll
          for (int e = 0; e < (VEC4 ? 4 : 1); ++e) {
            // along x: value and d/dx on the four (y,z) edges
            const float d00 = v[1][e] - v[0][e], d10 = v[3][e] - v[2][e], d01 = v[5][e] - v[4][e], d11 = v[7][e] - v[6][e];
            const float a00 = v[0][e] + fx * d00, a10 = v[2][e] + fx * d10, a01 = v[4][e] + fx * d01, a11 = v[6][e] + fx * d11;
            // along y
            const float e0 = a10 - a00, e1 = a11 - a01;
            const float b0 = a00 + fy * e0, b1 = a01 + fy * e1;
            const float gx0 = d00 + fy * (d10 - d00), gx1 = d01 + fy * (d11 - d01);
            // along z
            const float fto = b0 + fz * (b1 - b0);
            const float gxs = gx0 + fz * (gx1 - gx0), gys = e0 + fz * (e1 - e0), gzs = b1 - b0;
            const float r = fs[lv.foff + ch + e] - fto;
            ss += r * r;
            if (pass == 1) {
              // d term / d f_to: L2 -> -2 r ; L1 -> -r / |r|
              const float gf = (k.loss_type == 2) ? -2.0f * r : -r * inv_norm;
              ax_ += gf * gxs; ay_ += gf * gys; az_ += gf * gzs;
            }
          }
        }
        if (pass == 1) { gq[0] += ax_ * ax.mult; gq[1] += ay_ * ay.mult; gq[2] += az_ * az.mult; }
      }
      if (k.loss_type == 1) {
        const float nrm = sqrtf(ss);
        term = nrm;
        inv_norm = nrm > 0.0f ? 1.0f / nrm : 0.0f;
      } else {
        term = ss;
      }
    }
    const float pp[3] = {px, py, pz};
    acc[0] += term; acc[1] += 1.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) acc[2 + a] += gq[a];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) acc[5 + a * 3 + b] += pp[a] * gq[b];
    }
    __builtin_amdgcn_wave_barrier();      // the list is rewritten by the next chunk's pass 1
  }
  // Reduction: a wavefront's 64 lanes by an fp32 shuffle tree (pairwise: a few ulps), then IN DOUBLE -- the four wave
  // sums of the workgroup and one fp64 atomic per value per workgroup.  Round 2 finished with fp32 atomics: up to 2048
  // workgroups per pair adding one after the other into an fp32 word (error grows with the length of that chain, and
  // the order of the atomics changes from run to run); in double the fan-in adds nothing measurable and the result is
  // reproducible to ~1e-16.  (Shuffling doubles -- two ds_bpermute per step instead of one DPP move -- made the tail of
  // every workgroup three times as long and the level-1 pair stage 1.24 -> 1.9 ms: the tree stays fp32.)
  __shared__ double red[5][NACC + 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NACC; ++i) {
    float v = acc[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    if (lane == 0) red[wave][i] = (double)v;
  }
  __syncthreads();
  if (threadIdx.x < NACC)
    red[4][threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  __syncthreads();
  if (threadIdx.x < 23) {
    // out[0] term, [1] count, [2..4] G, [5 + 3a + b] sum d_a g_b = (Rs S)_ab + (ts - td)_a G_b, [14 + 3a + b] sum (Rd g)_a p_b
    // = sum_c Rd[a][c] S[b][c]
    const double* T = red[4];
    const int t = threadIdx.x;
    double v;
    if (t < 5) {
      v = T[t];
    } else if (t < 14) {
      const int a = (t - 5) / 3, b = (t - 5) % 3;
      // (the poses from memory: indexing the register copies by thread would move them into LDS for the whole kernel)
      v = ((double)pose_s[3 * a] * T[5 + b] + (double)pose_s[3 * a + 1] * T[8 + b]) + (double)pose_s[3 * a + 2] * T[11 + b] +
          ((double)pose_s[9 + a] - (double)pose_d[9 + a]) * T[2 + b];
    } else {
      const int a = (t - 14) / 3, b = (t - 14) % 3;
      v = ((double)pose_d[3 * a] * T[5 + 3 * b] + (double)pose_d[3 * a + 1] * T[6 + 3 * b]) + (double)pose_d[3 * a + 2] * T[7 + 3 * b];
    }
    if (v != 0.0) atomic_add_f64(k.out + t, v);
  }
}

template <bool VEC4>
__global__ __launch_bounds__(256, 4) void pair_latent_kernel(GridK g, PairK k) {
  pair_latent_body<VEC4>(g, k.pose, k.pose + 12, k, blockIdx.x, gridDim.x);
}

// All pairs of one alignment iteration in one launch: blockIdx.y = pair, the pair's descriptor (destination
// levels, source vertices and features) is read from the device-resident plan, the two poses from the (S,12)
// table the prologue kernel of align.hip wrote.  out_all: (P,24) doubles, zeroed by that prologue.
template <bool VEC4>
__global__ __launch_bounds__(256, 4) void pair_latent_batch_kernel(const AlignPairK* __restrict__ plan,
                                                               const float* __restrict__ pose_all, int loss_type,
                                                               double* __restrict__ out_all,
                                                               const int32_t* __restrict__ stopped,
                                                               const int32_t* __restrict__ order) {
  if (stopped && *stopped) return;
  const int o_ = order ? order[blockIdx.y] : 0;      // launch slot -> pair, as in pair_stage_kernel
  const unsigned pair = o_ > 0 ? (unsigned)(o_ - 1) : blockIdx.y;
  const AlignPairK& d = plan[pair];
  if ((int64_t)blockIdx.x * blockDim.x * 8 >= d.n) return;
  PairK k{nullptr, d.p, d.fsrc, d.ld, d.n, loss_type, out_all + 24 * pair, d.boxes};
  pair_latent_body<VEC4>(d.g, pose_all + 12 * d.src, pose_all + 12 * d.dst, k, blockIdx.x, gridDim.x);
}

// Overlap test of two submaps (GridAtlas.check_submap_intersection, grid_opt/models/grid_atlas.py:405-420):
// how many of the source vertices land inside the destination bound after src -> world -> dst.
// The reference materialises both (N,3) transforms and a bool mask for ~4e6 vertices per pair per
// alignment iteration; here one pass, one float out.
__device__ __forceinline__ void overlap_count_body(const float* __restrict__ pose_s, const float* __restrict__ pose_d,
                                                   const float* __restrict__ p, int64_t n, float bmin0,
                                                   float bmin1, float bmin2, float bmax0, float bmax1,
                                                   float bmax2, float* __restrict__ out, unsigned bx, unsigned nbx) {
  float Rs[9], ts[3], Rd[9], td[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Rs[i] = pose_s[i]; Rd[i] = pose_d[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { ts[i] = pose_s[9 + i]; td[i] = pose_d[9 + i]; }
  float cnt = 0.0f;
  for (int64_t idx = (int64_t)bx * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)nbx * blockDim.x) {
    const float px = p[idx * 3 + 0], py = p[idx * 3 + 1], pz = p[idx * 3 + 2];
    // the pair stage's own arithmetic, so both agree on which vertices are inside
    float d[3], q[3];
    src_to_dst(Rs, ts, Rd, td, px, py, pz, d, q);
    const float q0 = q[0], q1 = q[1], q2 = q[2];
    if (q0 >= bmin0 && q0 <= bmax0 && q1 >= bmin1 && q1 <= bmax1 && q2 >= bmin2 && q2 <= bmax2) cnt += 1.0f;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1]) + (red[2] + red[3]);
    if (v != 0.0f) atomic_add_f32(out, v);      // integer-valued partial counts: exact below 2^24 in total
  }
}

__global__ __launch_bounds__(256) void overlap_count_kernel(const float* __restrict__ pose,
                                                           const float* __restrict__ p, int64_t n, float bmin0,
                                                           float bmin1, float bmin2, float bmax0, float bmax1,
                                                           float bmax2, float* __restrict__ out) {
  overlap_count_body(pose, pose + 12, p, n, bmin0, bmin1, bmin2, bmax0, bmax1, bmax2, out, blockIdx.x, gridDim.x);
}

// The overlap gate of every pair of the plan in one launch (blockIdx.y = pair); cnt_all (P) zeroed by the prologue.
// Lattice form (gate_ax): the source's finest level is a lattice whose vertex positions come from three per-axis
// tables (the values FeatureGrid.vertex_positions builds the meshgrid from), and the map into the destination frame is
// affine, so along a lattice row (fixed j, k) every face of the destination bound cuts the row at one point: the
// in-bound vertices of a row are one index interval.  A lane takes a row, solves the six inequalities for the interval
// in fp32, and then decides with the EXACT per-vertex arithmetic of the point-list path (the reference's op order)
// only where rounding could matter -- the two vertices either side of each interval end; the vertices strictly
// between are counted without being evaluated, provided the analytic map leaves them GATE_SLACK inside every face
// (else, and for rows nearly parallel to a face they nearly touch, the whole row is evaluated exactly, all lanes of
// the wavefront on it).  Same count as evaluating every vertex (tests compare the two), at ~1/15 of the instructions:
// the gate of 28 pairs x 4 M vertices took 147 us of a 210 us level-0 alignment iteration.
constexpr int GATE_ROWS = 256;          // rows (lanes) per workgroup
constexpr float GATE_SLACK = 2e-3f;     // metres; fp32 rounding of the map is ~1e-5 at 100 m
constexpr int GATE_AX_LDS = 512;        // x-table entries kept in LDS (ScanNet: 200; a longer table is read from global memory)
constexpr float GATE_ERR = 2e-4f;       // metres: bound on the rounding of one mapped coordinate, generous

// bx of nbx workgroups on pair `pair` (the standalone kernel: blockIdx.x of gridDim.x; the merged pair stage: the
// workgroups behind the pair kernel's own)
__device__ __forceinline__ void overlap_batch_body(const AlignPairK* __restrict__ plan, const float* __restrict__ pose_all,
                                                   float* __restrict__ cnt_all, unsigned pair, unsigned bx, unsigned nbx) {
  const AlignPairK& d = plan[pair];
  if (!d.gate_p) return;
  if (!d.gate_ax[0]) {
    if ((int64_t)bx * blockDim.x >= d.gate_n) return;
    overlap_count_body(pose_all + 12 * d.src, pose_all + 12 * d.dst, d.gate_p, d.gate_n, d.g.bmin[0], d.g.bmin[1],
                       d.g.bmin[2], d.g.bmax[0], d.g.bmax[1], d.g.bmax[2], cnt_all + pair, bx, nbx);
    return;
  }
  const int nx = d.gate_dim[0], ny = d.gate_dim[1], nz = d.gate_dim[2];
  const int nrows = ny * nz;
  const int row0 = bx * GATE_ROWS;
  if (row0 >= nrows) return;
  const float* ps = pose_all + 12 * d.src;
  const float* pd = pose_all + 12 * d.dst;
  float Rs[9], ts[3], Rd[9], td[3];
#pragma unroll
  for (int i = 0; i < 9; ++i) { Rs[i] = ps[i]; Rd[i] = pd[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { ts[i] = ps[9 + i]; td[i] = pd[9 + i]; }
  const float lo[3] = {d.g.bmin[0], d.g.bmin[1], d.g.bmin[2]}, hi[3] = {d.g.bmax[0], d.g.bmax[1], d.g.bmax[2]};
  const float* __restrict__ axg = d.gate_ax[0];
  // the x table in LDS when it fits: the exact tests walk it vertex by vertex, a dependent read each
  __shared__ float s_ax[GATE_AX_LDS];
  const bool ax_lds = nx <= GATE_AX_LDS;
  if (ax_lds) {
    for (int i = threadIdx.x; i < nx; i += blockDim.x) s_ax[i] = axg[i];
    __syncthreads();
  }
  auto ax = [&](int i) -> float { return ax_lds ? s_ax[i] : axg[i]; };
  // the exact test of one vertex: the arithmetic of overlap_count_body / the reference's tensor ops
  auto inside = [&](float px, float py, float pz) -> bool {
    float e[3], q[3];
    src_to_dst(Rs, ts, Rd, td, px, py, pz, e, q);
    const float q0 = q[0], q1 = q[1], q2 = q[2];
    return q0 >= lo[0] && q0 <= hi[0] && q1 >= lo[1] && q1 <= hi[1] && q2 >= lo[2] && q2 <= hi[2];
  };
  const int lane = threadIdx.x & 63;
  const int row = row0 + threadIdx.x;
  const bool live = row < nrows;
  const int j = live ? row % ny : 0, k = live ? row / ny : 0;
  const float py = d.gate_ax[1][j], pz = d.gate_ax[2][k];
  float cnt = 0.0f;
  bool whole = false;                     // this lane's row needs every vertex evaluated
  if (live) {
    if (nx <= 8) {
      for (int i = 0; i < nx; ++i) cnt += inside(ax(i), py, pz) ? 1.0f : 0.0f;
    } else {
      // q_c(px) = alpha_c px + beta_c;  interval of px with all six faces satisfied
      const float x0 = ax(0), x1 = ax(nx - 1);
      float pxl = x0, pxh = x1;           // the row itself
      float unc = 0.0f;                   // how far (in px) rounding can move an interval end: GATE_ERR / |alpha|
      float wpx = 0.0f;                   // how far (in px) from an interval end a vertex can still be within GATE_SLACK of a
                                          // face: GATE_SLACK / |alpha| -- a face the row meets at a shallow angle (two submaps
                                          // a fraction of a degree apart) keeps many vertices that close
      bool empty = false;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float al = Rd[c] * Rs[0] + Rd[3 + c] * Rs[3] + Rd[6 + c] * Rs[6];
        const float be = Rd[c] * (Rs[1] * py + Rs[2] * pz + ts[0] - td[0]) +
                         Rd[3 + c] * (Rs[4] * py + Rs[5] * pz + ts[1] - td[1]) +
                         Rd[6 + c] * (Rs[7] * py + Rs[8] * pz + ts[2] - td[2]);
        // across the row q_c moves by |al| (x1 - x0): a row that hardly moves against this face pair is decided by
        // beta alone -- clearly inside both faces, clearly outside one, or too close to call
        if (fabsf(al) * (x1 - x0) < GATE_SLACK) {
          if (be < lo[c] - 2.0f * GATE_SLACK || be > hi[c] + 2.0f * GATE_SLACK) empty = true;
          else if (!(be > lo[c] + 2.0f * GATE_SLACK && be < hi[c] - 2.0f * GATE_SLACK)) whole = true;
          continue;
        }
        const float a = (lo[c] - be) / al, b = (hi[c] - be) / al;
        pxl = fmaxf(pxl, fminf(a, b));
        pxh = fminf(pxh, fmaxf(a, b));
        unc = fmaxf(unc, GATE_ERR / fabsf(al));
        wpx = fmaxf(wpx, GATE_SLACK / fabsf(al));
      }
      pxl -= unc; pxh += unc;
      if (!(pxl == pxl) || !(pxh == pxh)) whole = true;          // NaN poses: let the exact path decide
      if (!whole && !empty && pxl <= pxh + 4.0f * (x1 - x0) / (float)(nx - 1)) {
        // index estimates from the mean spacing (the tables are linspace-like; the tests around the ends absorb +-1)
        const float inv_dx = (float)(nx - 1) / (x1 - x0);
        // (clamped while still floats: a far-away interval end must not overflow the conversion)
        int il = (int)fminf(fmaxf(floorf((pxl - x0) * inv_dx), -4.0f), (float)nx + 4.0f);
        int ih = (int)fminf(fmaxf(ceilf((pxh - x0) * inv_dx), -4.0f), (float)nx + 4.0f);
        il = max(il - 2, 0); ih = min(ih + 2, nx - 1);           // first / last vertex that could be in bound
        // m vertices either side exactly: 5 when every face is met at more than ~0.4 degrees (a 0.1 m lattice), more for
        // shallower ones -- with a fixed 5 every row of such a pair failed the depth test below and went down the
        // one-row-at-a-time path (cfg-4 level 0: five of the 56 ordered pairs, 28 of the iteration's 80 us)
        const int m = 4 + (int)fminf(ceilf(wpx * inv_dx), 4096.0f);
        if (ih - il < 2 * m + 2) {
          for (int i = il; i <= ih; ++i) cnt += inside(ax(i), py, pz) ? 1.0f : 0.0f;
        } else {
          // vertices il .. il+m-1 and ih-m+1 .. ih exactly; il+m .. ih-m counted if the analytic map keeps both ends of
          // that stretch GATE_SLACK inside every face (q is linear in px, so everything between is at least as deep)
          const int a0 = il + m, a1 = ih - m;
          bool deep = true;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const float al = Rd[c] * Rs[0] + Rd[3 + c] * Rs[3] + Rd[6 + c] * Rs[6];
            const float be = Rd[c] * (Rs[1] * py + Rs[2] * pz + ts[0] - td[0]) +
                             Rd[3 + c] * (Rs[4] * py + Rs[5] * pz + ts[1] - td[1]) +
                             Rd[6 + c] * (Rs[7] * py + Rs[8] * pz + ts[2] - td[2]);
            const float qa = al * ax(a0) + be, qb = al * ax(a1) + be;
            deep = deep && fminf(qa, qb) >= lo[c] + GATE_SLACK && fmaxf(qa, qb) <= hi[c] - GATE_SLACK;
          }
          if (!deep) {
            whole = true;
          } else {
            for (int i = il; i < a0; ++i) cnt += inside(ax(i), py, pz) ? 1.0f : 0.0f;
            for (int i = a1 + 1; i <= ih; ++i) cnt += inside(ax(i), py, pz) ? 1.0f : 0.0f;
            cnt += (float)(a1 - a0 + 1);
          }
        }
      }
      if (whole) cnt = 0.0f;
    }
  }
  // rows that need every vertex: one at a time, all lanes of the wavefront along x
  unsigned long long todo = __ballot(whole);
  while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    const float ry = __shfl(py, src, 64), rz = __shfl(pz, src, 64);
    for (int i = lane; i < nx; i += 64) cnt += inside(ax(i), ry, rz) ? 1.0f : 0.0f;
  }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float v = (red[0] + red[1]) + (red[2] + red[3]);
    if (v != 0.0f) atomic_add_f32(cnt_all + pair, v);
  }
}

__global__ __launch_bounds__(256) void overlap_count_batch_kernel(const AlignPairK* __restrict__ plan,
                                                                 const float* __restrict__ pose_all,
                                                                 float* __restrict__ cnt_all,
                                                                 const int32_t* __restrict__ stopped) {
  if (stopped && *stopped) return;
  overlap_batch_body(plan, pose_all, cnt_all, blockIdx.y, blockIdx.x, gridDim.x);
}

// The pair stage of an alignment iteration as ONE launch (round 4): workgroups [0, pair_blocks) of a pair run the
// latent residual (pair_latent_batch_kernel's body), the gate_blocks behind them the overlap gate
// (overlap_count_batch_kernel's) -- the two do not depend on each other (epilogue A applies the gate), and at level 0
// each is a few tens of microseconds of latency on a mostly idle chip: side by side they take the longer of the two
// instead of the sum plus a launch boundary (cfg-4 level 0: 31 + 25 + ~5 us -> see DESIGN 4.6).
template <bool VEC4>
__global__ __launch_bounds__(256, 4) void pair_stage_kernel(const AlignPairK* __restrict__ plan,
                                                           const float* __restrict__ pose_all, int loss_type,
                                                           double* __restrict__ out_all, float* __restrict__ cnt_all,
                                                           const int32_t* __restrict__ stopped, unsigned pair_blocks,
                                                           const int32_t* __restrict__ order) {
  if (stopped && *stopped) return;
  // launch slot blockIdx.y -> pair: heaviest first when epilogue A has ranked them (align.hip), the list's order until then
  const int o_ = order ? order[blockIdx.y] : 0;
  const unsigned pair = o_ > 0 ? (unsigned)(o_ - 1) : blockIdx.y;
  if (blockIdx.x >= pair_blocks) {
    overlap_batch_body(plan, pose_all, cnt_all, pair, blockIdx.x - pair_blocks, gridDim.x - pair_blocks);
    return;
  }
  const AlignPairK& d = plan[pair];
  if ((int64_t)blockIdx.x * blockDim.x * 8 >= d.n) return;
  PairK k{nullptr, d.p, d.fsrc, d.ld, d.n, loss_type, out_all + 24 * pair, d.boxes};
  pair_latent_body<VEC4>(d.g, pose_all + 12 * d.src, pose_all + 12 * d.dst, k, blockIdx.x, pair_blocks);
}

// boxes[r] = {min, max} over source vertices [64 r, 64 r + 64): one wavefront per run (fminf / fmaxf drop NaNs)
__global__ __launch_bounds__(256) void src_boxes_kernel(const float* __restrict__ p, int64_t n, float* __restrict__ boxes) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t i0 = r * ALIGN_BOX_VERTS;
  if (i0 >= n) return;
  float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
  for (int64_t i = i0 + lane; i < min(n, i0 + ALIGN_BOX_VERTS); i += 64)
#pragma unroll
    for (int a = 0; a < 3; ++a) { const float v = p[i * 3 + a]; lo[a] = fminf(lo[a], v); hi[a] = fmaxf(hi[a], v); }
#pragma unroll
  for (int a = 0; a < 3; ++a)
    for (int o = 32; o > 0; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
  if (lane < 3) boxes[r * 6 + lane] = lane == 0 ? lo[0] : (lane == 1 ? lo[1] : lo[2]);
  else if (lane < 6) boxes[r * 6 + lane] = lane == 3 ? hi[0] : (lane == 4 ? hi[1] : hi[2]);
}

hipError_t launch_src_boxes(const float* p, int64_t n, float* boxes, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const int64_t runs = (n + ALIGN_BOX_VERTS - 1) / ALIGN_BOX_VERTS;
  src_boxes_kernel<<<(unsigned)((runs + 3) / 4), 256, 0, s>>>(p, n, boxes);
  return hipGetLastError();
}

hipError_t launch_overlap_count(const float* pose, const float* p, int64_t n, const float* bmin, const float* bmax,
                                float* out, hipStream_t s) {
  hipError_t e = launch_zero_words(out, 1, s);
  if (e != hipSuccess || n == 0) return e;
  // enough workgroups to pull 12 B per vertex at HBM speed, few enough that their one atomic each
  // (same address, ~13 ns apiece) stays a short tail
  unsigned blocks = (unsigned)((n + 2047) / 2048);
  if (blocks > 512u) blocks = 512u;
  if (blocks < 1u) blocks = 1u;
  overlap_count_kernel<<<blocks, 256, 0, s>>>(pose, p, n, bmin[0], bmin[1], bmin[2], bmax[0], bmax[1], bmax[2], out);
  return hipGetLastError();
}

hipError_t launch_pair_latent(const GridK& g, bool vec4, const float* pose, const float* p, const float* fsrc,
                              int64_t ld, int64_t n, int loss_type, double* out, hipStream_t s) {
  hipError_t e = launch_zero_words(out, 48, s);
  if (e != hipSuccess || n == 0) return e;
  PairK k{pose, p, fsrc, ld, n, loss_type, out, nullptr};
  unsigned blocks = (unsigned)((n + 2047) / 2048);      // a workgroup takes 256 x PAIR_K vertices per trip
  if (blocks > 2048u) blocks = 2048u;
  if (vec4) pair_latent_kernel<true><<<blocks, 256, 0, s>>>(g, k);
  else pair_latent_kernel<false><<<blocks, 256, 0, s>>>(g, k);
  return hipGetLastError();
}

// The pair stage of one fused alignment iteration (align.hip): overlap counts, then the pair residuals.
hipError_t launch_pair_batch(const AlignPairK* plan_dev, int n_pairs, int64_t max_n, int64_t max_gate_n, bool vec4,
                             const float* pose_all, int loss_type, double* out_all, float* cnt_all,
                             const int32_t* stopped, int64_t max_gate_rows, const int32_t* order, hipStream_t s) {
  if (n_pairs <= 0) return hipSuccess;
  // gate workgroups per pair: point-list gates (max_gate_n: the longest list) grid-stride whatever the count, lattice
  // gates (max_gate_rows) want one workgroup per GATE_ROWS rows -- and no more: sized by the lattice's VERTEX count (4 M)
  // the launch carried 512 workgroups per pair of which 79 had rows, 12 000 idle workgroups per iteration at cfg-4
  unsigned gate_blocks = 0;
  if (max_gate_n > 0) {
    gate_blocks = (unsigned)((max_gate_n + 2047) / 2048);
    if (gate_blocks > 512u) gate_blocks = 512u;
  }
  {
    const unsigned lat = (unsigned)((max_gate_rows + GATE_ROWS - 1) / GATE_ROWS);
    if (lat > gate_blocks) gate_blocks = lat;
  }
  static const bool split = getenv("MISO_PAIR_STAGE_SPLIT") != nullptr;      // dev / tests: the two launches of rounds 2-3
  if (gate_blocks && (max_n <= 0 || split))
    overlap_count_batch_kernel<<<dim3(gate_blocks, (unsigned)n_pairs), 256, 0, s>>>(plan_dev, pose_all, cnt_all, stopped);
  if (max_n > 0) {
    unsigned blocks = (unsigned)((max_n + 2047) / 2048);
    // ~7000 workgroups in all (seven rounds of the 1024 that are resident at four waves per SIMD), at least 128 and at
    // most 2048 per pair: a workgroup's fixed part -- poses, the 23-value reduction, 23 atomics -- is paid per
    // workgroup, and with 2048 of them for each of cfg-4's 28 pairs it was a fifth of the level-1 pair stage
    // (1.15 -> 0.89 ms per iteration at 256 per pair)
    static const int cap_all = [] { const char* e = getenv("MISO_PAIR_WGS"); return e ? atoi(e) : 7168; }();      // dev: scan
    unsigned cap = (unsigned)(cap_all / n_pairs);
    cap = cap < 128u ? 128u : (cap > 2048u ? 2048u : cap);
    if (blocks > cap) blocks = cap;
    if (gate_blocks && !split) {
      const dim3 grid(blocks + gate_blocks, (unsigned)n_pairs);
      if (vec4) pair_stage_kernel<true><<<grid, 256, 0, s>>>(plan_dev, pose_all, loss_type, out_all, cnt_all, stopped, blocks, order);
      else pair_stage_kernel<false><<<grid, 256, 0, s>>>(plan_dev, pose_all, loss_type, out_all, cnt_all, stopped, blocks, order);
    } else {
      const dim3 grid(blocks, (unsigned)n_pairs);
      if (vec4) pair_latent_batch_kernel<true><<<grid, 256, 0, s>>>(plan_dev, pose_all, loss_type, out_all, stopped, order);
      else pair_latent_batch_kernel<false><<<grid, 256, 0, s>>>(plan_dev, pose_all, loss_type, out_all, stopped, order);
    }
  }
  return hipGetLastError();
}

}  // namespace miso
