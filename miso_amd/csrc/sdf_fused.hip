// Fused multi-level encode + frozen-decoder MLP, forward and backward, for gfx950.
//
// Replaces GridNet.forward (grid_opt/models/grid_net.py:306-325) and its autograd
// backward with a frozen decoder: features never touch HBM, the decoder runs on
// the matrix cores -- by default as bf16x3 split products on v_mfma_f32_32x32x16_bf16
// (decoder.hpp, mlp_split.hpp: error against float64 equal to exact fp32), behind
// MISO_F_EXACT_F32 as exact fp32 FMA chains on v_mfma_f32_32x32x2_f32 (described
// below; both forms chain the accumulators of one layer into the next) -- and the
// backward needs only the ReLU sign bits saved by the forward (2*H bits per point)
// because the decoder's weights take no gradient.
//
// Work decomposition: one 64-lane wavefront owns a chunk of 64 points and never
// synchronises with other waves (no __syncthreads in the loop), so on one SIMD
// the gather phase of one wave overlaps the MFMA phase of its neighbour.
//
// MFMA chaining.  With D = A(32x2) * B(2x32) + C, points are the N (column)
// dimension and neurons the M (row) dimension.  The accumulator layout of
// 32x32x2 (col = lane&31, row = (j&3) + 8*(j>>2) + 4*(lane>>5) for register j)
// is, register by register, already a legal B operand of the next layer for the
// k-pair {row(j), row(j)+4}: lanes 0-31 hold k=row(j,0), lanes 32-63 hold
// k=row(j,1).  So activations stay in the accumulator registers from layer to
// layer; only the A operands (weights, pre-permuted by mlp_pack_kernel) come
// from LDS.  The first layer's B operand is built from the lane-per-point
// features with one v_permlane32_swap per k-pair.
#include <stdlib.h>

#include "decoder.hpp"

namespace miso {

// Chunk schedule of the persistent waves.  Plain batches: chunk = global wave id,
// grid-strided.  Tile-sorted batches (perm != nullptr): the chunk range is cut into 8
// contiguous parts, one per XCD (blocks are dispatched round-robin over the XCDs,
// block b -> XCD b % 8; a different placement only costs speed).  Spatially
// neighbouring points then stay on one XCD, so its L2 keeps ownership of the grid
// lines they gather from and scatter into: on MI355X an fp32 atomic that misses L2
// costs ~50 ns of request slot (21 G requests/s chip-wide, tools/ubench/atomics.hip),
// and a line bouncing between XCD L2s is the worst case.
// Two wavefronts share a SIMD (and its MFMA pipe).  A wave raises its issue priority while it is in
// a memory phase (corner gathers, scatter) and drops it for the MFMA chain, so that its loads and
// address arithmetic slip in between the co-resident wave's matrix instructions instead of queueing
// behind them.  Measured: forward over unsorted points 80 -> 66 us; sorted 46 -> 45 us.  (Fixed
// per-slot priorities and start delays were tried first: no effect.)
__device__ __forceinline__ void memory_phase(bool on, uint32_t tune, bool first = false) {
  if (tune & 16u) return;   // dev ablation
  if (on) __builtin_amdgcn_s_setprio(3);
  else if ((tune & 64u) && first) __builtin_amdgcn_s_setprio(2);      // dev: the SIMD's first wavefront computes ahead of its second
  else __builtin_amdgcn_s_setprio(0);
}

#ifndef MISO_FWD_OCC
#define MISO_FWD_OCC 2
#endif
#define MISO_FUSED_KERNEL_ATTR

struct ChunkSched {
  int64_t cur, end, step;
  __device__ __forceinline__ ChunkSched(int64_t nchunks, int wave, int nw, bool xcd_local) {
    if (xcd_local && gridDim.x >= 8) {
      const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3;
      const int nlb = (gridDim.x - xcd + 7) >> 3;  // blocks that share this residue
      const int64_t per = (nchunks + 7) / 8;
      const int64_t lo = per * xcd;
      end = lo + per < nchunks ? lo + per : nchunks;
      cur = lo + (int64_t)lb * nw + wave;
      step = (int64_t)nlb * nw;
    } else {
      cur = (int64_t)blockIdx.x * nw + wave;
      end = nchunks;
      step = (int64_t)gridDim.x * nw;
    }
  }
};

// (packed decoder layout: decoder.hpp)

#ifndef MISO_SDF_TRAIN_TU      // (this part is compiled into sdf_fused.o; sdf_train.o holds the training kernel: Makefile)
__global__ void mlp_pack_kernel(MlpK m, int F, int H, int NH, float* __restrict__ out) {
  PackLayout pl(F, H, NH);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pl.total_all) return;
  if (i >= pl.total) {
    // ---- bf16x3 section (decoder.hpp): dword i holds elements 2e, 2e+1 of piece q of one lane's A operand -----------------
    const float* w0 = m.w[0];
    const float* wo = m.w[1 + NH];
    int e = i, rt_m = pl.RT, kind = 0, h = 0;      // kind 0: W0 fwd, 1: Wh[h] fwd, 2: first backward product, 3: Wh[h]^T, 4: W0^T
    if (i < pl.s_wh) { e -= pl.s_w0; kind = 0; }
    else if (i < pl.s_fwd_end) { e -= pl.s_wh; kind = 1; h = e / split_matrix_dwords(pl.KBH, pl.RT); e %= split_matrix_dwords(pl.KBH, pl.RT); }
    else if (i < pl.s_whT) { e -= pl.s_bfirst; kind = 2; if (NH == 0) rt_m = 1; }
    else if (i < pl.s_w0T) { e -= pl.s_whT; kind = 3; h = e / split_matrix_dwords(pl.KBH, pl.RT); e %= split_matrix_dwords(pl.KBH, pl.RT); }
    else { e -= pl.s_w0T; kind = 4; rt_m = 1; }
    const int d = e & 3, lane = (e >> 2) & 63, q = (e >> 8) % 3, kr = (e >> 8) / 3, r = kr % rt_m, kb = kr / rt_m;
    uint32_t word = 0;
    for (int half = 0; half < 2; ++half) {
      const int el = 2 * d + half, hi = lane >> 5, row = 32 * r + (lane & 31);
      float v = 0.0f;
      if (kind == 0) {
        const int k = split_k_feat(kb, hi, el);
        v = (k < F) ? w0[row * F + k] : 0.0f;
      } else if (kind == 1) {
        v = m.w[1 + h][row * H + split_k_acc(kb, hi, el)];
      } else if (kind == 2) {
        const int mm = split_k_acc(kb, hi, el);      // the neuron of the last ReLU this element multiplies
        if (NH >= 1) v = __fmul_rn(m.w[NH][mm * H + row], wo[mm]);
        else v = (row < F) ? __fmul_rn(w0[mm * F + row], wo[mm]) : 0.0f;
      } else if (kind == 3) {
        v = m.w[1 + h][split_k_acc(kb, hi, el) * H + row];
      } else {
        v = (row < F) ? w0[split_k_acc(kb, hi, el) * F + row] : 0.0f;
      }
      uint32_t pc[3];
      bf16_split3(v, pc);
      word |= pc[q] << (16 * half);
    }
    reinterpret_cast<uint32_t*>(out)[i] = word;
    return;
  }
  float v = 0.0f;
  const int RT = pl.RT;
  if (i < pl.o_wh) {  // W0p
    int e = i - pl.o_w0;
    int r = e % RT, l = (e / RT) % 64, s = e / (RT * 64);
    int row = 32 * r + (l & 31), col = 2 * s + (l >> 5);
    v = (col < F) ? m.w[0][row * F + col] : 0.0f;
  } else if (i < pl.o_b0) {  // Whp
    int e = i - pl.o_wh;
    int r = e % RT, l = (e / RT) % 64, ks = (e / (RT * 64)) % pl.KS1, h = e / (RT * 64 * pl.KS1);
    int rp = ks / 16, j = ks % 16;
    int row = 32 * r + (l & 31), col = 32 * rp + row_of(j, l >> 5);
    v = m.w[1 + h][row * H + col];
  } else if (i < pl.o_bh) {
    int e = i - pl.o_b0;
    v = m.b[0] ? m.b[0][e] : 0.0f;
  } else if (i < pl.o_wo) {
    int e = i - pl.o_bh;
    int h = e / H;
    v = m.b[1 + h] ? m.b[1 + h][e % H] : 0.0f;
  } else if (i < pl.o_bo) {
    v = m.w[1 + NH][i - pl.o_wo];
  } else if (i < pl.fwd_end) {
    v = (i == pl.o_bo && m.b[1 + NH]) ? m.b[1 + NH][0] : 0.0f;
  } else if (i < pl.o_w0T) {  // WhTp
    int e = i - pl.o_whT;
    int r = e % RT, l = (e / RT) % 64, ks = (e / (RT * 64)) % pl.KS1, h = e / (RT * 64 * pl.KS1);
    int rp = ks / 16, j = ks % 16;
    int k = 32 * rp + row_of(j, l >> 5), irow = 32 * r + (l & 31);
    v = m.w[1 + h][k * H + irow];
  } else {  // W0Tp
    int e = i - pl.o_w0T;
    int l = e % 64, ks = e / 64;
    int rp = ks / 16, j = ks % 16;
    int k = 32 * rp + row_of(j, l >> 5), f = l & 31;
    v = (f < F) ? m.w[0][k * F + f] : 0.0f;
  }
  out[i] = v;
}

// ---------------------------------------------------------------------------
#endif
#ifndef MISO_SDF_TRAIN_TU      // (this part is compiled into sdf_fused.o; sdf_train.o holds the training kernel: Makefile)
template <int C, int L, int H, int NH, bool SPLIT>
__global__ __launch_bounds__(256, MISO_FWD_OCC) MISO_FUSED_KERNEL_ATTR void sdf_fwd_kernel(GridK g, const float* __restrict__ packed,
                                                        const float* __restrict__ x, int64_t n,
                                                        float* __restrict__ sdf,
                                                        uint32_t* __restrict__ mask,
                                                        const int* __restrict__ perm, LossInK lin) {
  // perm != nullptr: x is the tile-sorted copy of the batch (sort.hip) and the
  // result is written back in the caller's order, sdf[perm[p]].
  // lin.p.loss_type != 0 (binned batches only): the mapping loss of loss.hip is evaluated on the
  // spot -- d loss / d sdf goes to lin.gsdf_sorted[p] (binned order: the backward reads it
  // coalesced), the two loss sums are accumulated into lin.loss_out; sdf may then be NULL.
  constexpr int F = C * L, RT = H / 32, KS0 = (F + 1) / 2, KS1 = H / 2;
  constexpr int MW = (NH + 1) * RT;  // mask words per lane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PackLayout pl(F, H, NH);
  // LDS: the exact form stages the fp32 forward pack [0, fwd_end); the split form the bf16x3 forward block
  // [s_w0, s_fwd_end) followed by the biases and output weights [o_b0, fwd_end)
  const int n_split = pl.s_fwd_end - pl.s_w0;
  if (SPLIT) {
    for (int i = threadIdx.x * 4; i < n_split; i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + pl.s_w0 + i);
    for (int i = threadIdx.x * 4; i < pl.n_bias(); i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + n_split + i) = *reinterpret_cast<const float4*>(packed + pl.o_b0 + i);
  } else {
    for (int i = threadIdx.x * 4; i < pl.fwd_end; i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + i);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hi = lane >> 5;
  const int64_t nchunks = (n + 63) / 64;
  const float* w0p = smem + pl.o_w0;
  const float* whp = smem + pl.o_wh;
  const float* b0 = smem + pl.o_b0;
  const float* bh = smem + pl.o_bh;
  const float* wo = smem + pl.o_wo;
  const uint32_t* s_fwd = reinterpret_cast<const uint32_t*>(smem);      // SPLIT
  const float* s_bias = smem + n_split;                                 // SPLIT
  const float bo = SPLIT ? s_bias[pl.o_bo - pl.o_b0] : smem[pl.o_bo];

  float loss_sdf = 0.0f, loss_fs = 0.0f;
  float inv_n = lin.inv_n;   // mean over the batch rows -- or over the live rows of a padded batch
  if (lin.p.loss_type && lin.n_live) { const int live = *lin.n_live; inv_n = 1.0f / (float)(live > 1 ? live : 1); }
  ChunkSched sched(nchunks, wave, 4, perm != nullptr);
  for (int64_t chunk = sched.cur; chunk < sched.end; chunk += sched.step) {
    // keep the (chunk-invariant) LDS reads of biases / weights inside the loop:
    // hoisted, they cost ~100 VGPRs and spill
    asm volatile("" ::: "memory");
    const int64_t p = chunk * 64 + lane;
    const bool valid = p < n;
    // loss inputs of this lane's point ({target, valid, sign, weight}, one 16-B row in the
    // caller's order): issued now, consumed after the MLP
    const int64_t po = (valid && perm) ? (int64_t)perm[p] : p;
    float4 l_in = make_float4(0.f, 1.f, 0.f, 1.f);
    if (lin.p.loss_type && valid) l_in = lin.aux[po];
    float f[2 * KS0];
#pragma unroll
    for (int i = 0; i < 2 * KS0; ++i) f[i] = 0.0f;
    memory_phase(true, g.tune);
    if (valid && !(g.tune & 2u)) {
      float px, py, pz;
      load_point(g, x, p, px, py, pz);
      float bmn[3] = {g.bmin[0], g.bmin[1], g.bmin[2]}, bmx[3] = {g.bmax[0], g.bmax[1], g.bmax[2]};
      asm volatile("" : "+s"(bmn[0]), "+s"(bmn[1]), "+s"(bmn[2]), "+s"(bmx[0]), "+s"(bmx[1]), "+s"(bmx[2]));
#pragma unroll
      for (int l = 0; l < L; ++l) {
        LevelK lv = g.lv[l];
        if ((g.ignore_mask >> l) & 1u) continue;
        // The level constants are wave-uniform (SGPRs), but everything derived from them in float
        // (sizes, bound) would be hoisted out of the chunk loop into VGPRs and stay live across the
        // MFMA chain; laundering the integers keeps the conversions inside the loop.
        asm volatile("" : "+s"(lv.X), "+s"(lv.Y), "+s"(lv.Z));
        Axis ax = axis_coord(px, bmn[0], bmx[0], lv.X, g.flags);
        Axis ay = axis_coord(py, bmn[1], bmx[1], lv.Y, g.flags);
        Axis az = axis_coord(pz, bmn[2], bmx[2], lv.Z, g.flags);
        Cell c = make_cell(ax, ay, az, lv);
        gather_level<C>(lv, c, &f[l * C]);
      }
    }
    memory_phase(false, g.tune);
    if (g.tune & 4u) {   // dev ablation: gather only
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * KS0; ++i) sum += f[i];
      if (valid) sdf[perm ? (int64_t)perm[p] : p] = sum;
      continue;
    }
    uint32_t mw[MW];
    float p0 = 0.0f, p1 = 0.0f, poison = 0.0f;
    if constexpr (SPLIT) {
      u32x4 no_mask[H / 16][2];
      decoder_fwd_split<F, H, NH, false, true, false>(s_fwd, s_bias, lane, f, mw, no_mask, p0, p1, poison);
    } else {
      decoder_fwd_exact<F, H, NH>(w0p, whp, b0, bh, wo, lane, f, mw, p0, p1);
    }      // exact fp32 chains
    p0 += __shfl_xor(p0, 32);
    p1 += __shfl_xor(p1, 32);
    const float sdf_v = SPLIT ? ((hi ? p1 : p0) + bo) + poison : (hi ? p1 : p0) + bo;
    if (valid && sdf) sdf[po] = sdf_v;
    if (lin.p.loss_type && valid) {
      float gsd, gfs;
      map_loss_one(lin.p, sdf_v, l_in.x, l_in.w, l_in.y == 1.0f, lin.p.w_fs > 0.f && l_in.z == 1.0f, gsd, gfs,
                   loss_sdf, loss_fs);
      lin.gsdf_sorted[p] = (gsd + gfs) * inv_n;
    }
    if (mask) {
      uint32_t* mo = mask + (chunk * 64 + lane) * MW;
#pragma unroll
      for (int i = 0; i < MW; ++i) mo[i] = mw[i];
    }
  }
  if (lin.p.loss_type) {
    // block reduction through LDS (the weights are dead by now), then every block STORES its pair
    // into its own slot (and clears the slots no block owns): no atomics -- 1024 of them on two
    // addresses would be a 13 us serialised tail -- and nothing for the caller to zero.
    for (int o = 32; o > 0; o >>= 1) { loss_sdf += __shfl_down(loss_sdf, o); loss_fs += __shfl_down(loss_fs, o); }
    __syncthreads();
    if (lane == 0) { smem[2 * wave] = loss_sdf; smem[2 * wave + 1] = loss_fs; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float a = (smem[0] + smem[2]) + (smem[4] + smem[6]), b = (smem[1] + smem[3]) + (smem[5] + smem[7]);
      float2* slots = reinterpret_cast<float2*>(lin.loss_out);
      slots[blockIdx.x] = make_float2(lin.p.w_sdf * a * inv_n, lin.p.w_fs * b * inv_n);
      for (int sl = blockIdx.x + gridDim.x; sl < MISO_LOSS_SLOTS; sl += gridDim.x) slots[sl] = make_float2(0.f, 0.f);
    }
  }
}

// ---------------------------------------------------------------------------
// Backward: grad_sdf -> (MFMA chain through the transposed weights, gated by the
// saved ReLU bits) -> d feats in accumulator layout -> scatter-add into the level
// gradients and/or grad_x.  Lane (hi, l&31) holds, for tile t, point 32t+(l&31):
//   C == 8: channels 4hi..4hi+3 of level j>>2      (registers j = 4*level + c)
//   C == 4: channels 0..3 of level 2*(j>>2) + hi   (registers j = 4*g + c)
template <int C, int L, int H, int NH, bool WANT_GRID, bool WANT_X, bool SPLIT>
__global__ __launch_bounds__(256, 2) MISO_FUSED_KERNEL_ATTR void sdf_bwd_kernel(GridK g, const float* __restrict__ packed,
                                                        const float* __restrict__ x, int64_t n,
                                                        const float* __restrict__ gsdf,
                                                        const uint32_t* __restrict__ mask,
                                                        float* __restrict__ gx,
                                                        const int* __restrict__ perm, int debug,
                                                        float* __restrict__ dfeat_out,
                                                        uint32_t defer_mask) {
  // perm != nullptr: x and mask are in tile-sorted order, gsdf / gx in the caller's.
  // dfeat_out != nullptr: rows of d(feats) (N,F, sorted order) are written out and the levels in
  // defer_mask are NOT scattered here: grad_pull_kernel (grad_pull.hip) forms their gradient owner-computes.
  // debug: ablation switches (MISO_DEBUG_BWD, dev only): 1 = no atomics, 8 = no scatter; bit 16 (set by the
  // launcher for MISO_F_GRAD_SDF_SORTED): gsdf is already in the binned order.
  constexpr int F = C * L, RT = H / 32, KS1 = H / 2;
  constexpr int MW = (NH + 1) * RT;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PackLayout pl(F, H, NH);
  // stage wo + the transposed weights: [o_wo, o_bo) and [o_whT, total); the split form its backward block
  // [s_bfirst, total_all) (the output weights are folded into its first matrix; the H floats behind it stay unused)
  const int nb = SPLIT ? pl.total_all - pl.s_bfirst : pl.total - pl.o_whT;
  for (int i = threadIdx.x * 4; i < nb; i += blockDim.x * 4)
    *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + (SPLIT ? pl.s_bfirst : pl.o_whT) + i);
  for (int i = threadIdx.x; i < H; i += blockDim.x) smem[nb + i] = packed[pl.o_wo + i];
  __syncthreads();
  const float* whT = smem;
  const float* w0T = smem + (pl.o_w0T - pl.o_whT);
  const float* wo = smem + nb;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hi = lane >> 5;
  constexpr int FP = ((F + 3) / 4) * 4 + 4;  // d-feat row pitch: 16-B aligned, conflict-free b128 writes
  constexpr int REC = 8;                     // ints per (point, level) cell record
  constexpr int WAVE_LDS = 64 * FP + 64 * L * REC;
  // debug bit 32 (set by the launcher when nothing is scattered from here): only the d-feat tile is
  // allocated per wave -- 50 KB per workgroup instead of 74, i.e. three workgroups per CU
  float* wave_lds = smem + ((nb + H + 3) / 4) * 4 + wave * ((debug & 32) ? 64 * FP : WAVE_LDS);
  const int64_t nchunks = (n + 63) / 64;
  // levels whose gradient is scattered from this kernel (binned training defers all of them to the
  // pull: then no per-point cell records are formed at all)
  uint32_t scatter_mask = 0;
  if (WANT_GRID)
    for (int l = 0; l < L; ++l)
      if (g.lv[l].grad && !((g.ignore_mask >> l) & 1u) && !((defer_mask >> l) & 1u)) scatter_mask |= 1u << l;

  ChunkSched sched(nchunks, wave, 4, perm != nullptr);
  for (int64_t chunk = sched.cur; chunk < sched.end; chunk += sched.step) {
    asm volatile("" ::: "memory");  // see sdf_fwd_kernel
    const int64_t pt[2] = {chunk * 64 + (lane & 31), chunk * 64 + 32 + (lane & 31)};
    float ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
      ds[t] = (pt[t] < n) ? gsdf[(perm && !(debug & 16)) ? (int64_t)perm[pt[t]] : pt[t]] : 0.0f;   // 16: gsdf is binned
    uint32_t mw[MW];
    {
      const uint32_t* mi = mask + (chunk * 64 + lane) * MW;
#pragma unroll
      for (int i = 0; i < MW; ++i) mw[i] = mi[i];
    }
    f32x16 df[2];
    if constexpr (SPLIT) {
      u32x4 maskB[H / 16][2];
      mask_operand_from_bits<H, NH, false>(mw, maskB);
      decoder_bwd_split<F, H, NH, false>(reinterpret_cast<const uint32_t*>(smem), lane, maskB, mw, ds, df);
    } else {
    // d(last hidden) = wo * ds, gated.  Two accumulator sets ping-pong.
    f32x16 dbuf[2][RT][2];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float wv = wo[32 * r + row_of(j, hi)];
#pragma unroll
        for (int t = 0; t < 2; ++t)
          dbuf[0][r][t][j] = gate(wv * ds[t], mw[NH * RT + r], t, j);
      }
#pragma unroll
    for (int hh = 0; hh < NH; ++hh) {
      const int h = NH - 1 - hh;
      const int ci = hh & 1, ni = ci ^ 1;
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) { dbuf[ni][r][0][j] = 0.0f; dbuf[ni][r][1][j] = 0.0f; }
#pragma unroll
      for (int rp = 0; rp < RT; ++rp)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int ks = rp * 16 + j;
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            float a = whT[((h * KS1 + ks) * 64 + lane) * RT + r];
            dbuf[ni][r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dbuf[ci][rp][0][j], dbuf[ni][r][0], 0, 0, 0);
            dbuf[ni][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dbuf[ci][rp][1][j], dbuf[ni][r][1], 0, 0, 0);
          }
        }
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j)
            dbuf[ni][r][t][j] = gate(dbuf[ni][r][t][j], mw[h * RT + r], t, j);
    }
    f32x16 (&d)[RT][2] = dbuf[NH & 1];
    // d feats = W0^T d   (one 32-row tile; rows >= F are zero)
#pragma unroll
    for (int j = 0; j < 16; ++j) { df[0][j] = 0.0f; df[1][j] = 0.0f; }
#pragma unroll
    for (int rp = 0; rp < RT; ++rp)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float a = w0T[(rp * 16 + j) * 64 + lane];
        df[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[rp][0][j], df[0], 0, 0, 0);
        df[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[rp][1][j], df[1], 0, 0, 0);
      }
    }      // exact fp32 chains
    // ---- scatter into the level gradients ----------------------------------------
    // The L2 executes fp32 atomics per 64-byte request (~21 G requests/s on MI355X,
    // tools/ubench/atomics.hip), however many of its 16 dwords carry data.  So the
    // scatter runs "row-major": the 2*C consecutive lanes of a group cover the
    // x-pair (i0, i0+1) x C channels = one contiguous run of 2*C floats, and one
    // atomic instruction serves 64/(2C) (point, row) pairs.  d feats move from the
    // accumulator layout to that lane order through a per-wave LDS tile; the cell
    // of every (point, level) is computed once (lane = point) and broadcast from LDS.
    memory_phase(true, g.tune);
    if (WANT_GRID && !(debug & 8)) {
      float* dF = wave_lds;                         // [64][FP]
      int* rec = reinterpret_cast<int*>(wave_lds + 64 * FP);   // [64][L][REC]
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int gq = 0; gq < (F + 7) / 8; ++gq) {
          const int f0 = 8 * gq + 4 * hi;
          if (f0 < F)
            *reinterpret_cast<float4*>(dF + (32 * t + (lane & 31)) * FP + f0) =
                make_float4(df[t][4 * gq], df[t][4 * gq + 1], df[t][4 * gq + 2], df[t][4 * gq + 3]);
        }
      if (scatter_mask) {     // cell records only where a level is still scattered from here
        const int64_t p = chunk * 64 + lane;
        const bool valid = p < n;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) load_point(g, x, p, px, py, pz);
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const LevelK& lv = g.lv[l];
          Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
          Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
          Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
          Cell c = make_cell(ax, ay, az, lv);
          int flags = (c.inx[0] ? 1 : 0) | (c.inx[1] ? 2 : 0) | (c.iny[0] ? 4 : 0) | (c.iny[1] ? 8 : 0) |
                      (c.inz[0] ? 16 : 0) | (c.inz[1] ? 32 : 0);
          if (!valid) flags = 0;
          int* r = rec + (lane * L + l) * REC;
          *reinterpret_cast<int4*>(r) = make_int4(c.k0 * lv.sZ + c.j0 * lv.sY + c.i0 * lv.sX, flags,
                                                  __float_as_int(c.wx[1]), __float_as_int(c.wy[1]));
          *reinterpret_cast<int4*>(r + 4) = make_int4(__float_as_int(c.wz[1]), __float_as_int(c.wx[0]),
                                                      __float_as_int(c.wy[0]), __float_as_int(c.wz[0]));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (dfeat_out) {
        // the chunk's 64 rows are contiguous in the (N,F) buffer: coalesced 16-B stores
        float* dst = dfeat_out + chunk * 64 * F;
        const int64_t rows_left = n - chunk * 64;
        for (int i = lane; i < 64 * F / 4; i += 64) {
          const int row = (i * 4) / F, col = (i * 4) % F;
          if (row < rows_left)
            *reinterpret_cast<float4*>(dst + row * F + col) = *reinterpret_cast<const float4*>(dF + row * FP + col);
        }
      }
      constexpr int LPR = 2 * C, SLOTS = 64 / LPR;
      const int slot = lane / LPR, dx = (lane / C) & 1, ch = lane % C;
#pragma unroll 1
      for (int pg = 0; pg < (scatter_mask ? 64 / SLOTS : 0); ++pg) {
        const int pt = pg * SLOTS + slot;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const LevelK& lv = g.lv[l];
          if (!lv.grad || ((g.ignore_mask >> l) & 1u) || ((defer_mask >> l) & 1u)) continue;
          const int* r = rec + (pt * L + l) * REC;
          const int4 r0 = *reinterpret_cast<const int4*>(r);
          const int4 r1 = *reinterpret_cast<const int4*>(r + 4);
          const int fl = r0.y;
          if (!((fl >> dx) & 1)) continue;
          const float v = dF[pt * FP + l * C + ch];
          const float wx = dx ? __int_as_float(r0.z) : __int_as_float(r1.y);
          const float wy[2] = {__int_as_float(r1.z), __int_as_float(r0.w)};
          const float wz[2] = {__int_as_float(r1.w), __int_as_float(r1.x)};
          float* base = lv.grad + r0.x + dx * lv.sX + ch;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int dy = q & 1, dz = q >> 1;
            if (((fl >> (2 + dy)) & 1) && ((fl >> (4 + dz)) & 1) && !(debug & 1)) {
              atomic_add_f32(base + dy * lv.sY + dz * lv.sZ, v * ((wx * wy[dy]) * wz[dz]));
              if (ch == 0) touch_chunk(lv, r0.x + dx * lv.sX + dy * lv.sY + dz * lv.sZ);   // C floats: one chunk
            }
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // ---- coordinate gradient (pose path): lane = (point, channel half) -----------
    if (WANT_X) {
      float gacc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bool valid = pt[t] < n;
        float px = 0.f, py = 0.f, pz = 0.f;
        if (valid) load_point(g, x, pt[t], px, py, pz);
        constexpr int NG = (C == 8) ? L : (L + 1) / 2;  // register groups of 4
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          const int l = (C == 8) ? gi : 2 * gi + hi;
          const int choff = (C == 8) ? 4 * hi : 0;
          if (l >= L || !valid) continue;
          if ((g.ignore_mask >> l) & 1u) continue;
          // C == 4: the two lane halves work on different levels; pick the level's
          // fields with per-lane selects (a lane-varying index into the kernel
          // arguments would be spilled to scratch).
          LevelK lv = g.lv[(C == 8) ? gi : 2 * gi];
          if (C == 4 && 2 * gi + 1 < L && hi) lv = g.lv[(2 * gi + 1 < L) ? 2 * gi + 1 : 0];
          Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
          Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
          Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
          Cell c = make_cell(ax, ay, az, lv);
          const float v0 = df[t][4 * gi + 0], v1 = df[t][4 * gi + 1], v2 = df[t][4 * gi + 2],
                      v3 = df[t][4 * gi + 3];
          float sx_ = 0.f, sy_ = 0.f, sz_ = 0.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
            bool in = c.inx[dx] && c.iny[dy] && c.inz[dz];
            if (!in) continue;
            int off = (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX + choff;
            float4 gv = *reinterpret_cast<const float4*>(lv.data + off);
            float dot = gv.x * v0 + gv.y * v1 + gv.z * v2 + gv.w * v3;
            float sx = dx ? 1.f : -1.f, sy = dy ? 1.f : -1.f, sz = dz ? 1.f : -1.f;
            sx_ += dot * sx * c.wy[dy] * c.wz[dz];
            sy_ += dot * sy * c.wx[dx] * c.wz[dz];
            sz_ += dot * sz * c.wx[dx] * c.wy[dy];
          }
          gacc[t][0] += sx_ * (g.gscale[0] * ax.mult);
          gacc[t][1] += sy_ * (g.gscale[1] * ay.mult);
          gacc[t][2] += sz_ * (g.gscale[2] * az.mult);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 3; ++a) gacc[t][a] += __shfl_xor(gacc[t][a], 32);
      const int64_t p = chunk * 64 + lane;
      if (p < n) {
        const int64_t po = perm ? (int64_t)perm[p] : p;
        gx[po * 3 + 0] = hi ? gacc[1][0] : gacc[0][0];
        gx[po * 3 + 1] = hi ? gacc[1][1] : gacc[0][1];
        gx[po * 3 + 2] = hi ? gacc[1][2] : gacc[0][2];
      }
    }
    memory_phase(false, g.tune);
  }
}

#endif
#ifdef MISO_SDF_TRAIN_TU       // (compiled into sdf_train.o, under the max-ILP machine scheduler: Makefile)
// ---------------------------------------------------------------------------
// Training step in ONE kernel (binned batches, frozen decoder, every level's grid gradient left to the pull / push):
// gather -> decoder forward -> mapping loss -> decoder backward -> d-feat rows, per 64-point chunk, per wavefront.
// What sdf_fwd_kernel + sdf_bwd_kernel<.., true, false> do as two launches, minus everything that only carried state
// from one to the other: the ReLU sign bits (16 B per point written and read back), d loss / d sdf (4 + 4 B), the second
// kernel's launch, ramp and tail, its own staging of the weights -- and, what matters most on gfx950, the two phases
// now share a SIMD: fp32 MFMA and VALU use one datapath (tools/ubench/mfma_valu.hip), so a kernel's floor is the SUM
// of its matrix and vector clocks, but a wavefront waiting for its corner gathers costs the neighbour's matrix chain
// nothing.  The backward pass is ~all matrix work (it reads 20 B per point), the forward has the long memory phase:
// with both in one kernel a wavefront's gathers hide behind TWICE the matrix work of its neighbour.
// The bodies are those of the two kernels above (kept separate: they also serve inference, the unsorted path, the
// coordinate backward and levels scattered from the backward); the sign bits stay in the registers they were formed
// in, d loss / d sdf moves between the point-per-lane layout of the loss and the two 32-point tiles of the backward
// with two lane reads.
// SCAT: levels with a gradient that are NOT in defer_mask (bricks beyond what the pull owns: cfg-3's fine level; or
// every level of an unbinned batch, perm == nullptr) are scattered from here with float atomics exactly as
// sdf_bwd_kernel<.., true, false> does it -- per-point cell records kept in LDS from the forward's gather, lanes
// (point slot, dx, channel) walking the chunk's d-feat tile.  dfeat_out may then be null (nothing deferred).
// HALF: 32 points per wavefront and trip instead of 64 -- lanes 32..63 mirror lanes 0..31 (the same point, the same gather),
// only the first of the two 32-point matrix tiles is computed.  For batches that are one chunk per wavefront anyway (a
// few thousand samples: Newer College's 6 144, the tracker's windows): the wavefront's chain of matrix instructions halves,
// twice as many wavefronts share the batch.  Same arithmetic per point.
#ifdef MISO_ABL_NO_GATHER     // dev ablation (wrong results): the cell arithmetic without the corner loads
#define MISO_TRAIN_GATHER_LEVEL(lv, c, fo) \
  for (int q = 0; q < C; ++q) (fo)[q] = c.wx[0] * (float)(c.i0 + q) + c.wy[1] * (float)c.j0 + c.wz[0] * (float)c.k0
#else
#define MISO_TRAIN_GATHER_LEVEL(lv, c, fo) gather_level<C>(lv, c, fo)
#endif
// A chunk's input side (sdf_train_kernel): its point, label row and corner gathers (-> f) and, scattering, its cell records
// (-> recw: base offset, in-bound bits, the six weights, as sdf_bwd_kernel forms them from the point again).  A macro, not a
// lambda: the non-scattering instantiations must compile to the loop they had before the scattering ones learnt to
// request a chunk's gathers one chunk early.
#define MISO_TRAIN_GATHER(CHUNK_, RECW_, P_O_, PO_O_, VALID_O_, LIN_O_)                                        \
  {                                                                                                         \
    const int64_t gp_ = HALF ? (CHUNK_) * 32 + (lane & 31) : (CHUNK_) * 64 + lane; \
    const bool gvalid_ = gp_ < n; \
    int64_t gpo_ = gp_; \
    if (gvalid_ && perm) gpo_ = (int64_t)perm[gp_]; \
    else if (gvalid_ && (g.flags & MISO_F_INDEX_IN_XN)) gpo_ = (int64_t)__float_as_int(reinterpret_cast<const float4*>(x)[gp_].w); \
    float4 glin_ = make_float4(0.f, 1.f, 0.f, 1.f); \
    if (gvalid_) glin_ = lin.aux[gpo_]; \
_Pragma("unroll") \
    for (int i = 0; i < 2 * KS0; ++i) f[i] = 0.0f; \
    memory_phase(true, g.tune); \
    if (gvalid_) { \
      float px, py, pz; \
      load_point(g, x, gp_, px, py, pz); \
      float bmn[3] = {g.bmin[0], g.bmin[1], g.bmin[2]}, bmx[3] = {g.bmax[0], g.bmax[1], g.bmax[2]}; \
      asm volatile("" : "+s"(bmn[0]), "+s"(bmn[1]), "+s"(bmn[2]), "+s"(bmx[0]), "+s"(bmx[1]), "+s"(bmx[2])); \
_Pragma("unroll") \
      for (int l = 0; l < L; ++l) { \
        LevelK lv = g.lv[l]; \
        if ((g.ignore_mask >> l) & 1u) continue; \
        asm volatile("" : "+s"(lv.X), "+s"(lv.Y), "+s"(lv.Z)); \
        Axis ax = axis_coord(px, bmn[0], bmx[0], lv.X, g.flags); \
        Axis ay = axis_coord(py, bmn[1], bmx[1], lv.Y, g.flags); \
        Axis az = axis_coord(pz, bmn[2], bmx[2], lv.Z, g.flags); \
        Cell c = make_cell(ax, ay, az, lv); \
      MISO_TRAIN_GATHER_LEVEL(lv, c, &f[l * C]); \
        if (SCAT && ((scatter_mask >> l) & 1u)) { \
          const int flags = (c.inx[0] ? 1 : 0) | (c.inx[1] ? 2 : 0) | (c.iny[0] ? 4 : 0) | (c.iny[1] ? 8 : 0) | \
                            (c.inz[0] ? 16 : 0) | (c.inz[1] ? 32 : 0); \
          int* r = (RECW_) + (row_l * L + l) * REC; \
          *reinterpret_cast<int4*>(r) = make_int4(c.k0 * lv.sZ + c.j0 * lv.sY + c.i0 * lv.sX, flags, \
                                                  __float_as_int(c.wx[1]), __float_as_int(c.wy[1])); \
          *reinterpret_cast<int4*>(r + 4) = make_int4(__float_as_int(c.wz[1]), __float_as_int(c.wx[0]), \
                                                      __float_as_int(c.wy[0]), __float_as_int(c.wz[0])); \
        } \
      } \
    } else if (SCAT) { \
_Pragma("unroll") \
      for (int l = 0; l < L; ++l) (RECW_)[(row_l * L + l) * REC + 1] = 0; \
    } \
    memory_phase(false, g.tune, wave < NW / 2); \
    P_O_ = gp_; PO_O_ = gpo_; VALID_O_ = gvalid_; LIN_O_ = glin_; \
  }
template <int C, int L, int H, int NH, bool SCAT, int NW = 4, bool HALF = false, bool SPLIT = false>
__global__ __launch_bounds__(64 * NW, 2) MISO_FUSED_KERNEL_ATTR void sdf_train_kernel(GridK g, const float* __restrict__ packed,
                                                          const float* __restrict__ x, int64_t n,
                                                          float* __restrict__ sdf, const int* __restrict__ perm,
                                                          LossInK lin, float* __restrict__ dfeat_out,
                                                          uint32_t defer_mask) {
  constexpr int F = C * L, RT = H / 32, KS0 = (F + 1) / 2, KS1 = H / 2;
  constexpr int MW = (NH + 1) * RT;
  constexpr int FP = ((F + 3) / 4) * 4 + 4;      // d-feat row pitch in LDS: 16-B aligned, conflict-free b128 writes
  constexpr int REC = 8;                         // ints per (point, level) cell record (SCAT)
  // SCAT: the cell records are double-buffered when the launcher found room for a second block (MISO_TUNE_ROTATE) -- the
  // next chunk's gathers are then issued in FRONT of this chunk's atomics, see the loop
  const bool rotate = SCAT && C * L <= MISO_ROTATE_MAX_F && (g.tune & MISO_TUNE_ROTATE) != 0;
  const int WAVE_LDS = 64 * FP + (SCAT ? (rotate ? 2 : 1) * 64 * L * REC : 0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const PackLayout pl(F, H, NH);
  // the whole pack: forward part [0, fwd_end), transposed weights [o_whT, total) right behind it; the split form: the
  // bf16x3 section [s_w0, total_all), then the biases and output weights [o_b0, fwd_end)
  const int n_split = pl.total_all - pl.s_w0;
  const int n_pack = SPLIT ? n_split + ((pl.n_bias() + 3) / 4) * 4 : ((pl.total + 3) / 4) * 4;
  if (SPLIT) {
    for (int i = threadIdx.x * 4; i < n_split; i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + pl.s_w0 + i);
    for (int i = threadIdx.x * 4; i < pl.n_bias(); i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + n_split + i) = *reinterpret_cast<const float4*>(packed + pl.o_b0 + i);
  } else {
    for (int i = threadIdx.x * 4; i < pl.total; i += blockDim.x * 4)
      *reinterpret_cast<float4*>(smem + i) = *reinterpret_cast<const float4*>(packed + i);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), hi = lane >> 5;
  constexpr int PTS = HALF ? 32 : 64;      // points per wavefront and trip
  const int64_t nchunks = (n + PTS - 1) / PTS;
  const float* w0p = smem + pl.o_w0;
  const float* whp = smem + pl.o_wh;
  const float* b0 = smem + pl.o_b0;
  const float* bh = smem + pl.o_bh;
  const float* wo = smem + pl.o_wo;
  const uint32_t* s_fwd = reinterpret_cast<const uint32_t*>(smem);                                   // SPLIT
  const uint32_t* s_bwd = reinterpret_cast<const uint32_t*>(smem) + (pl.s_bfirst - pl.s_w0);        // SPLIT
  const float* s_bias = smem + n_split;                                                              // SPLIT
  const float bo = SPLIT ? s_bias[pl.o_bo - pl.o_b0] : smem[pl.o_bo];
  const float* whT = smem + pl.o_whT;
  const float* w0T = smem + pl.o_w0T;
  float* dF = smem + n_pack + wave * WAVE_LDS;       // this wavefront's d-feat tile [64][FP]
  int* rec = reinterpret_cast<int*>(dF + 64 * FP);                      // SCAT: its cell records [64][L][REC]
  uint32_t scatter_mask = 0;
  if (SCAT)
    for (int l = 0; l < L; ++l)
      if (g.lv[l].grad && !((g.ignore_mask >> l) & 1u) && !((defer_mask >> l) & 1u)) scatter_mask |= 1u << l;

  float loss_sdf = 0.0f, loss_fs = 0.0f;
  float inv_n = lin.inv_n;
  if (lin.n_live) { const int live = *lin.n_live; inv_n = 1.0f / (float)(live > 1 ? live : 1); }
  ChunkSched sched(nchunks, wave, NW, true);
  // dev (MISO_TUNE bits 8..15): the second wavefront of every SIMD starts k x 1024 clocks late
  if ((g.tune >> 8) & 255u) {
    if (wave >= NW / 2)
      for (uint32_t i = 0; i < ((g.tune >> 8) & 255u); ++i) __builtin_amdgcn_s_sleep(16);
  }
  const int row_l = HALF ? (lane & 31) : lane;      // this lane's row of the wavefront's LDS tile / records
  float f[2 * KS0];
  // Rotated (scattering, room for two record blocks): chunk k+1's gathers are requested between chunk k's decoder
  // backward and its atomics.  The atomics execute at the memory side at a fixed rate and queue up in the CU's memory
  // pipeline; a gather requested behind them waits for all of them, and with every wavefront of the launch in the same
  // phase the kernel took (decoder time) + (atomic time).  Requested in front of them, the next chunk's rows arrive
  // while the atomics drain and its decoder runs under them (cfg-3 trainer step 295 -> 279 us).
  int* rec_cur = rec;
  int* rec_nxt = rotate ? rec + 64 * L * REC : rec;
  int64_t p_n = 0, po_n = 0;      // (rotated) the coming chunk's point index (binned / caller order), ...
  bool valid_n = false;
  float4 l_in_n = make_float4(0.f, 1.f, 0.f, 1.f);
  if (rotate && sched.cur < sched.end) MISO_TRAIN_GATHER(sched.cur, rec_cur, p_n, po_n, valid_n, l_in_n)
  for (int64_t chunk = sched.cur; chunk < sched.end; chunk += sched.step) {
    asm volatile("" ::: "memory");      // see sdf_fwd_kernel: keeps the LDS reads of weights / biases inside the loop
    int64_t p, po;
    bool valid;
    float4 l_in;
    if (!rotate) MISO_TRAIN_GATHER(chunk, rec_cur, p, po, valid, l_in)
    else { p = p_n; po = po_n; valid = valid_n; l_in = l_in_n; }
    // ================================ forward =====================================================================
    uint32_t mw[MW];
    float p0 = 0.0f, p1 = 0.0f, poison = 0.0f;
    u32x4 maskB[H / 16][HALF ? 1 : 2];      // SPLIT: the last ReLU's mask as the first backward product's B operand
    if constexpr (SPLIT) {
      decoder_fwd_split<F, H, NH, HALF, false, true>(s_fwd, s_bias, lane, f, mw, maskB, p0, p1, poison);
    } else {
      f32x16 buf[2][RT][2];
      {
        f32x16 bias[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) bias[r][j] = b0[32 * r + row_of(j, hi)];
#pragma unroll
        for (int s = 0; s < KS0; ++s) {
          float bt0, bt1 = 0.0f;
          if (HALF) {      // both halves hold the same point: k = 0 from the low half, k = 1 from the high one
            bt0 = hi ? f[2 * s + 1] : f[2 * s];
          } else {
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(f[2 * s]), __float_as_uint(f[2 * s + 1]), false, false);
            bt0 = __uint_as_float(sw[0]); bt1 = __uint_as_float(sw[1]);
          }
#pragma unroll
          for (int r = 0; r < RT; ++r) {
            float a = w0p[(s * 64 + lane) * RT + r];
            buf[0][r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt0, s == 0 ? bias[r] : buf[0][r][0], 0, 0, 0);
            if (!HALF) buf[0][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt1, s == 0 ? bias[r] : buf[0][r][1], 0, 0, 0);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        uint32_t m = 0;
#pragma unroll
        for (int t = 0; t < (HALF ? 1 : 2); ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            buf[0][r][t][j] = relu1(buf[0][r][t][j]);
            push_gt0(m, buf[0][r][t][j]);
          }
        mw[r] = HALF ? (m << 16) : m;      // (tile 0's bits at 31..16 either way: mask_bit)
      }
#pragma unroll
      for (int h = 0; h + 1 < NH; ++h) {
        const int ci = h & 1, ni = ci ^ 1;
        f32x16 bias[RT];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) bias[r][j] = bh[h * H + 32 * r + row_of(j, hi)];
#pragma unroll
        for (int rp = 0; rp < RT; ++rp)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int ks = rp * 16 + j;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
              float a = whp[((h * KS1 + ks) * 64 + lane) * RT + r];
              buf[ni][r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][0][j], ks == 0 ? bias[r] : buf[ni][r][0], 0, 0, 0);
              if (!HALF) buf[ni][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][1][j], ks == 0 ? bias[r] : buf[ni][r][1], 0, 0, 0);
            }
          }
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          uint32_t m = 0;
#pragma unroll
          for (int t = 0; t < (HALF ? 1 : 2); ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              buf[ni][r][t][j] = relu1(buf[ni][r][t][j]);
              push_gt0(m, buf[ni][r][t][j]);
            }
          mw[(h + 1) * RT + r] = HALF ? (m << 16) : m;
        }
      }
      if (NH == 0) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            float wv = wo[32 * r + row_of(j, hi)];
            p0 += wv * buf[0][r][0][j];
            if (!HALF) p1 += wv * buf[0][r][1][j];
          }
      } else {
        constexpr int h = NH > 0 ? NH - 1 : 0, ci = h & 1;
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          f32x16 a0, a1, bias;
#pragma unroll
          for (int j = 0; j < 16; ++j) bias[j] = bh[h * H + 32 * r + row_of(j, hi)];
#pragma unroll
          for (int rp = 0; rp < RT; ++rp)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int ks = rp * 16 + j;
              float a = whp[((h * KS1 + ks) * 64 + lane) * RT + r];
              a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][0][j], ks == 0 ? bias : a0, 0, 0, 0);
              if (!HALF) a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][1][j], ks == 0 ? bias : a1, 0, 0, 0);
            }
          uint32_t m = 0, m1 = 0;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float y0 = relu1(a0[j]), y1 = HALF ? 0.0f : relu1(a1[j]);
            push_gt0(m, y0);
            if (!HALF) push_gt0(m1, y1);
            float wv = wo[32 * r + row_of(j, hi)];
            p0 += wv * y0;
            if (!HALF) p1 += wv * y1;
          }
          mw[(h + 1) * RT + r] = (m << 16) | m1;
        }
      }
    }
    p0 += __shfl_xor(p0, 32);
    if (!HALF) p1 += __shfl_xor(p1, 32);
    const float sdf_v = SPLIT ? (((!HALF && hi) ? p1 : p0) + bo) + poison : ((!HALF && hi) ? p1 : p0) + bo;
    const bool mine = !(HALF && hi);      // HALF: the high half mirrors the low one -- stored / counted once
    if (valid && sdf && mine) sdf[po] = sdf_v;
    // ================================ loss: lane = point ============================================================
    float gl = 0.0f;
    if (valid && mine) {
      float gsd, gfs;
      map_loss_one(lin.p, sdf_v, l_in.x, l_in.w, l_in.y == 1.0f, lin.p.w_fs > 0.f && l_in.z == 1.0f, gsd, gfs,
                   loss_sdf, loss_fs);
      gl = (gsd + gfs) * inv_n;
    }
    // the backward works on two tiles of 32 points, lane (hi, c) on point 32 t + c of tile t
    float ds[2];
    ds[0] = __shfl(gl, lane & 31);
    ds[1] = HALF ? 0.0f : __shfl(gl, 32 + (lane & 31));
    // ================================ backward ======================================================================
    f32x16 df[2];
    if constexpr (SPLIT) {
      decoder_bwd_split<F, H, NH, HALF>(s_bwd, lane, maskB, mw, ds, df);
    } else {
      f32x16 dbuf[2][RT][2];
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float wv = wo[32 * r + row_of(j, hi)];
#pragma unroll
          for (int t = 0; t < (HALF ? 1 : 2); ++t) dbuf[0][r][t][j] = gate(wv * ds[t], mw[NH * RT + r], t, j);
        }
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        const int h = NH - 1 - hh;
        const int ci = hh & 1, ni = ci ^ 1;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int j = 0; j < 16; ++j) { dbuf[ni][r][0][j] = 0.0f; if (!HALF) dbuf[ni][r][1][j] = 0.0f; }
#pragma unroll
        for (int rp = 0; rp < RT; ++rp)
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int ks = rp * 16 + j;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
              float a = whT[((h * KS1 + ks) * 64 + lane) * RT + r];
              dbuf[ni][r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dbuf[ci][rp][0][j], dbuf[ni][r][0], 0, 0, 0);
              if (!HALF) dbuf[ni][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, dbuf[ci][rp][1][j], dbuf[ni][r][1], 0, 0, 0);
            }
          }
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
          for (int t = 0; t < (HALF ? 1 : 2); ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) dbuf[ni][r][t][j] = gate(dbuf[ni][r][t][j], mw[h * RT + r], t, j);
      }
      f32x16 (&d)[RT][2] = dbuf[NH & 1];
#pragma unroll
      for (int j = 0; j < 16; ++j) { df[0][j] = 0.0f; if (!HALF) df[1][j] = 0.0f; }
#pragma unroll
      for (int rp = 0; rp < RT; ++rp)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          float a = w0T[(rp * 16 + j) * 64 + lane];
          df[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[rp][0][j], df[0], 0, 0, 0);
          if (!HALF) df[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, d[rp][1][j], df[1], 0, 0, 0);
        }
    }
    // ---- d-feat rows: accumulator layout -> LDS tile -> 64 contiguous rows of the (N, F) buffer, 16-B stores --------
    memory_phase(true, g.tune);
#pragma unroll
    for (int t = 0; t < (HALF ? 1 : 2); ++t)
#pragma unroll
      for (int gq = 0; gq < (F + 7) / 8; ++gq) {
        const int f0 = 8 * gq + 4 * hi;
        if (f0 < F)
          *reinterpret_cast<float4*>(dF + (32 * t + (lane & 31)) * FP + f0) =
              make_float4(df[t][4 * gq], df[t][4 * gq + 1], df[t][4 * gq + 2], df[t][4 * gq + 3]);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (!SCAT || dfeat_out) {
      float* dst = dfeat_out + chunk * PTS * F;
      const int64_t rows_left = n - chunk * PTS;
      for (int i = lane; i < PTS * F / 4; i += 64) {
        const int row = (i * 4) / F, col = (i * 4) % F;
        if (row < rows_left)
          *reinterpret_cast<float4*>(dst + row * F + col) = *reinterpret_cast<const float4*>(dF + row * FP + col);
      }
    }
    if (rotate && chunk + sched.step < sched.end) MISO_TRAIN_GATHER(chunk + sched.step, rec_nxt, p_n, po_n, valid_n, l_in_n)
    if (SCAT) {
      // the scatter of sdf_bwd_kernel: 64 / SLOTS trips, lane = (point slot, dx, channel), four (dy, dz) atomics each
      constexpr int LPR = 2 * C, SLOTS = 64 / LPR;
      const int slot = lane / LPR, dx = (lane / C) & 1, ch = lane % C;
#pragma unroll 1
      for (int pg = 0; pg < (scatter_mask ? PTS / SLOTS : 0); ++pg) {
        const int pt = pg * SLOTS + slot;
#pragma unroll
        for (int l = 0; l < L; ++l) {
          const LevelK& lv = g.lv[l];
          if (!((scatter_mask >> l) & 1u)) continue;
          const int* r = rec_cur + (pt * L + l) * REC;
          const int4 r0 = *reinterpret_cast<const int4*>(r);
          const int4 r1 = *reinterpret_cast<const int4*>(r + 4);
          const int fl = r0.y;
          if (!((fl >> dx) & 1)) continue;
          const float v = dF[pt * FP + l * C + ch];
          const float wx = dx ? __int_as_float(r0.z) : __int_as_float(r1.y);
          const float wy[2] = {__int_as_float(r1.z), __int_as_float(r0.w)};
          const float wz[2] = {__int_as_float(r1.w), __int_as_float(r1.x)};
          float* base = lv.grad + r0.x + dx * lv.sX + ch;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int dy = q & 1, dz = q >> 1;
            if (((fl >> (2 + dy)) & 1) && ((fl >> (4 + dz)) & 1)) {
#ifndef MISO_ABL_NO_SCATTER      // dev ablation (wrong results): everything of the scatter but the atomics
              atomic_add_f32(base + dy * lv.sY + dz * lv.sZ, v * ((wx * wy[dy]) * wz[dz]));
#else
              asm volatile("" ::"v"(base + dy * lv.sY + dz * lv.sZ), "v"(v * ((wx * wy[dy]) * wz[dz])));
#endif
              if (ch == 0) touch_chunk(lv, r0.x + dx * lv.sX + dy * lv.sY + dz * lv.sZ);   // C floats: one chunk
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();      // the next chunk overwrites the tile (and the records)
    memory_phase(false, g.tune, wave < NW / 2);
    if (rotate) { int* t_ = rec_cur; rec_cur = rec_nxt; rec_nxt = t_; }
  }
  // loss sums: as sdf_fwd_kernel (every block stores its pair into its own slot, the slots nobody owns are cleared)
  for (int o = 32; o > 0; o >>= 1) { loss_sdf += __shfl_down(loss_sdf, o); loss_fs += __shfl_down(loss_fs, o); }
  __syncthreads();
  if (lane == 0) { smem[2 * wave] = loss_sdf; smem[2 * wave + 1] = loss_fs; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = (smem[0] + smem[2]) + (smem[4] + smem[6]), b = (smem[1] + smem[3]) + (smem[5] + smem[7]);
    if (NW == 8) {
      a += (smem[8] + smem[10]) + (smem[12] + smem[14]);
      b += (smem[9] + smem[11]) + (smem[13] + smem[15]);
    }
    float2* slots = reinterpret_cast<float2*>(lin.loss_out);
    slots[blockIdx.x] = make_float2(lin.p.w_sdf * a * inv_n, lin.p.w_fs * b * inv_n);
    for (int sl = blockIdx.x + gridDim.x; sl < MISO_LOSS_SLOTS; sl += gridDim.x) slots[sl] = make_float2(0.f, 0.f);
  }
}

#endif
// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
static hipError_t allow_lds(const void* k, size_t lds) { return allow_dynamic_lds(k, lds); }

// Decoder arithmetic of a launch: bf16x3 split products (default) or the exact fp32 chains (MISO_F_EXACT_F32 in the grid's
// flags; MISO_EXACT_F32=1 in the environment forces it for a whole process -- dev A/B)
static bool use_split(const GridK& g) {
  static const bool env_exact = [] { const char* e = getenv("MISO_EXACT_F32"); return e && atoi(e) != 0; }();
  return !(g.flags & MISO_F_EXACT_F32) && !env_exact;
}

#ifndef MISO_SDF_TRAIN_TU      // (this part is compiled into sdf_fused.o; sdf_train.o holds the training kernel: Makefile)
template <int C, int L, int H, int NH>
static hipError_t launch_fwd_t(const GridK& g, const float* packed, const float* x, int64_t n,
                               float* sdf, uint32_t* mask, const int* perm, const LossInK& lin, hipStream_t s) {
  PackLayout pl(C * L, H, NH);
  const bool split = use_split(g);
  size_t lds = (size_t)(split ? pl.s_fwd_end - pl.s_w0 + (pl.n_bias() + 3) / 4 * 4 : (pl.fwd_end + 3) / 4 * 4) * sizeof(float);
  int64_t nchunks = (n + 63) / 64;
  unsigned blocks = (unsigned)((nchunks + 3) / 4);
  if (blocks > 512u) blocks = 512u;      // persistent: two workgroups per CU
  if (lin.p.loss_type && blocks > MISO_LOSS_SLOTS) blocks = MISO_LOSS_SLOTS;   // one loss slot per block
  auto k = split ? sdf_fwd_kernel<C, L, H, NH, true> : sdf_fwd_kernel<C, L, H, NH, false>;
  hipError_t e = allow_lds((const void*)k, lds);
  if (e != hipSuccess) return e;
  k<<<blocks, 256, lds, s>>>(g, packed, x, n, sdf, mask, perm, lin);
  return hipGetLastError();
}

template <int C, int L, int H, int NH>
static hipError_t launch_bwd_t(const GridK& g, const float* packed, const float* x, int64_t n,
                               const float* gsdf, const uint32_t* mask, float* gx, bool want_grid,
                               const int* perm, float* dfeat_out, uint32_t defer_mask, bool gsdf_sorted,
                               hipStream_t s) {
  PackLayout pl(C * L, H, NH);
  constexpr int F = C * L, FP = ((F + 3) / 4) * 4 + 4, WAVE_LDS = 64 * FP + 64 * L * 8;
  bool lean = want_grid && !gx && dfeat_out != nullptr;      // every gradient level deferred to the pull?
  for (int l = 0; l < L && lean; ++l)
    if (g.lv[l].grad && !((g.ignore_mask >> l) & 1u) && !((defer_mask >> l) & 1u)) lean = false;
  const bool split = use_split(g);
  const int nb = split ? pl.total_all - pl.s_bfirst : pl.total - pl.o_whT;
  size_t lds = (size_t)(((nb + H + 3) / 4) * 4 + (want_grid ? 4 * (lean ? 64 * FP : WAVE_LDS) : 0)) * sizeof(float);
  int64_t nchunks = (n + 63) / 64;
  unsigned blocks = (unsigned)((nchunks + 3) / 4);
  const unsigned use_cap = lean ? 768u : 512u;
  if (blocks > use_cap) blocks = use_cap;
  int debug = 0;
  if (const char* d = getenv("MISO_DEBUG_BWD")) debug = atoi(d) & ~(16 | 32);
  if (gsdf_sorted) debug |= 16;
  if (lean) debug |= 32;
  void (*k)(GridK, const float*, const float*, int64_t, const float*, const uint32_t*, float*, const int*, int,
            float*, uint32_t) =
      split ? ((want_grid && gx) ? sdf_bwd_kernel<C, L, H, NH, true, true, true>
               : want_grid       ? sdf_bwd_kernel<C, L, H, NH, true, false, true>
                                 : sdf_bwd_kernel<C, L, H, NH, false, true, true>)
            : ((want_grid && gx) ? sdf_bwd_kernel<C, L, H, NH, true, true, false>
               : want_grid       ? sdf_bwd_kernel<C, L, H, NH, true, false, false>
                                 : sdf_bwd_kernel<C, L, H, NH, false, true, false>);
  hipError_t e = allow_lds((const void*)k, lds);
  if (e != hipSuccess) return e;
  k<<<blocks, 256, lds, s>>>(g, packed, x, n, gsdf, mask, gx, perm, debug, dfeat_out, defer_mask);
  return hipGetLastError();
}


#endif
#ifdef MISO_SDF_TRAIN_TU       // (compiled into sdf_train.o, under the max-ILP machine scheduler: Makefile)
template <int C, int L, int H, int NH>
static hipError_t launch_train_t(const GridK& g, const float* packed, const float* x, int64_t n, float* sdf,
                                 const int* perm, const LossInK& lin, float* dfeat_out, uint32_t defer_mask, bool scat,
                                 hipStream_t s) {
  PackLayout pl(C * L, H, NH);
  constexpr int F = C * L, FP = ((F + 3) / 4) * 4 + 4;
  const bool split = use_split(g);
  const int n_pack = split ? pl.total_all - pl.s_w0 + ((pl.n_bias() + 3) / 4) * 4 : ((pl.total + 3) / 4) * 4;
  size_t lds = (size_t)(n_pack + 4 * (64 * FP + (scat ? 64 * L * 8 : 0))) * sizeof(float);
  GridK gr = g;      // (the scattering forms: with room for a second block of cell records the kernel rotates its loop)
  gr.tune &= ~MISO_TUNE_ROTATE;
  static const bool no_rotate = getenv("MISO_TRAIN_NO_ROTATE") != nullptr;      // dev A/B
  if (scat && !no_rotate && C * L <= MISO_ROTATE_MAX_F && lds + (size_t)4 * 64 * L * 8 * sizeof(float) <= (size_t)MISO_LDS_LIMIT) {
    lds += (size_t)4 * 64 * L * 8 * sizeof(float);
    gr.tune |= MISO_TUNE_ROTATE;
  }
  int64_t nchunks = (n + 63) / 64;
  // Nothing scattered from the kernel (the mapping step): ONE workgroup of eight wavefronts per CU instead of two of
  // four -- the same two wavefronts per SIMD, half the copies of the 48 KB pack out of L2 at the start of the launch,
  // one barrier per CU (cfg-2: 72.9 -> 72.0 us, A/B in one process; MISO_TRAIN_NW4 keeps the four-wavefront form).
  // The scattering variant keeps four where its cell records would not fit beside eight d-feat tiles (wide feature
  // rows); for narrow ones see below.
  static const bool nw8 = getenv("MISO_TRAIN_NW4") == nullptr;
  if (nw8 && !scat) {
    size_t lds8 = (size_t)(n_pack + 8 * 64 * FP) * sizeof(float);
    unsigned blocks8 = (unsigned)((nchunks + 7) / 8);
    if (blocks8 > 256u) blocks8 = 256u;
    auto k8 = split ? sdf_train_kernel<C, L, H, NH, false, 8, false, true> : sdf_train_kernel<C, L, H, NH, false, 8, false, false>;
    hipError_t e8 = allow_lds((const void*)k8, lds8);
    if (e8 != hipSuccess) return e8;
    k8<<<blocks8, 512, lds8, s>>>(g, packed, x, n, sdf, perm, lin, dfeat_out, defer_mask);
    return hipGetLastError();
  }
  // Scattering, narrow feature rows (cfg-3's C = 4, L = 2): eight wavefronts and their two record blocks each fit beside the
  // pack, and the four-wavefront form is ONE workgroup per CU there (the bf16x3 pack is 67 KB) -- one wavefront per SIMD.
  // Binned batches only (the unbinned small-batch forms below stay as they are).
  if constexpr (C * L <= MISO_ROTATE_MAX_F) {
    static const bool nw8s = getenv("MISO_TRAIN_SCAT_NW4") == nullptr;      // dev A/B
    const size_t lds8 = (size_t)(n_pack + 8 * (64 * FP + 2 * 64 * L * 8)) * sizeof(float);
    if (scat && nw8s && !no_rotate && (perm || (g.flags & MISO_F_INDEX_IN_XN)) && lds8 <= (size_t)MISO_LDS_LIMIT) {
      unsigned blocks8 = (unsigned)((nchunks + 7) / 8);
      if (blocks8 > 256u) blocks8 = 256u;
      auto k8 = split ? sdf_train_kernel<C, L, H, NH, true, 8, false, true> : sdf_train_kernel<C, L, H, NH, true, 8, false, false>;
      hipError_t e8 = allow_lds((const void*)k8, lds8);
      if (e8 != hipSuccess) return e8;
      gr.tune |= MISO_TUNE_ROTATE;
      k8<<<blocks8, 512, lds8, s>>>(gr, packed, x, n, sdf, perm, lin, dfeat_out, defer_mask);
      return hipGetLastError();
    }
  }
  // an unbinned batch of at most one 64-point chunk per SIMD (<= 65 536 samples): 32-point trips -- the batch is latency, not
  // throughput, and half the matrix chain per wavefront on twice the wavefronts is what shortens it (MISO_TRAIN_NO_HALF: dev)
  static const bool no_half = getenv("MISO_TRAIN_NO_HALF") != nullptr;
  if (scat && !perm && !dfeat_out && nchunks <= 1024 && !no_half && !(g.flags & MISO_F_FULL_TRIPS)) {
    const int64_t nhalf = (n + 31) / 32;
    unsigned bh_ = (unsigned)((nhalf + 3) / 4);
    if (bh_ > MISO_LOSS_SLOTS) bh_ = MISO_LOSS_SLOTS;
    auto kh = split ? sdf_train_kernel<C, L, H, NH, true, 4, true, true> : sdf_train_kernel<C, L, H, NH, true, 4, true, false>;
    hipError_t eh = allow_lds((const void*)kh, lds);
    if (eh != hipSuccess) return eh;
    kh<<<bh_, 256, lds, s>>>(gr, packed, x, n, sdf, perm, lin, dfeat_out, defer_mask);
    return hipGetLastError();
  }
  unsigned blocks = (unsigned)((nchunks + 3) / 4);
  if (blocks > 512u) blocks = 512u;      // persistent: two workgroups per CU (256 .. 1024 measured: 512 and up equal)
  if (blocks > MISO_LOSS_SLOTS) blocks = MISO_LOSS_SLOTS;   // one loss slot per block
  auto k = split ? (scat ? sdf_train_kernel<C, L, H, NH, true, 4, false, true> : sdf_train_kernel<C, L, H, NH, false, 4, false, true>)
                 : (scat ? sdf_train_kernel<C, L, H, NH, true, 4, false, false> : sdf_train_kernel<C, L, H, NH, false, 4, false, false>);
  hipError_t e = allow_lds((const void*)k, lds);
  if (e != hipSuccess) return e;
  k<<<blocks, 256, lds, s>>>(gr, packed, x, n, sdf, perm, lin, dfeat_out, defer_mask);
  return hipGetLastError();
}

#endif
#define MISO_FUSED_SHAPES(X) \
  X(4, 1, 32, 1) X(4, 1, 64, 1) X(4, 2, 32, 1) X(4, 2, 64, 1) X(4, 3, 64, 1) X(4, 4, 64, 1) \
  X(8, 1, 64, 1) X(8, 2, 64, 1) X(8, 3, 64, 1) X(8, 4, 64, 1) X(8, 3, 32, 1)

#ifndef MISO_SDF_TRAIN_TU      // (this part is compiled into sdf_fused.o; sdf_train.o holds the training kernel: Makefile)
bool fused_shape_supported(int C, int L, int H, int NH) {
#define X(c, l, h, nh) if (C == c && L == l && H == h && NH == nh) return true;
  MISO_FUSED_SHAPES(X)
#undef X
  return false;
}

hipError_t launch_sdf_fwd(int C, int L, int H, int NH, const GridK& g, const float* packed,
                          const float* x, int64_t n, float* sdf, uint32_t* mask, const int* perm,
                          const LossInK& lin, hipStream_t s) {
  if (n == 0) return hipSuccess;
#define X(c, l, h, nh) \
  if (C == c && L == l && H == h && NH == nh) \
    return launch_fwd_t<c, l, h, nh>(g, packed, x, n, sdf, mask, perm, lin, s);
  MISO_FUSED_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t launch_sdf_bwd(int C, int L, int H, int NH, const GridK& g, const float* packed,
                          const float* x, int64_t n, const float* gsdf, const uint32_t* mask,
                          float* gx, bool want_grid, const int* perm, float* dfeat_out,
                          uint32_t defer_mask, bool gsdf_sorted, hipStream_t s) {
  if (n == 0) return hipSuccess;
#define X(c, l, h, nh) \
  if (C == c && L == l && H == h && NH == nh) \
    return launch_bwd_t<c, l, h, nh>(g, packed, x, n, gsdf, mask, gx, want_grid, perm, dfeat_out, defer_mask, gsdf_sorted, s);
  MISO_FUSED_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}


#endif
#ifdef MISO_SDF_TRAIN_TU       // (compiled into sdf_train.o, under the max-ILP machine scheduler: Makefile)
// forward + mapping loss + decoder backward of a batch in one launch (sdf_train_kernel): d-feat rows for the levels in
// defer_mask (formed by the pull / push afterwards), float atomics for the other levels with a gradient (scat: there
// are such levels), loss slots; sdf (caller order) optional; perm == nullptr: an unbinned batch
hipError_t launch_sdf_train(int C, int L, int H, int NH, const GridK& g, const float* packed, const float* x,
                            int64_t n, float* sdf, const int* perm, const LossInK& lin, float* dfeat_out,
                            uint32_t defer_mask, bool scat, hipStream_t s) {
  if (n == 0) return hipSuccess;
#define X(c, l, h, nh) \
  if (C == c && L == l && H == h && NH == nh) \
    return launch_train_t<c, l, h, nh>(g, packed, x, n, sdf, perm, lin, dfeat_out, defer_mask, scat, s);
  MISO_FUSED_SHAPES(X)
#undef X
  return hipErrorInvalidValue;
}

#endif
#ifndef MISO_SDF_TRAIN_TU      // (this part is compiled into sdf_fused.o; sdf_train.o holds the training kernel: Makefile)
hipError_t launch_mlp_pack(const MlpK& m, int F, int H, int NH, float* out, hipStream_t s) {
  PackLayout pl(F, H, NH);
  mlp_pack_kernel<<<(pl.total_all + 255) / 256, 256, 0, s>>>(m, F, H, NH, out);
  return hipGetLastError();
}

int64_t mlp_packed_floats(int F, int H, int NH) { return PackLayout(F, H, NH).total_all; }

// dynamic LDS of the sdf_train_kernel launch for this shape (launch_train_t above): the pack, then per wavefront a d-feat
// tile [64][FP] and, scattering, its cell records [64][L][8] -- four wavefronts, or eight when nothing is scattered
int64_t sdf_train_lds_bytes(int C, int L, int H, int NH, bool scat) {
  if (!fused_shape_supported(C, L, H, NH)) return 0;
  const PackLayout pl(C * L, H, NH);
  const int F = C * L, FP = ((F + 3) / 4) * 4 + 4;
  // the larger of the two decoder forms (the split pack is the larger one for every covered shape)
  const int64_t pack_exact = ((pl.total + 3) / 4) * 4, pack_split = pl.total_all - pl.s_w0 + ((pl.n_bias() + 3) / 4) * 4;
  const int64_t pack = pack_split > pack_exact ? pack_split : pack_exact;
  int64_t four = pack + 4 * (64 * FP + (scat ? 64 * L * 8 : 0));
  const int64_t eight = pack + 8 * 64 * FP;
  // (scattering: a second block of cell records where it fits, launch_train_t)
  if (scat && F <= MISO_ROTATE_MAX_F && (four + 4 * 64 * L * 8) * (int64_t)sizeof(float) <= (int64_t)MISO_LDS_LIMIT)
    four += 4 * 64 * L * 8;
  // (and its eight-wavefront form where that fits)
  const int64_t eight_s = pack + 8 * (64 * FP + 2 * 64 * L * 8);
  if (scat && F <= MISO_ROTATE_MAX_F && eight_s * (int64_t)sizeof(float) <= (int64_t)MISO_LDS_LIMIT && eight_s > four) four = eight_s;
  return (int64_t)sizeof(float) * (scat ? four : (eight > four ? eight : four));
}

#endif
}  // namespace miso
