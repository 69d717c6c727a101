// Frozen-decoder MLP (MLPNet: grid_opt/models/modules.py:11-32) on the matrix cores: the packed-weight layout shared by
// mlp_pack_kernel and every kernel that decodes (sdf_fused.hip, atlas.hip), the register-level helpers of the ReLU / gate
// passes, and the split-precision (bf16x3, mlp_split.hpp) forward and backward chains.
#pragma once
#include "common.hpp"
#include "mlp_split.hpp"

namespace miso {

// On gfx950 the fp32 MFMA shares the vector FMA datapath (tools/ubench/mfma_valu.hip): every VALU
// instruction between two MFMAs costs matrix throughput, so the MLP phase is trimmed of them.
// * ReLU on the raw bits: max_i32(bits, 0) is ONE instruction and exact (negative floats, -0 included,
//   are negative integers); fmaxf(x, 0) costs two (a canonicalising v_max x,x first).
// * Sign bit for the backward from the ReLU output: min_u32(bits, 1) then shift-or -- two instructions
//   instead of compare + select + or.
// * The bias enters as the C operand of the first MFMA of each chain (no accumulator init moves).
// (Inline-asm variants were tried: the hazard recogniser does not see that the asm reads MFMA results,
//  and the missing wait states returned stale accumulators.)
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }
// y = relu1(x): bit <- (x > 0)
// The bits are SHIFTED IN (the k-th of 32 pushes ends at bit 31 - k): y's bits are a non-negative integer, so
// bit 31 of y + 0x7fffffff is (y != 0) and v_alignbit(m, that, 31) = (m << 1) | bit -- two instructions.  (min(y, 1)
// << k | m is canonicalised by the compiler into compare + select + or, with a wait state after every compare.)
__device__ __forceinline__ void push_gt0(uint32_t& m, float y) {
  m = __builtin_amdgcn_alignbit(m, __float_as_uint(y) + 0x7fffffffu, 31);
}
__device__ __forceinline__ bool mask_bit(uint32_t m, int t, int j) { return (m >> (31 - (t * 16 + j))) & 1u; }
// mask_bit ? x : 0 as v_bfe_i32 (the bit, sign-extended: 0 or ~0) + v_and -- the select form costs and + compare +
// select and a wait state per element
__device__ __forceinline__ float gate(float x, uint32_t m, int t, int j) {
  int e = __builtin_amdgcn_sbfe((int)m, 31 - (t * 16 + j), 1);
  asm("" : "+v"(e));      // opaque: and(x, sext(bit)) would be folded back into compare + select
  return __uint_as_float(__float_as_uint(x) & (uint32_t)e);
}

// lane-per-point trilinear gather of one level (channels-last rows of C floats, 16-B loads): f[0..C) = sum_k w_k v_k
template <int C>
__device__ __forceinline__ void gather_level(const LevelK& lv, const Cell& c, float* f) {
  // lane-per-point gather of one level: 8 corners x C channels, channels-last.
#pragma unroll
  for (int q = 0; q < C; ++q) f[q] = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
    bool in = c.inx[dx] && c.iny[dy] && c.inz[dz];
    float w = in ? (c.wx[dx] * c.wy[dy]) * c.wz[dz] : 0.0f;
    int off = in ? (c.k0 + dz) * lv.sZ + (c.j0 + dy) * lv.sY + (c.i0 + dx) * lv.sX : 0;
#ifdef MISO_ABL_UNIFORM_GATHER      // dev ablation (wrong results): every lane reads lane 0's rows -- perfectly coalesced gathers
    off = __builtin_amdgcn_readfirstlane(off);
#endif
#ifdef MISO_ABL_QUAD_GATHER         // dev ablation (wrong results): groups of four lanes read one lane's rows
    off = __builtin_amdgcn_mov_dpp(off, 0x00, 0xf, 0xf, true);
#endif
#pragma unroll
    for (int q = 0; q < C; q += 4) {
      float4 v = *reinterpret_cast<const float4*>(lv.data + off + q);
      f[q + 0] += v.x * w; f[q + 1] += v.y * w; f[q + 2] += v.z * w; f[q + 3] += v.w * w;
    }
  }
}


// ---------------------------------------------------------------------------
// Packed decoder layout (floats).  RT = H/32 row tiles, KS0 = ceil(F/2),
// KS1 = H/2 k-steps for an HxH layer, NH hidden (HxH) layers.
//   fwd: W0p [KS0][64][RT]   A(l) = W0[32r + (l&31)][2s + (l>>5)]
//        Whp [NH][KS1][64][RT] A(l) = Wh[32r + (l&31)][32rp + row_of(j, l>>5)], ks = 16rp + j
//        b0 [H], bh [NH][H], wo [H], bo [4]
//   bwd: WhTp [NH][KS1][64][RT] A(l) = Wh[32rp + row_of(j, l>>5)][32r + (l&31)]
//        W0Tp [KS1][64]         A(l) = W0[32rp + row_of(j, l>>5)][l&31]   (0 for l&31 >= F)
// Behind it (round 6), the same matrices as bf16x3 pieces for v_mfma_f32_32x32x16_bf16 (mlp_split.hpp; dword offsets,
// one block [kb][r][piece][64 lanes][4 dwords] per matrix, KB0 = ceil(F/16) and KBH = H/16 k-blocks):
//   forward:  s_w0  W0 rows 32r + (l&31), contraction split_k_feat;  s_wh [NH] Wh rows 32r + (l&31), contraction split_k_acc
//   backward: s_bfirst  the FIRST backward product with the output weights folded in -- its B operand is the last ReLU's
//                       0 / 1 mask, exact in one bf16 piece (three matrix instructions per k-block instead of six, nothing to
//                       split): NH >= 1: V[k][m] = fl(Wh[NH-1][m][k] wo[m]) (RT row tiles); NH == 0: U[f][m] = fl(W0[m][f] wo[m])
//             s_whT [NH-1]  Wh[h]^T, h = 0 .. NH-2;   s_w0T  W0^T (one row tile; NH >= 1 only)
// ---------------------------------------------------------------------------
struct PackLayout {
  int F, H, NH, RT, KS0, KS1;
  int o_w0, o_wh, o_b0, o_bh, o_wo, o_bo, fwd_end;
  int o_whT, o_w0T, total;
  int KB0, KBH;
  int s_w0, s_wh, s_fwd_end, s_bfirst, s_whT, s_w0T, total_all;
  __host__ __device__ PackLayout(int F_, int H_, int NH_) {
    F = F_; H = H_; NH = NH_; RT = H / 32; KS0 = (F + 1) / 2; KS1 = H / 2;
    int o = 0;
    o_w0 = o; o += KS0 * 64 * RT;
    o_wh = o; o += NH * KS1 * 64 * RT;
    o_b0 = o; o += H;
    o_bh = o; o += NH * H;
    o_wo = o; o += H;
    o_bo = o; o += 4;
    fwd_end = o;
    o_whT = o; o += NH * KS1 * 64 * RT;
    o_w0T = o; o += KS1 * 64;
    total = o;
    KB0 = (F + 15) / 16; KBH = H / 16;
    s_w0 = o; o += split_matrix_dwords(KB0, RT);
    s_wh = o; o += NH * split_matrix_dwords(KBH, RT);
    s_fwd_end = o;
    s_bfirst = o; o += split_matrix_dwords(KBH, NH >= 1 ? RT : 1);
    s_whT = o; o += (NH > 1 ? NH - 1 : 0) * split_matrix_dwords(KBH, RT);
    s_w0T = o; o += NH >= 1 ? split_matrix_dwords(KBH, 1) : 0;
    total_all = o;
  }
  __host__ __device__ int n_bias() const { return fwd_end - o_b0; }      // b0, bh, wo, bo: contiguous
};

// ---------------------------------------------------------------------------------------------------------------------
// Split-precision chains.  NT = 1 (HALF: 32-point trips) or 2 point tiles per wavefront.
// One output row tile at a time (r outermost): its accumulators complete while the next row tile's products are still
// being issued, so the ReLU / split pass of row tile r runs beside the matrix instructions of row tile r + 1.
template <int KB, int NT>
__device__ __forceinline__ void mma_split_row(const uint32_t* __restrict__ A, int r, int RT, int lane,
                                              const Split3 (&B)[KB][NT], f32x16 (&acc)[NT], const f32x16& init) {
  constexpr int QA[6] = {2, 1, 0, 1, 0, 0};      // (piece of A, piece of B) in the order of accumulation: smallest first
  constexpr int QB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    u32x4 a[3];
#pragma unroll
#ifdef MISO_ABL_NO_AREAD     // dev ablation (wrong results): no LDS reads of the weights
    for (int q = 0; q < 3; ++q) a[q] = u32x4{0x3f803f80u + (unsigned)lane, 0x3f803f80u, 0x3e803f80u, 0x3f803f00u + (unsigned)(kb + q)};
#else
    for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(A + split_a_dword(kb, r, q, lane, RT));
#endif
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        acc[t] = mfma_bf16(a[QA[c]], B[kb][t].q[QB[c]], (kb == 0 && c == 0) ? init : acc[t]);
  }
}
// B exact in one piece (a 0 / 1 mask): the three pieces of A against it
template <int KB, int NT>
__device__ __forceinline__ void mma_mask_row(const uint32_t* __restrict__ A, int r, int RT, int lane,
                                             const u32x4 (&B)[KB][NT], f32x16 (&acc)[NT], const f32x16& init) {
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    u32x4 a[3];
#pragma unroll
#ifdef MISO_ABL_NO_AREAD
    for (int q = 0; q < 3; ++q) a[q] = u32x4{0x3f803f80u + (unsigned)lane, 0x3f803f80u, 0x3e803f80u, 0x3f803f00u + (unsigned)(kb + q)};
#else
    for (int q = 0; q < 3; ++q) a[q] = *reinterpret_cast<const u32x4*>(A + split_a_dword(kb, r, q, lane, RT));
#endif
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int t = 0; t < NT; ++t) acc[t] = mfma_bf16(a[2 - c], B[kb][t], (kb == 0 && c == 0) ? init : acc[t]);
  }
}

__device__ __forceinline__ f32x16 bias_block(const float* __restrict__ b, int r, int hi) {
  f32x16 v;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = b[32 * r + row_of(j, hi)];
  return v;
}
__device__ __forceinline__ f32x16 zero_block() {
  f32x16 v;
#pragma unroll
  for (int j = 0; j < 16; ++j) v[j] = 0.0f;
  return v;
}

// bf16 1.0 / 0.0 of (y > 0) for a ReLU output y (bits of a non-negative float; 0x3f80 as an integer is below every
// normal float's pattern), two elements per dword
__device__ __forceinline__ uint32_t mask_pair_bf16(float y0, float y1) {
  const uint32_t a = min(__float_as_uint(y0), 0x3f80u), b = min(__float_as_uint(y1), 0x3f80u);
  return a | (b << 16);
}
// the same dword from two adjacent sign bits of a mask word (bit position of element j: 31 - (16 t + j))
__device__ __forceinline__ uint32_t mask_pair_from_bits(uint32_t m, int t, int j) {
  const uint32_t b0 = (m >> (31 - (t * 16 + j))) & 1u, b1 = (m >> (31 - (t * 16 + j + 1))) & 1u;
  return (b0 ? 0x3f80u : 0u) | (b1 ? 0x3f800000u : 0u);
}

// Forward: lane-per-point feature rows f[0..F) -> partial output sums p0 (tile 0), p1 (tile 1) of this lane's 16 rows per row
// tile (the caller adds the two lane halves and the output bias).  mw: ReLU sign bits per layer and row tile (the layout
// of the exact kernels: word l*RT + r, tile t's element j at bit 31 - (16 t + j)); BITS_LAST = false leaves the last
// layer's words unwritten and MASKB = true returns the last ReLU's mask as the B operand of the first backward product.
//   sw: LDS, the pack's [s_w0, s_fwd_end) block;  bias: LDS, the pack's [o_b0, fwd_end) block (b0, bh, wo, bo)
template <int F, int H, int NH, bool HALF, bool BITS_LAST, bool MASKB, int FN>
__device__ __forceinline__ void decoder_fwd_split(const uint32_t* __restrict__ sw, const float* __restrict__ bias, int lane,
                                                  const float (&f)[FN], uint32_t (&mw)[(NH + 1) * (H / 32)],
                                                  u32x4 (&maskB)[H / 16][HALF ? 1 : 2], float& p0, float& p1,
                                                  float& poison) {
  constexpr int RT = H / 32, KB0 = (F + 15) / 16, KBH = H / 16, NT = HALF ? 1 : 2;
  const int hi = lane >> 5;
  // A NaN among the features must come out as a NaN SDF (the trainers' NaN guards rely on it: trainer.py:205-211).  The
  // exact chains propagate it through their FMAs; here a NaN piece can leave the matrix core with its sign set, and the
  // integer ReLU would clear it.  So: 0 for a finite feature row, NaN otherwise, added to this lane's SDF by the caller.
  {
    float acc = f[0];
#pragma unroll
    for (int i = 1; i < F; ++i) acc += f[i];
    poison = acc - acc;
  }
  const float* b0 = bias;
  const float* bh = bias + H;
  const float* wo = bias + H + NH * H;
  // ---- feature rows -> B operand of layer 0: element i of k-block kb, lane half hi <-> feature 16 kb + 8 hi + i ---------
  Split3 B0[KB0][NT];
#pragma unroll
  for (int kb = 0; kb < KB0; ++kb) {
    auto fe = [&](int i) -> float { return (16 * kb + i < F) ? f[(16 * kb + i < FN) ? 16 * kb + i : 0] : 0.0f; };
    const Split3 lo = split8(fe(0), fe(1), fe(2), fe(3), fe(4), fe(5), fe(6), fe(7));
    Split3 up;
    if (16 * kb + 8 < F) up = split8(fe(8), fe(9), fe(10), fe(11), fe(12), fe(13), fe(14), fe(15));
    else {
#pragma unroll
      for (int q = 0; q < 3; ++q) up.q[q] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (HALF) {      // both lane halves hold the same point
          B0[kb][0].q[q][d] = hi ? up.q[q][d] : lo.q[q][d];
        } else {         // tile 0: (own low | low half's upper), tile 1: (high half's low | own upper)
          auto sw2 = __builtin_amdgcn_permlane32_swap(lo.q[q][d], up.q[q][d], false, false);
          B0[kb][0].q[q][d] = sw2[0];
          B0[kb][1].q[q][d] = sw2[1];
        }
      }
  }
  p0 = 0.0f; p1 = 0.0f;
  if constexpr (NH == 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f32x16 acc[NT];
      mma_split_row<KB0, NT>(sw, r, RT, lane, B0, acc, bias_block(b0, r, hi));
      uint32_t m = 0, m1 = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float y0 = relu1(acc[0][j]), y1 = HALF ? 0.0f : relu1(acc[NT - 1][j]);
        if (BITS_LAST) { push_gt0(m, y0); if (!HALF) push_gt0(m1, y1); }
        const float wv = wo[32 * r + row_of(j, hi)];
        p0 += wv * y0;
        if (!HALF) p1 += wv * y1;
        if (MASKB && (j & 1)) {
          maskB[2 * r + (j >> 3)][0][(j & 7) >> 1] = mask_pair_bf16(relu1(acc[0][j - 1]), y0);
          if (!HALF) maskB[2 * r + (j >> 3)][NT - 1][(j & 7) >> 1] = mask_pair_bf16(relu1(acc[NT - 1][j - 1]), y1);
        }
      }
      if (BITS_LAST) mw[r] = (m << 16) | m1;
    }
  } else {
    // ---- layer 0 -------------------------------------------------------------------------------------------------
    Split3 Bh[KBH][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f32x16 acc[NT];
      mma_split_row<KB0, NT>(sw, r, RT, lane, B0, acc, bias_block(b0, r, hi));
      uint32_t m = 0;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc[t][j] = relu1(acc[t][j]); push_gt0(m, acc[t][j]); }
      mw[r] = HALF ? (m << 16) : m;
#pragma unroll
      for (int t = 0; t < NT; ++t) { Bh[2 * r][t] = split_acc<0>(acc[t]); Bh[2 * r + 1][t] = split_acc<1>(acc[t]); }
    }
    const uint32_t* swh = sw + split_matrix_dwords(KB0, RT);
    // ---- hidden layers but the last ----------------------------------------------------------------------------------
#pragma unroll
    for (int h = 0; h + 1 < NH; ++h) {
      Split3 Bn[KBH][NT];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f32x16 acc[NT];
        mma_split_row<KBH, NT>(swh + h * split_matrix_dwords(KBH, RT), r, RT, lane, Bh, acc, bias_block(bh + h * H, r, hi));
        uint32_t m = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j) { acc[t][j] = relu1(acc[t][j]); push_gt0(m, acc[t][j]); }
        mw[(h + 1) * RT + r] = HALF ? (m << 16) : m;
#pragma unroll
        for (int t = 0; t < NT; ++t) { Bn[2 * r][t] = split_acc<0>(acc[t]); Bn[2 * r + 1][t] = split_acc<1>(acc[t]); }
      }
#pragma unroll
      for (int kb = 0; kb < KBH; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) Bh[kb][t] = Bn[kb][t];
    }
    // ---- last hidden layer + output layer (out_dim = 1): reduced row tile by row tile -----------------------------------
    constexpr int h = NH - 1;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f32x16 acc[NT];
      mma_split_row<KBH, NT>(swh + h * split_matrix_dwords(KBH, RT), r, RT, lane, Bh, acc, bias_block(bh + h * H, r, hi));
      uint32_t m = 0, m1 = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float y0 = relu1(acc[0][j]), y1 = HALF ? 0.0f : relu1(acc[NT - 1][j]);
        if (BITS_LAST) { push_gt0(m, y0); if (!HALF) push_gt0(m1, y1); }
        const float wv = wo[32 * r + row_of(j, hi)];
        p0 += wv * y0;
        if (!HALF) p1 += wv * y1;
        if (MASKB && (j & 1)) {
          maskB[2 * r + (j >> 3)][0][(j & 7) >> 1] = mask_pair_bf16(relu1(acc[0][j - 1]), y0);
          if (!HALF) maskB[2 * r + (j >> 3)][NT - 1][(j & 7) >> 1] = mask_pair_bf16(relu1(acc[NT - 1][j - 1]), y1);
        }
      }
      if (BITS_LAST) mw[(h + 1) * RT + r] = (m << 16) | m1;
    }
  }
}

// the last ReLU's mask as a B operand, from its sign-bit words (the backward kernel; the training kernel forms it directly)
template <int H, int NH, bool HALF>
__device__ __forceinline__ void mask_operand_from_bits(const uint32_t (&mw)[(NH + 1) * (H / 32)],
                                                       u32x4 (&maskB)[H / 16][HALF ? 1 : 2]) {
  constexpr int RT = H / 32, NT = HALF ? 1 : 2;
#pragma unroll
  for (int r = 0; r < RT; ++r)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; j += 2)
        maskB[2 * r + (j >> 3)][t][(j & 7) >> 1] = mask_pair_from_bits(mw[NH * RT + r], t, j);
}

// Backward: d sdf of the wavefront's two point tiles (ds[t]: lane (hi, c) <-> point 32 t + c) -> d feats in accumulator
// layout df[t] (row = feature row_of(j, hi), see sdf_bwd_kernel).  sb: LDS, the pack's [s_bfirst, total_all) block.
template <int F, int H, int NH, bool HALF>
__device__ __forceinline__ void decoder_bwd_split(const uint32_t* __restrict__ sb, int lane,
                                                  const u32x4 (&maskB)[H / 16][HALF ? 1 : 2],
                                                  const uint32_t (&mw)[(NH + 1) * (H / 32)], const float (&ds)[2],
                                                  f32x16 (&df)[2]) {
  constexpr int RT = H / 32, KBH = H / 16, NT = HALF ? 1 : 2;
  if constexpr (NH == 0) {
    f32x16 acc[NT];
    mma_mask_row<KBH, NT>(sb, 0, 1, lane, maskB, acc, zero_block());
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) df[t][j] = acc[t][j] * ds[t];
  } else {
    Split3 Bd[KBH][NT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      f32x16 acc[NT];
      mma_mask_row<KBH, NT>(sb, r, RT, lane, maskB, acc, zero_block());
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[t][j] = gate(acc[t][j], mw[(NH - 1) * RT + r], t, j);
#pragma unroll
      for (int t = 0; t < NT; ++t) { Bd[2 * r][t] = split_acc<0>(acc[t]); Bd[2 * r + 1][t] = split_acc<1>(acc[t]); }
    }
    const uint32_t* swhT = sb + split_matrix_dwords(KBH, RT);
#pragma unroll
    for (int hh = 1; hh < NH; ++hh) {
      const int h = NH - 1 - hh;      // the product with Wh[h]^T, gated by layer h's ReLU
      Split3 Bn[KBH][NT];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        f32x16 acc[NT];
        mma_split_row<KBH, NT>(swhT + h * split_matrix_dwords(KBH, RT), r, RT, lane, Bd, acc, zero_block());
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[t][j] = gate(acc[t][j], mw[h * RT + r], t, j);
#pragma unroll
        for (int t = 0; t < NT; ++t) { Bn[2 * r][t] = split_acc<0>(acc[t]); Bn[2 * r + 1][t] = split_acc<1>(acc[t]); }
      }
#pragma unroll
      for (int kb = 0; kb < KBH; ++kb)
#pragma unroll
        for (int t = 0; t < NT; ++t) Bd[kb][t] = Bn[kb][t];
    }
    const uint32_t* sw0T = swhT + (NH - 1) * split_matrix_dwords(KBH, RT);
    f32x16 acc[NT];
    mma_split_row<KBH, NT>(sw0T, 0, 1, lane, Bd, acc, zero_block());
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) df[t][j] = acc[t][j] * ds[t];
  }
}

// The exact fp32 chains of the forward (v_mfma_f32_32x32x2_f32; the accumulators of a layer ARE the next layer's B operand:
// sdf_fused.hip's header), as sdf_fwd_kernel and atlas_sdf_kernel run them behind MISO_F_EXACT_F32.  w0p / whp / b0 / bh / wo:
// the fp32 pack's forward part in LDS.  Same outputs as decoder_fwd_split (p0 / p1: this lane's partial output sums).
template <int F, int H, int NH, int FN>
__device__ __forceinline__ void decoder_fwd_exact(const float* __restrict__ w0p, const float* __restrict__ whp,
                                                  const float* __restrict__ b0, const float* __restrict__ bh,
                                                  const float* __restrict__ wo, int lane, const float (&f)[FN],
                                                  uint32_t (&mw)[(NH + 1) * (H / 32)], float& p0, float& p1) {
  constexpr int RT = H / 32, KS0 = (F + 1) / 2, KS1 = H / 2;
  const int hi = lane >> 5;
  // ---- layer 0: buf[0][r][t] = b0 + W0 * feats -------------------------------
  // Two accumulator sets ping-pong between layers (ReLU is applied in place, so
  // no third copy of the 64 activation registers is ever live).
  f32x16 buf[2][RT][2];
  {
    // the bias enters as the C operand of the first MFMA of each chain (one register block per
    // row tile, shared by both point tiles): no accumulator initialisation moves
    f32x16 bias[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) bias[r][j] = b0[32 * r + row_of(j, hi)];
#pragma unroll
    for (int s = 0; s < KS0; ++s) {
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(f[2 * s]), __float_as_uint(f[2 * s + 1]),
                                                 false, false);
      float bt0 = __uint_as_float(sw[0]), bt1 = __uint_as_float(sw[1]);
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        float a = w0p[(s * 64 + lane) * RT + r];
        buf[0][r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt0, s == 0 ? bias[r] : buf[0][r][0], 0, 0, 0);
        buf[0][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bt1, s == 0 ? bias[r] : buf[0][r][1], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    uint32_t m = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        buf[0][r][t][j] = relu1(buf[0][r][t][j]);
        push_gt0(m, buf[0][r][t][j]);
      }
    mw[r] = m;
  }
  // ---- hidden HxH layers but the last ----------------------------------------------
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
          buf[ni][r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][1][j], ks == 0 ? bias[r] : buf[ni][r][1], 0, 0, 0);
        }
      }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      uint32_t m = 0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          buf[ni][r][t][j] = relu1(buf[ni][r][t][j]);
          push_gt0(m, buf[ni][r][t][j]);
        }
      mw[(h + 1) * RT + r] = m;
    }
  }
  // ---- last hidden layer + output layer (out_dim = 1), one 32-row tile at a time ----------
  // The last hidden activations feed only the output dot product, so each row tile is reduced
  // into (p0, p1) as soon as its MFMA chain ends: 32 accumulator registers live instead of 64.
  if (NH == 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float wv = wo[32 * r + row_of(j, hi)];
        p0 += wv * buf[0][r][0][j];
        p1 += wv * buf[0][r][1][j];
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
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, buf[ci][rp][1][j], ks == 0 ? bias : a1, 0, 0, 0);
        }
      uint32_t m = 0, m1 = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float y0 = relu1(a0[j]), y1 = relu1(a1[j]);
        push_gt0(m, y0);
        push_gt0(m1, y1);
        float wv = wo[32 * r + row_of(j, hi)];
        p0 += wv * y0;
        p1 += wv * y1;
      }
      mw[(h + 1) * RT + r] = (m << 16) | m1;
    }
  }
}

}  // namespace miso
