// Split-precision decoder products on the 16-bit matrix cores of gfx950 (round 6).
//
// The decoder of GridNet (grid_opt/models/modules.py:11-32: Linear / ReLU chains in fp32) is bound by the fp32 matrix rate
// when it runs on v_mfma_f32_32x32x2_f32 (64 clocks for 4 096 FLOP per SIMD).  v_mfma_f32_32x32x16_bf16 delivers 32 768 FLOP
// in 32 clocks.  Every fp32 operand is therefore written as the exact sum of three bf16 pieces,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (round to nearest even; the
//                                                                                       residuals are exact in fp32)
// which carries 3 x 8 = 24 significand bits -- all of an fp32 -- and a product a.b is evaluated as the six piece products of
// order <= 2 (a0 b2, a1 b1, a2 b0, a0 b1, a1 b0, a0 b0; the three dropped ones are <= 2^-24 |a b| together), each exact in the
// matrix core's fp32 accumulation, smallest first.  A 64 x 64 layer on 64 points is then 96 bf16 matrix instructions of 32
// clocks instead of 128 fp32 ones of 64: 2.67 x fewer matrix clocks, and -- unlike the fp32 form, which shares the vector
// datapath (tools/ubench/mfma_valu.hip) -- they run beside the vector instructions.  The weights are split once, in
// mlp_pack_kernel; activations are split in registers right where ReLU / the gate produces them (5.5 vector instructions
// per element).  Measured error against float64: tools/ubench/mlp_split.hip, tests/test_split_precision.py.
//
// Operand layouts (v_mfma_f32_32x32x16_bf16, D = A(32 x 16) B(16 x 32) + C):
//   A: lane l holds row (l & 31), contraction elements 8 (l >> 5) + i, i = 0..7 (four VGPRs, element i in dword i / 2,
//      half i & 1);  B: lane l holds column (l & 31), the same contraction elements;  C / D: as every 32 x 32 form --
//      register j of lane l is row row_of(j, l >> 5) = (j & 3) + 8 (j >> 2) + 4 (l >> 5), column l & 31.
//   The contraction index is only a label: A and B must agree on it, nothing else.  So, as in the fp32 chains of
//   sdf_fused.hip, a layer's accumulators ARE the next layer's B operand without any cross-lane movement: registers
//   8 h .. 8 h + 7 (h = 0, 1) of row tile rp form k-block kb = 2 rp + h, element i <-> neuron 32 rp + row_of(8 h + i, l >> 5);
//   the weights are packed to match (split_k_acc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace miso {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ __forceinline__ int row_of(int j, int hi) { return (j & 3) + 8 * (j >> 2) + 4 * hi; }

// contraction label -> input index, for B operands that are a previous layer's accumulators (k-block kb, lane half hi,
// element i) and for B operands built from lane-per-point feature rows
__host__ __device__ __forceinline__ int split_k_acc(int kb, int hi, int i) {
  return 32 * (kb >> 1) + row_of(8 * (kb & 1) + i, hi);
}
__host__ __device__ __forceinline__ int split_k_feat(int kb, int hi, int i) { return 16 * kb + 8 * hi + i; }

// bf16 (round to nearest even) of a float, as its 16 bits; and the three pieces of a weight (host or device, pack time)
__host__ __device__ __forceinline__ uint32_t bf16_rne_bits(float x) {
  union { float f; uint32_t u; } c; c.f = x;
  if ((c.u & 0x7fffffffu) > 0x7f800000u) return (c.u >> 16) | 0x40u;      // NaN stays NaN
  return (c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16;
}
__host__ __device__ __forceinline__ float bf16_bits_to_float(uint32_t b) {
  union { float f; uint32_t u; } c; c.u = b << 16; return c.f;
}
__host__ __device__ __forceinline__ void bf16_split3(float w, uint32_t out[3]) {
  out[0] = bf16_rne_bits(w);
  const float r1 = w - bf16_bits_to_float(out[0]);
  out[1] = bf16_rne_bits(r1);
  const float r2 = r1 - bf16_bits_to_float(out[1]);
  out[2] = bf16_rne_bits(r2);
}

// Dword offset (inside one matrix's split block) of the A operand for (k-block kb, row tile r, piece q), lane l: 16 B per
// lane, consecutive lanes consecutive -- one conflict-free ds_read_b128 per wavefront.
__host__ __device__ __forceinline__ int split_a_dword(int kb, int r, int q, int lane, int RT) {
  return (((kb * RT + r) * 3 + q) * 64 + lane) * 4;
}
__host__ __device__ __forceinline__ int split_matrix_dwords(int KB, int RT) { return KB * RT * 3 * 64 * 4; }

#if defined(__HIPCC__)
// ---------------------------------------------------------------------------------------------------------------------
// three bf16x8 pieces of eight fp32 values (one k-block half of a B operand)
struct Split3 { u32x4 q[3]; };

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {      // v_cvt_pk_bf16_f32: low half = bf16(a)
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void split_pair(float a, float b, uint32_t& h, uint32_t& m, uint32_t& l) {
#pragma clang fp contract(off)
#ifdef MISO_ABL_NO_SPLIT      // dev ablation (wrong results): one instruction instead of nine
  h = cvt_pk_bf16(a, b); m = h; l = h;
  return;
#endif
  h = cvt_pk_bf16(a, b);
  const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
  m = cvt_pk_bf16(ra, rb);
  const float ra2 = ra - __uint_as_float(m << 16), rb2 = rb - __uint_as_float(m & 0xffff0000u);
  l = cvt_pk_bf16(ra2, rb2);
}

__device__ __forceinline__ Split3 split8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  uint32_t h[4], m[4], l[4];
  split_pair(v0, v1, h[0], m[0], l[0]);
  split_pair(v2, v3, h[1], m[1], l[1]);
  split_pair(v4, v5, h[2], m[2], l[2]);
  split_pair(v6, v7, h[3], m[3], l[3]);
  Split3 s;
  s.q[0] = u32x4{h[0], h[1], h[2], h[3]};
  s.q[1] = u32x4{m[0], m[1], m[2], m[3]};
  s.q[2] = u32x4{l[0], l[1], l[2], l[3]};
  return s;
}
// registers 8 h .. 8 h + 7 of an accumulator block
template <int h>
__device__ __forceinline__ Split3 split_acc(const f32x16& v) {
  return split8(v[8 * h + 0], v[8 * h + 1], v[8 * h + 2], v[8 * h + 3], v[8 * h + 4], v[8 * h + 5], v[8 * h + 6], v[8 * h + 7]);
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
#ifdef MISO_ABL_NO_MFMA       // dev ablation (wrong results): no matrix instructions (their operands' producers go with them)
  return c;
#endif
#ifdef MISO_ABL_FAKE_MFMA     // dev ablation (wrong results): operands still produced and read, one vector instruction instead
  f32x16 r = c;
  r[0] += __uint_as_float((a[0] ^ b[0] ^ a[1] ^ b[1] ^ a[2] ^ b[2] ^ a[3] ^ b[3]) & 0x007fffffu);
  return r;
#endif
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// (the all-row-tiles form; the kernels use decoder.hpp's mma_split_row, one row tile at a time -- this one serves
// tools/ubench/mlp_split.hip)
// out[r][t] (+)= A B for KB k-blocks, RT output row tiles, NT point tiles; the six piece products per k-block, smallest
// first.  FIRST: the very first product of every accumulator takes `init[r]` as its C operand (a bias block shared by the
// point tiles, or zeros) -- no accumulator initialisation moves.  A: the matrix's split block in LDS (split_a_dword).
template <int KB, int RT, int NT>
__device__ __forceinline__ void mma_split(const uint32_t* __restrict__ A, int lane, const Split3 (&B)[KB][NT],
                                          f32x16 (&out)[RT][NT], const f32x16 (&init)[RT]) {
  // (piece of A, piece of B) in the order of accumulation
  constexpr int QA[6] = {2, 1, 0, 1, 0, 0};
  constexpr int QB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    u32x4 a[RT][3];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[r][q] = *reinterpret_cast<const u32x4*>(A + split_a_dword(kb, r, q, lane, RT));
#pragma unroll
    for (int c = 0; c < 6; ++c)
#pragma unroll
      for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          out[r][t] = mfma_bf16(a[r][QA[c]], B[kb][t].q[QB[c]], (kb == 0 && c == 0) ? init[r] : out[r][t]);
  }
}
#endif  // __HIPCC__

}  // namespace miso
