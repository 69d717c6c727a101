// Owner-computes grid gradient on the fp32 matrix cores ("matrix-core pull"): the fast path of miso_grad_pull /
// the binned training step for first-order gradients.  Semantics: the grid half of grid_sampler_3d_backward
// (grad_input[c, corner] += w_corner * gOut[c]; third_party/cuda_gridsample_grad2/gridsample_cuda.cu:462-481 for the
// second-order sibling, ATen's kernel for the first order), formed without atomics and without a zero-fill.
//
// grad_pull_block_kernel (grad_pull.hip) gives a wavefront a tile and lets lane = vertex walk the records of its eight
// adjacent cells: ~2.5 K vector instructions per tile for 160 useful multiply-adds at full lanes (the finest level of
// a uniform batch has ~1 sample per cell), a ~50 us serial chain per workgroup of counting sorts, scans and loops at
// ~10 % lane utilisation.  Here the sum is a small dense product instead.  A level's vertices are cut into SUB-BRICKS
// of 4 x 4 x ZS vertices (ZS = 16 / C: 2 for C = 8, 4 for C = 4).  Every sample that touches a sub-brick contributes
//     G[(vz,c)][(vx,vy)] += (wz[vz] d[c]) * (wx[vx] wy[vy])
// with its per-axis trilinear weights relative to the sub-brick (two non-zeros among four per axis, zero where its
// corner lies outside) -- a 16 x 16 outer product, four samples per v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains,
// 32 cycles per SIMD).  One eighth of the products are non-zero, which is 4 x the cost of perfectly packed vector FMAs
// and a fraction of what finding and ordering the non-zeros costs the vector formulation.  The accumulator tile comes
// out as one float4 (four channels of one vertex) per lane: 128-byte runs along x, whole rows stored.
//
// One workgroup (8 wavefronts) per 2 x 2 x 2 block of sort tiles, as in grad_pull_block_kernel:
//   (1) sweep the tiles around the block once, cooperatively; a sample inside the block's widest catchment box goes
//       into an LDS table {xn, sorted index} (resumable: a crowded block is worked off in epochs, the first one stores,
//       the others add -- no slice queue, no second launch);
//   (2) route: per table entry and level the sub-bricks it touches (1..8), counted per sub-brick, prefix-summed with
//       every list padded to a multiple of four, then filled: ONE pool of (sample, sub-brick) pairs ordered by
//       sub-brick;
//   (3) multiply: the pool is dealt to the wavefronts in equal contiguous shares -- load balance by pair count whatever
//       the distribution; a lane stages one pair (three cell_of, 12 weights, the d-feat row: 20 floats in LDS), four
//       pairs feed one MFMA; a sub-brick that ends inside a share is stored straight from the accumulators, one that
//       straddles a boundary leaves partial tiles in LDS that the wavefront where it starts sums and stores.
// Used when no pulled level has fewer than 2/3 as many vertices as tiles on an axis (the sweep then stays within 7 tile
// rows per axis); everything else -- second-order weights (MODE 1), grids coarser than the binning -- keeps grad_pull.hip.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "grad_pull.hpp"

namespace miso {

constexpr int MC_WAVES = 8;
constexpr int MC_CAND = 1152;          // table entries (a uniform cfg-2 batch: 1000 +- 32 per block: at 1024 every fifth block ran a
                                       // second epoch)
constexpr int MC_POOL = 4608;          // (sample, sub-brick) pairs incl. padding (uniform cfg-2: 3830 +- 130, max 4200; more: halved ranges)
constexpr int MC_ITEMS = 512;          // sub-brick code space: level (2 bits) | sz (3) | sy (2) | sx (2)
constexpr int MC_UN = 8;               // 64-sample steps in flight per wavefront in the sweep
constexpr int MC_NULL = 0xFFFF;        // pool padding
constexpr int MC_LVL = 20;             // words per level record in LDS
constexpr int MC_MAXL = 4;

// one pulled level, as the kernel reads it (everything by value: no dependent kernel-argument loads)
struct McLv {
  float* grad;
  unsigned char* touched;    // miso_level_t.grad_touched or nullptr
  int X, Y, Z, foff;
  int sX, sY, sZ;            // element strides of grad (channels are contiguous)
  int live;                  // 0: an ignored level -- zero-filled, nothing routed
  int bdiv[3];               // size / T where T divides the size, else 0
  float inv_size[3];
};

struct McK {
  int T[3];                  // sort tiles per axis
  int nb[3];                 // blocks per axis: ceil(T / 2)
  const int* tile_off;
  const float4* xn;          // tile-sorted normalised coordinates
  const float* dfeat;        // d-feat rows: row p (tile-sorted order) or, with perm, row perm[p]
  const int* perm;
  int64_t ld;
  int nl;                    // pulled levels
  int overwrite;             // 1: grad = sum, 0: grad += sum
  unsigned int drow_bytes;   // bytes of the d-feat rows: n * ld * 4 (< 2^31: row offsets are 32-bit buffer offsets)
  int debug;                 // dev ablation (MISO_DEBUG_PULL): 8 no multiply phase, 16 nothing after the sweep, 64 / 128 half the
                             // workgroups.  (Bits 1 2 4 -- no MFMAs / staging loads / stores -- sat inside the multiply loop
                             // and are gone: the branch around the loads alone made the compiler wait for them at every
                             // later join)
  int cand_cap, pool_cap;    // table entries / pool pairs in use: MC_CAND / MC_POOL (tests: MISO_MC_SMALL shrinks them so that the
                             // epoch and pool-overflow paths run on ordinary batches)
  int prof_wave;             // dev: the wavefront that stamps (MISO_MC_PROF=1+wave)
  unsigned long long* prof;  // dev (MISO_MC_PROF): clocks per phase, or nullptr
  McLv lv[MC_MAXL];
};

// LDS layout (32-bit words)
constexpr int MC_CS = 68;              // words per staged component: 64 pairs, padded so that 16-byte reads of different
                                       // components fall into different banks
template <int C, int NLV> struct McLds {
  static constexpr int ZS = 16 / C;
  static constexpr int O_CAND = 0;                                   // float4[MC_CAND + 1]: the last is the null entry
  static constexpr int O_POOL = O_CAND + (MC_CAND + 1) * 4;          // uint32[MC_POOL]: table slot | sub-brick code << 16
  static constexpr int O_CNT = O_POOL + MC_POOL;                     // int[MC_ITEMS]: pairs per sub-brick
  static constexpr int O_CUR = O_CNT + MC_ITEMS;                     // int[MC_ITEMS]: fill cursors (absolute)
  static constexpr int O_MISC = O_CUR + MC_ITEMS;                    // 96 ints
  static constexpr int O_LVL = O_MISC + 96;                          // MC_MAXL block-geometry records of MC_LVL words
  static constexpr int O_LVG = O_LVL + MC_MAXL * MC_LVL;             // MC_MAXL records of 8 words: grad, strides, touched
  static constexpr int O_GEO = O_LVG + MC_MAXL * 8;                  // the block's catchment box and tile-row bounds (G_*)
  static constexpr int O_STAGE = O_GEO + 144;                        // per wavefront: (8 + ZS weights + C d-feats) x MC_CS
  static constexpr int NCOMP = 8 + ZS + C;
  static constexpr int STAGE = NCOMP * MC_CS;
  static constexpr int WORDS = O_STAGE + MC_WAVES * STAGE;
  // while routing runs the staging area is idle: it holds the routing codes (uint16 per table entry and level) and,
  // behind them, the sub-bricks' cost prefix
  static constexpr int O_COST = O_STAGE + MC_MAXL * MC_CAND / 2;
  static_assert(O_COST + MC_ITEMS <= WORDS, "codes + cost prefix must fit the staging area");
  static_assert(WORDS * 4 <= 80 * 1024, "two workgroups per CU");
};
// O_MISC slots
constexpr int M_NSURV = 0, M_FULL = 1, M_MORE = 2, M_PTOT = 3, M_A = 8, M_B = 16, M_CONT = 24, M_EMPTY = 32, M_BOUND = 40,
              M_PROF = 64;
// a share's cost: 2 per group of four pairs + MC_FLUSH per sub-brick that ends in it (storing a tile costs ~2.5 groups)
// O_GEO slots
constexpr int G_ULO = 0, G_UHI = 4, G_NROWS = 8, G_RS = 16, G_RE = 80;
constexpr int MC_FLUSH = 5;

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Geometry of level d for the block, in its LDS record (wave-uniform address: broadcast reads):
//   [0..2] size X Y Z   [3] foff   [4..6] v0   [7] live   [8..10] E   [12..14] size / 2   [16..18] (size - 1) / 2 - v0  (floats)
struct McLevel { int size[3], v0[3], E[3], foff; };
__device__ __forceinline__ McLevel mc_level(const int* lvl, int d) {
  const int4 r0 = *reinterpret_cast<const int4*>(lvl + d * MC_LVL);
  const int4 r1 = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 4);
  const int4 r2 = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 8);
  McLevel m;
  m.size[0] = r0.x; m.size[1] = r0.y; m.size[2] = r0.z; m.foff = r0.w;
  m.v0[0] = r1.x; m.v0[1] = r1.y; m.v0[2] = r1.z;
  m.E[0] = r2.x; m.E[1] = r2.y; m.E[2] = r2.z;
  return m;
}

// dev: wavefront 0 adds the clocks since the previous stamp to prof[phase] (MISO_MC_PROF; see the launcher)
#define MC_STAMP(PH)                                                                              \
  if (pk.prof && wave == pk.prof_wave) {                                                          \
    const unsigned long long now_ = __builtin_readcyclecounter();                                 \
    if (lane == 0) misc[M_PROF + (PH)] += (int)(now_ - t_prev);                                       \
    t_prev = now_;                                                                                \
  }

template <int C, int NLV>
__global__ __launch_bounds__(64 * MC_WAVES, 4) void grad_pull_mc_kernel(McK pk) {
  using L = McLds<C, NLV>;
  constexpr int ZS = 16 / C;                    // sub-brick vertices along z
  constexpr int ZSH = (C == 8) ? 1 : 2;         // log2(ZS)
  constexpr int NT = 64 * MC_WAVES;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* ismem = reinterpret_cast<int*>(smem);
  float4* cand = reinterpret_cast<float4*>(smem + L::O_CAND);
  unsigned* pool = reinterpret_cast<unsigned*>(ismem + L::O_POOL);
  int* cnt = ismem + L::O_CNT;
  int* cur = ismem + L::O_CUR;
  int* misc = ismem + L::O_MISC;
  int* coff = ismem + L::O_COST;                 // (in the staging area: idle while routing runs)
  int* lvl = ismem + L::O_LVL;
  int* lvg = ismem + L::O_LVG;
  int* geo = ismem + L::O_GEO;
  uint16_t* codes = reinterpret_cast<uint16_t*>(ismem + L::O_STAGE);      // [NLV][MC_CAND]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* stg = smem + L::O_STAGE + wave * L::STAGE;       // [12 + C][MC_CS]: wx[4], wy[4], wz[4], d[C]; pair (g, k) at k 16 + g
  // set bits of a ballot below this lane (v_mbcnt: no lane mask to keep in registers)
  auto below = [](unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  };
  // accumulator-tile coordinates of this lane: column (vx, vy) = lane & 15, rows 4 (lane >> 4) .. + 3 = (vz, c)
  const int a_vx = lane & 3, a_vy = (lane >> 2) & 3;
  const int a_vz = (C == 8) ? (lane >> 5) : (lane >> 4), a_c0 = (C == 8) ? ((lane >> 4) & 1) * 4 : 0;
  // operand coordinates: A[i][k], B[k][j] with i = j = lane & 15, k = lane >> 4; row i = (vz, c)
  const int o_k = lane >> 4, o_i = lane & 15;
  const int o_vz = (C == 8) ? (o_i >> 3) : (o_i >> 2), o_c = (C == 8) ? (o_i & 7) : (o_i & 3);
  // this lane's four operand components in the staging area (word offsets; + 4 Q for quad Q of the chunk)
  const int rd_x = (o_i & 3) * MC_CS + o_k * 16, rd_y = (4 + (o_i >> 2)) * MC_CS + o_k * 16;
  const int rd_z = (8 + o_vz) * MC_CS + o_k * 16, rd_d = (8 + ZS + o_c) * MC_CS + o_k * 16;
  // and where it writes the pair it stages: pair = lane, group g = lane >> 2, k = lane & 3
  const int wr_p = (lane & 3) * 16 + (lane >> 2);

  if (threadIdx.x == 0) {
#pragma unroll
    for (int d = 0; d < NLV; ++d) {
      if (d >= pk.nl) continue;
      const McLv& lv = pk.lv[d];
      *reinterpret_cast<float**>(lvg + d * 8) = lv.grad;
      lvg[d * 8 + 2] = lv.sX; lvg[d * 8 + 3] = lv.sY; lvg[d * 8 + 4] = lv.sZ;
      *reinterpret_cast<unsigned char**>(lvg + d * 8 + 6) = lv.touched;
    }
  }
  {
    const int lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);      // (workgroup i runs on XCD i % 8)
    if ((pk.debug & 64) && ((lin >> 3) & 1)) return;       // dev: half the workgroups of every XCD, alternating
    if ((pk.debug & 128) && lin >= (int)(gridDim.x * gridDim.y * gridDim.z) / 2) return;      // dev: the first half only
  }      // dev: half the workgroups (one per CU): latency- or throughput-bound?
  if (pk.prof && threadIdx.x < 16) misc[M_PROF + threadIdx.x] = 0;
  unsigned long long t_prev = pk.prof ? __builtin_readcyclecounter() : 0ull;
  {
    // ---- per-block geometry -> LDS records (computed by the first threads, read back where needed) ----------------
    // level d, axis a: the block owns vertices [v0, v0 + E) = the bricks of its two tiles; points that touch them lie in
    // tiles [tlo, thi] (widened by one numerator unit against the rounding of the sort's tile_of)
    // Wavefront 0 forms it lane-parallel, lane = (level, axis), and leaves the block's catchment and the bounds of its tile
    // rows in `geo`; the others read `geo` behind the epoch's first barrier.  (As ~600 scalar instructions in each of
    // the eight wavefronts it took the same time -- the set-up waits for the kernel arguments and the tile offsets --
    // but cost seven more spilled registers.)
    if (wave == 0) {
      int sz = 1, B = 0, foff = 0, live_i = 0, Ta = 1, ba = 0;
      float inv = 1.0f;
#pragma unroll
      for (int d = 0; d < NLV; ++d) {
        const McLv& lv = pk.lv[d];
#pragma unroll
        for (int a = 0; a < 3; ++a)
          if (lane == 3 * d + a) {
            sz = (a == 0) ? lv.X : (a == 1) ? lv.Y : lv.Z;
            B = lv.bdiv[a]; inv = lv.inv_size[a]; foff = lv.foff; live_i = lv.live;
          }
      }
      const int d_ = lane / 3, a_ = lane - 3 * d_;
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (a_ == a) { Ta = pk.T[a]; ba = (a == 0) ? (int)blockIdx.x : (a == 1) ? (int)blockIdx.y : (int)blockIdx.z; }
      const bool in = lane < 3 * pk.nl;
      const int t0 = 2 * ba, t1 = min(2 * ba + 2, Ta);
      int lo, hi, v0, E;
      if (B) {
        v0 = t0 * B; E = (t1 - t0) * B;
        lo = t0 - 1; hi = t1;                      // exact for size = B T
      } else {
        v0 = t0 * sz / Ta;
        E = t1 * sz / Ta - v0;
        lo = floor_div((2 * v0 - 1) * Ta - 1, 2 * sz);
        hi = floor_div((2 * (v0 + E) + 1) * Ta + 1, 2 * sz);
      }
      if (in) {
        lvl[d_ * MC_LVL + a_] = sz; lvl[d_ * MC_LVL + 4 + a_] = v0; lvl[d_ * MC_LVL + 8 + a_] = E;
        if (a_ == 0) { lvl[d_ * MC_LVL + 3] = foff; lvl[d_ * MC_LVL + 7] = live_i; }
        // the routing's position relative to the block, one fma: xn size/2 + ((size - 1)/2 - v0)
        reinterpret_cast<float*>(lvl)[d_ * MC_LVL + 12 + a_] = 0.5f * (float)sz;
        reinterpret_cast<float*>(lvl)[d_ * MC_LVL + 16 + a_] = 0.5f * (float)(sz - 1) - (float)v0;
      }
      // (an ignored level is still zero-filled, its points are just not routed)
      const bool use = in && live_i != 0;
      int c_lo = use ? max(lo, 0) : (1 << 20), c_hi = use ? min(hi, Ta - 1) : -1;
      float c_ul = use ? (2.0f * v0 - 1.0f) * inv - 1.0f - 8e-6f : 3e30f;
      float c_uh = use ? (2.0f * (v0 + E) + 1.0f) * inv - 1.0f + 8e-6f : -3e30f;
      {     // lane a (< 3) gathers its axis over the levels
        const int l0 = c_lo, h0 = c_hi;
        const float ul0 = c_ul, uh0 = c_uh;
#pragma unroll
        for (int k = 1; k < NLV; ++k) {
          c_lo = min(c_lo, __shfl_down(l0, 3 * k)); c_hi = max(c_hi, __shfl_down(h0, 3 * k));
          c_ul = fminf(c_ul, __shfl_down(ul0, 3 * k)); c_uh = fmaxf(c_uh, __shfl_down(uh0, 3 * k));
        }
      }
      if (lane < 3) {
        geo[G_ULO + lane] = __float_as_int(c_ul); geo[G_UHI + lane] = __float_as_int(c_uh);
      }
      int tlo[3], thi[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) { tlo[a] = __builtin_amdgcn_readlane(c_lo, a); thi[a] = __builtin_amdgcn_readlane(c_hi, a); }
      const bool any_rows = thi[0] >= tlo[0] && thi[1] >= tlo[1] && thi[2] >= tlo[2];
      const int ny = any_rows ? thi[1] - tlo[1] + 1 : 0;
      const int nrows_ = any_rows ? ny * (thi[2] - tlo[2] + 1) : 0;      // <= 49 (launcher: 3 size >= 2 T on every axis)
      int rs = 0, re = 0;
      if (lane < nrows_) {
        const int ry = tlo[1] + lane % ny, rz = tlo[2] + lane / ny;
        rs = pk.tile_off[(rz * pk.T[1] + ry) * pk.T[0] + tlo[0]];
        re = pk.tile_off[(rz * pk.T[1] + ry) * pk.T[0] + thi[0] + 1];
      }
      geo[G_RS + lane] = rs; geo[G_RE + lane] = re;
      if (lane == 0) geo[G_NROWS] = nrows_;
    }
    float ulo[3] = {0.f, 0.f, 0.f}, uhi[3] = {0.f, 0.f, 0.f};
    int nrows = 0, rs_v = 0, re_v = 0;
    int r_cur = wave;                                            // this wavefront's rows: wave, wave + 8, ...
    int p_cur = -1;                                              // (read from rs_v after the barrier below)
    bool first = true;                                           // nothing of this block has been stored yet

    for (;;) {   // ---- epochs: fill the table, route and multiply it; once for a block the table can hold ------------
      if (threadIdx.x == 0) {
        misc[M_NSURV] = 0; misc[M_FULL] = 0; misc[M_MORE] = 0;
        cand[MC_CAND] = make_float4(2e30f, 2e30f, 2e30f, __int_as_float(-1));      // what pool padding points at
      }
      for (int i = threadIdx.x; i < MC_ITEMS; i += NT) cnt[i] = 0;
      for (int i = threadIdx.x; i < MC_POOL; i += NT) pool[i] = 0xFFFFFFFFu;
      __syncthreads();
      if (p_cur < 0) {      // the first epoch: the block's catchment and row bounds, as wavefront 0 left them
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          ulo[a] = __int_as_float(__builtin_amdgcn_readfirstlane(geo[G_ULO + a]));
          uhi[a] = __int_as_float(__builtin_amdgcn_readfirstlane(geo[G_UHI + a]));
        }
        nrows = __builtin_amdgcn_readfirstlane(geo[G_NROWS]);
        rs_v = geo[G_RS + lane]; re_v = geo[G_RE + lane];
        p_cur = (r_cur < nrows) ? __builtin_amdgcn_readlane(rs_v, min(r_cur, 63)) : 0;
      }
      MC_STAMP(0)
      // ---- (1) sweep ------------------------------------------------------------------------------------------------
      while (r_cur < nrows) {
        const int p_end = __builtin_amdgcn_readlane(re_v, r_cur);
        if (p_cur >= p_end) {
          r_cur += MC_WAVES;
          if (r_cur < nrows) p_cur = __builtin_amdgcn_readlane(rs_v, r_cur);
          continue;
        }
        if (__builtin_amdgcn_readfirstlane(
                __hip_atomic_load(&misc[M_FULL], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))
          break;
        const int nstep = __builtin_amdgcn_readfirstlane(min(MC_UN, (p_end - p_cur + 63) / 64));
        float4 c4[MC_UN];
#pragma unroll
        for (int u = 0; u < MC_UN; ++u) {
          const int p = p_cur + u * 64 + lane;
          c4[u] = make_float4(2e30f, 2e30f, 2e30f, 0.f);
          if (u < nstep && p < p_end) c4[u] = pk.xn[p];
        }
        // one reservation for the whole trip: the table slots of step u start at base + (survivors of the steps before)
        unsigned long long m[MC_UN];
        int tot = 0;
#pragma unroll
        for (int u = 0; u < MC_UN; ++u) {
          const bool in = c4[u].x >= ulo[0] && c4[u].x < uhi[0] && c4[u].y >= ulo[1] && c4[u].y < uhi[1] &&
                          c4[u].z >= ulo[2] && c4[u].z < uhi[2];
          m[u] = __ballot(in);
          tot += (int)__popcll(m[u]);
        }
        int done = nstep;
        if (tot) {
          int base = 0;
          if (lane == 0) base = atomicAdd(&misc[M_NSURV], tot);
          base = __builtin_amdgcn_readfirstlane(base);
          int run = base;
#pragma unroll
          for (int u = 0; u < MC_UN; ++u) {
            if (u >= done) continue;
            const int nin = (int)__popcll(m[u]);
            if (run + nin > pk.cand_cap) { done = u; continue; }   // table full: steps u.. are redone next epoch
            if ((m[u] >> lane) & 1ull)
              cand[run + below(m[u])] =
                  make_float4(c4[u].x, c4[u].y, c4[u].z, __int_as_float(p_cur + u * 64 + lane));
            run += nin;
          }
          if (done < nstep) {
            // what was reserved beyond the steps that fitted becomes null entries (outside every box)
            for (int i = run + lane; i < min(base + tot, pk.cand_cap); i += 64) cand[i] = make_float4(2e30f, 2e30f, 2e30f, 0.f);
            if (lane == 0) misc[M_FULL] = 1;
          }
        }
        p_cur += 64 * done;
        if (done < nstep) break;
      }
      MC_STAMP(1)
      if (r_cur < nrows && lane == 0) misc[M_MORE] = 1;
      __syncthreads();
      MC_STAMP(2)
      const int ncand = (pk.debug & 16) ? 0 : min(__builtin_amdgcn_readfirstlane(misc[M_NSURV]), pk.cand_cap);
      const bool more = __builtin_amdgcn_readfirstlane(misc[M_MORE]) != 0;

      // ---- (2) + (3): route and multiply the table, in one range unless the pool overflows --------------------------
      int lo = 0, hi = ncand;
      bool counted = false;
      while (lo < ncand || (first && !counted)) {
        // (2a) count the pairs of every sub-brick; remember every entry's code.  A level with at most four sub-bricks
        // in the block (the coarsest of a pyramid) would put half a wavefront on one counter -- same-address LDS
        // atomics serialise, ~10 clocks per lane -- so its counts are formed with ballots, one atomic per sub-brick.
        for (int t0 = lo; t0 < hi; t0 += NT) {
          const int t = t0 + (int)threadIdx.x;
          const bool act = t < hi;
          const float4 c4 = act ? cand[t] : make_float4(2e30f, 2e30f, 2e30f, 0.f);
          const float xs[3] = {c4.x, c4.y, c4.z};
#pragma unroll
          for (int d = 0; d < NLV; ++d) {
            if (d >= pk.nl || !lvl[d * MC_LVL + 7]) continue;
            // (cell relative to the block by ONE fma: a sample within an ulp of a cell face may be routed by the neighbouring
            // cell -- the sub-brick it then misses would have got a weight of that ulp)
            const int4 re = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 8);
            const float4 ra = *reinterpret_cast<const float4*>(lvl + d * MC_LVL + 12);
            const float4 rb = *reinterpret_cast<const float4*>(lvl + d * MC_LVL + 16);
            const int E3[3] = {re.x, re.y, re.z};
            const float fa[3] = {ra.x, ra.y, ra.z}, fb[3] = {rb.x, rb.y, rb.z};
            int s0[3], two[3];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
              const float pr = fmaf(xs[a], fa[a], fb[a]);
              const int rel = (int)fminf(fmaxf(floorf(pr), -2.0f), 64.0f);
              const int vl = max(rel, 0), vh = min(rel + 1, E3[a] - 1);
              ok = ok && vl <= vh;
              const int sh = (a == 2) ? ZSH : 2;
              s0[a] = vl >> sh;
              two[a] = ((vh >> sh) != s0[a]) ? 1 : 0;
            }
            const int code = ok ? (0x8000 | s0[0] | (s0[1] << 2) | (s0[2] << 4) | (two[0] << 7) | (two[1] << 8) |
                                   (two[2] << 9)) : 0;
            if (act) codes[d * MC_CAND + t] = (uint16_t)code;
            const int nsx = __builtin_amdgcn_readfirstlane((E3[0] + 3) >> 2),
                      nsy = __builtin_amdgcn_readfirstlane((E3[1] + 3) >> 2),
                      nsz = __builtin_amdgcn_readfirstlane((E3[2] + ZS - 1) >> ZSH);
            if (nsx * nsy * nsz <= 4) {
              int mycnt = 0, mycode = 0, j = 0;
              for (int jz = 0; jz < nsz; ++jz)
                for (int jy = 0; jy < nsy; ++jy)
                  for (int jx = 0; jx < nsx; ++jx, ++j) {
                    const bool member = ok && (unsigned)(jx - s0[0]) <= (unsigned)two[0] &&
                                        (unsigned)(jy - s0[1]) <= (unsigned)two[1] && (unsigned)(jz - s0[2]) <= (unsigned)two[2];
                    const unsigned long long mm = __ballot(member);
                    if (lane == j) { mycnt = (int)__popcll(mm); mycode = (d << 7) | (jz << 4) | (jy << 2) | jx; }
                  }
              if (mycnt) atomicAdd(&cnt[mycode], mycnt);
            } else if (ok) {
              const int item0 = (d << 7) | (s0[2] << 4) | (s0[1] << 2) | s0[0];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                if (dx <= two[0] && dy <= two[1] && dz <= two[2]) atomicAdd(&cnt[item0 + (dz << 4) + (dy << 2) + dx], 1);
              }
            }
          }
        }
        MC_STAMP(3)
        __syncthreads();
        MC_STAMP(4)
        // (2b) offsets: every sub-brick's list padded to a multiple of four pairs
        if (wave == 0) {
          int c8[8], tot = 0, nz = 0;
#pragma unroll
          for (int k = 0; k < 8; ++k) { c8[k] = (cnt[lane * 8 + k] + 3) & ~3; tot += c8[k]; nz += c8[k] ? 1 : 0; }
          int inc = tot | (nz << 16);                              // pairs (< 2^16) and non-empty sub-bricks in one scan
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
          const int all = __builtin_amdgcn_readlane(inc, 63);
          int run = (inc & 0xffff) - tot, nzr = (inc >> 16) - nz;
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            cur[lane * 8 + k] = run;                                // cursors are absolute pool positions (until the fill: the
                                                                    // sub-brick's first pair)
            coff[lane * 8 + k] = (run >> 1) + MC_FLUSH * nzr;
            run += c8[k]; nzr += c8[k] ? 1 : 0;
          }
          const int ptot_ = all & 0xffff, ctot = (ptot_ >> 1) + MC_FLUSH * (all >> 16);
          wave_sync_lds();
          // share bounds: wavefront b starts at the group where the cost reaches b / 8 of the total (lane b searches)
          if (lane <= MC_WAVES) {
            int bound = (lane == MC_WAVES) ? ptot_ >> 2 : 0;
            if (lane > 0 && lane < MC_WAVES) {
              const int target = (int)(((int64_t)lane * ctot) / MC_WAVES);
              int i = 0;                                           // the last sub-brick whose cost prefix is <= target
#pragma unroll
              for (int st = MC_ITEMS / 2; st > 0; st >>= 1)
                if (coff[i + st] <= target) i += st;
              const int groups = (cnt[i] + 3) >> 2;
              bound = (cur[i] >> 2) + min((target - coff[i]) >> 1, groups);
            }
            misc[M_BOUND + lane] = bound;
          }
          if (lane == 63) misc[M_PTOT] = ptot_;
        }
        __syncthreads();
        MC_STAMP(5)
        const int ptot = __builtin_amdgcn_readfirstlane(misc[M_PTOT]);
        if (ptot > pk.pool_cap) {      // a crowd on few sub-bricks: half the range at a time
          hi = lo + max(1, (hi - lo) >> 1);
          for (int i = threadIdx.x; i < MC_ITEMS; i += NT) cnt[i] = 0;
          __syncthreads();
          continue;
        }
        counted = true;
        // (2c) fill: all of an entry's returning atomics are issued before the first result is used
        for (int t0 = lo; t0 < hi; t0 += NT) {
          const int t = t0 + (int)threadIdx.x;
          const bool act = t < hi;
#pragma unroll
          for (int d = 0; d < NLV; ++d) {
            if (d >= pk.nl || !lvl[d * MC_LVL + 7]) continue;
            const int code = act ? (int)codes[d * MC_CAND + t] : 0;
            const bool ok = (code & 0x8000) != 0;
            const int s0x = code & 3, s0y = (code >> 2) & 3, s0z = (code >> 4) & 7;
            const int twx = (code >> 7) & 1, twy = (code >> 8) & 1, twz = (code >> 9) & 1;
            const int e0 = lvl[d * MC_LVL + 8], e1 = lvl[d * MC_LVL + 9], e2 = lvl[d * MC_LVL + 10];
            const int nsx = __builtin_amdgcn_readfirstlane((e0 + 3) >> 2), nsy = __builtin_amdgcn_readfirstlane((e1 + 3) >> 2),
                      nsz = __builtin_amdgcn_readfirstlane((e2 + ZS - 1) >> ZSH);
            const int n = nsx * nsy * nsz;
            if (n <= 4) {
              unsigned long long mm[4] = {0ull, 0ull, 0ull, 0ull};
              int mycnt = 0, mycode = 0;
              int jx = 0, jy = 0, jz = 0;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (j < n) {
                  const bool member = ok && (unsigned)(jx - s0x) <= (unsigned)twx && (unsigned)(jy - s0y) <= (unsigned)twy &&
                                      (unsigned)(jz - s0z) <= (unsigned)twz;
                  mm[j] = __ballot(member);
                  if (lane == j) { mycnt = (int)__popcll(mm[j]); mycode = (d << 7) | (jz << 4) | (jy << 2) | jx; }
                  if (++jx == nsx) { jx = 0; if (++jy == nsy) { jy = 0; ++jz; } }
                }
              }
              int base = 0;
              if (mycnt) base = atomicAdd(&cur[mycode], mycnt);
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                if (j < n && mm[j]) {
                  const int b = __builtin_amdgcn_readlane(base, j), cj = __builtin_amdgcn_readlane(mycode, j);
                  if ((mm[j] >> lane) & 1ull) pool[b + below(mm[j])] = (unsigned)t | ((unsigned)cj << 16);
                }
              }
            } else {
              const int item0 = (d << 7) | (code & 0x7f);
              int pos[8];
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                pos[k] = -1;
                if (ok && dx <= twx && dy <= twy && dz <= twz) {
                  const int item = item0 + (dz << 4) + (dy << 2) + dx;
                  pos[k] = atomicAdd(&cur[item], 1);
                }
              }
#pragma unroll
              for (int k = 0; k < 8; ++k) {
                const int dx = k & 1, dy = (k >> 1) & 1, dz = k >> 2;
                if (pos[k] >= 0) pool[pos[k]] = (unsigned)t | ((unsigned)(item0 + (dz << 4) + (dy << 2) + dx) << 16);
              }
            }
          }
        }
        MC_STAMP(6)
        // the first range also zero-fills the sub-bricks nothing touches (the gradient buffer is not cleared): wavefront w
        // looks at codes 16 w .. 16 w + 15 of every level
        if (first && pk.overwrite) {
          // (all levels' reads first, one round trip: the usual answer is "nothing to fill")
          const int j = 16 * wave + (lane & 15);
          const int sx = j & 3, sy = (j >> 2) & 3, sz = j >> 4;
          unsigned long long em[NLV];
#pragma unroll
          for (int d = 0; d < NLV; ++d) {
            em[d] = 0ull;
            if (d >= pk.nl) continue;
            const int4 v0l = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 4);      // v0, live
            const int4 e = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 8);
            const int c = cnt[(d << 7) | j];
            const bool valid = lane < 16 && 4 * sx < e.x && 4 * sy < e.y && ZS * sz < e.z;
            em[d] = __ballot(valid && !(v0l.w != 0 && c != 0));
          }
#pragma unroll
          for (int d = 0; d < NLV; ++d) {
            if (em[d] == 0ull) continue;
            const McLevel m = mc_level(lvl, d);
            const int4 gq = *reinterpret_cast<const int4*>(lvg + d * 8);                 // grad, sX, sY
            float* gp = *reinterpret_cast<float* const*>(lvg + d * 8);
            const int sZ_ = lvg[d * 8 + 4];
            unsigned long long e_ = em[d];
            while (e_) {
              const int jj = 16 * wave + (int)__builtin_ctzll(e_);
              e_ &= e_ - 1;
              const int lx = 4 * (jj & 3) + a_vx, ly = 4 * ((jj >> 2) & 3) + a_vy, lz = ZS * (jj >> 4) + a_vz;
              if (lx < m.E[0] && ly < m.E[1] && lz < m.E[2])
                *reinterpret_cast<float4*>(gp + (m.v0[2] + lz) * sZ_ + (m.v0[1] + ly) * gq.w + (m.v0[0] + lx) * gq.z + a_c0) =
                    make_float4(0.f, 0.f, 0.f, 0.f);
            }
          }
        }
        MC_STAMP(7)
        __syncthreads();
        MC_STAMP(8)

        // ---- (3) multiply: this wavefront's share of the pool ------------------------------------------------------
        const int G = (pk.debug & 8) ? 0 : ptot >> 2;
        const int g_begin = min(G, __builtin_amdgcn_readfirstlane(misc[M_BOUND + wave])),
                  g_end = min(G, __builtin_amdgcn_readfirstlane(misc[M_BOUND + wave + 1]));
        int first_code = -1, last_code = -1;
        bool sh_before = false, sh_after = false;
        if (g_begin < g_end) {
          first_code = __builtin_amdgcn_readfirstlane((int)(pool[4 * g_begin] >> 16));
          last_code = __builtin_amdgcn_readfirstlane((int)(pool[4 * g_end - 4] >> 16));
          sh_before = g_begin > 0 && __builtin_amdgcn_readfirstlane((int)(pool[4 * g_begin - 4] >> 16)) == first_code;
          sh_after = g_end < G && __builtin_amdgcn_readfirstlane((int)(pool[4 * g_end] >> 16)) == last_code;
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, part_a = {0.f, 0.f, 0.f, 0.f};
        int cur_code = -1;
        bool in_first = true;           // still inside the share's first sub-brick
        const bool add_mode = !(first && pk.overwrite);

        // stores the accumulator tile of sub-brick `code` (or adds it: later epochs, overwrite off).  The pool is ordered
        // by level first, so the level's constants are refreshed a handful of times per share.
        int fl_d = -1, fl_sX = 0, fl_sY = 0, fl_sZ = 0, fl_e0 = 0, fl_e1 = 0, fl_e2 = 0, fl_lane = 0;
        bool fl_full = false;           // every sub-brick of the level is whole: nothing to mask
        float* fl_grad = nullptr;
        unsigned char* fl_touched = nullptr;
        auto store_tile = [&](int code, const f32x4& v) {
          const int d = code >> 7;
          if (d != fl_d) {
            fl_d = d;
            const McLevel m = mc_level(lvl, d);
            const int4 g0_ = *reinterpret_cast<const int4*>(lvg + d * 8), g1_ = *reinterpret_cast<const int4*>(lvg + d * 8 + 4);
            fl_sX = __builtin_amdgcn_readfirstlane(g0_.z); fl_sY = __builtin_amdgcn_readfirstlane(g0_.w);
            fl_sZ = __builtin_amdgcn_readfirstlane(g1_.x);
            fl_grad = reinterpret_cast<float*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane(g0_.y) << 32) |
                                               (unsigned)__builtin_amdgcn_readfirstlane(g0_.x));
            fl_touched = reinterpret_cast<unsigned char*>(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane(g1_.w) << 32) |
                                                          (unsigned)__builtin_amdgcn_readfirstlane(g1_.z));
            fl_e0 = __builtin_amdgcn_readfirstlane(m.E[0]); fl_e1 = __builtin_amdgcn_readfirstlane(m.E[1]);
            fl_e2 = __builtin_amdgcn_readfirstlane(m.E[2]);
            fl_full = !(fl_e0 & 3) && !(fl_e1 & 3) && !(fl_e2 & (ZS - 1));
            fl_lane = (__builtin_amdgcn_readfirstlane(m.v0[2]) + a_vz) * fl_sZ +
                      (__builtin_amdgcn_readfirstlane(m.v0[1]) + a_vy) * fl_sY +
                      (__builtin_amdgcn_readfirstlane(m.v0[0]) + a_vx) * fl_sX + a_c0;
          }
          const int sx = code & 3, sy = (code >> 2) & 3, sz = (code >> 4) & 7;
          if (!fl_full && (a_vx >= fl_e0 - 4 * sx || a_vy >= fl_e1 - 4 * sy || a_vz >= fl_e2 - ZS * sz)) return;
          const int eo = fl_lane + ZS * sz * fl_sZ + 4 * sy * fl_sY + 4 * sx * fl_sX;
          // (pointers rebuilt from LDS words are generic to the compiler: say they are global, or every store is a flat one
          // that also holds the LDS counter)
          typedef __attribute__((address_space(1))) f32x4 gf32x4;
          typedef __attribute__((address_space(1))) unsigned char gu8;
          gf32x4* dst = (gf32x4*)(fl_grad + eo);
          f32x4 o = v;
          if (fl_touched && (o[0] != 0.0f || o[1] != 0.0f || o[2] != 0.0f || o[3] != 0.0f))
            *(gu8*)(fl_touched + (eo >> ADAM_CHUNK_SHIFT)) = 1;
          if (add_mode) {
            o += *dst;
            *dst = o;
          } else {
            // a STREAMING store (the `nt` bit): this gradient is read once, by the optimizer (itself with streaming loads,
            // adam.hip) or by nobody (a step without one), and as ordinary stores its 76 MB per step at cfg-2 displace the
            // feature grids the next forward gathers: headline step 143.6 -> 142.1 us, cfg-2 trainer step 245 -> 241.5, three
            // runs each way twice (tools/experiments/pull_nt_store_r5.sh.txt).  As assembly: __builtin_nontemporal_store
            // does not survive here -- the two branches' stores are sunk into one and the mark is dropped.
            // (s_nop: a > 64-bit VMEM store followed by a VALU write of its data registers needs a wait state, and the
            // backend's hazard recogniser does not look inside inline assembly -- ADVICE r5)
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 0" ::"v"(dst), "v"(o) : "memory");
          }
        };

        // one staged pair per lane, in registers until the previous chunk's MFMAs are through with the LDS area.  Two
        // steps, so that the LDS round trips of the next chunk run under the MFMAs of this one: (a) which pair, (b) its
        // table entry and level, the request for its d-feat row, the twelve weights.
        float rw[12], rd[C];
        int rcode = 0, nslot = MC_CAND;
        const __amdgpu_buffer_rsrc_t drows =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk.dfeat), 0, (int)pk.drow_bytes, 0x00020000);
        auto stage_a = [&](int g0, int ng) {
          const int q = 4 * g0 + lane;
          const bool live = lane < 4 * ng;
          const unsigned e = live ? pool[q] : 0xFFFFFFFFu;
          nslot = min((int)(e & 0xffffu), MC_CAND);                // padding (0xFFFF) and idle lanes: the null entry
          // the group's code is its first pair's (always a real entry); idle groups of the last chunk continue its last item
          rcode = __builtin_amdgcn_mov_dpp(live ? (int)(e >> 16) : last_code, 0x00 /* quad_perm [0,0,0,0] */, 0xf, 0xf, true);
        };
        auto stage_l = [&]() {      // the d-feat row through a buffer load: the null entry's row -1 is out of range, reads as zeros
          int row = __float_as_int(smem[L::O_CAND + 4 * nslot + 3]);
          const int foff = lvl[(rcode >> 7) * MC_LVL + 3];
          if (pk.perm && row >= 0) row = pk.perm[row];
          const int boff = row * (int)(pk.ld * 4) + foff * 4;
          typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int c = 0; c < C; c += 4) {
            const u32x4 t4 = __builtin_amdgcn_raw_buffer_load_b128(drows, boff + 4 * c, 0, 0);
            rd[c] = __uint_as_float(t4[0]); rd[c + 1] = __uint_as_float(t4[1]);
            rd[c + 2] = __uint_as_float(t4[2]); rd[c + 3] = __uint_as_float(t4[3]);
          }
        };
        auto stage_w = [&]() {
          // per axis the sample's position relative to the sub-brick's first vertex, t; the weight of slot o is the hat
          // function max(0, 1 - |t - o|): (1 - frac) at the base corner, frac at the next one, 0 elsewhere (the null
          // entry is 1e30 away from everything).  pos op for op as common.hpp:axis_coord.
          const float4 c4 = cand[nslot];
          const int d = rcode >> 7;
          const int4 r0 = *reinterpret_cast<const int4*>(lvl + d * MC_LVL);          // X Y Z foff
          const int4 r1 = *reinterpret_cast<const int4*>(lvl + d * MC_LVL + 4);      // v0
          const float xs[3] = {c4.x, c4.y, c4.z};
          const int sz3[3] = {r0.x, r0.y, r0.z}, vv0[3] = {r1.x, r1.y, r1.z};
          const int sb[3] = {(rcode & 3) << 2, rcode & 12, ((rcode >> 4) & 7) << ZSH};
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const float pos = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xs[a], 1.0f), (float)sz3[a]), 1.0f), 0.5f);
            const float t = __fsub_rn(pos, (float)(vv0[a] + sb[a]));
            const int S = (a == 2) ? ZS : 4;
#pragma unroll
            for (int o = 0; o < 4; ++o)
              rw[a * 4 + o] = (o < S) ? __builtin_amdgcn_fmed3f(1.0f - fabsf(t - (float)o), 0.0f, 1.0f) : 0.0f;
          }
        };
        // two groups (one 8-byte read per operand component) of the chunk in LDS
#define MC_DREAD(P, X, Y, Z, D)                                                                   \
        {                                                                                         \
          X = *reinterpret_cast<const float2*>(stg + rd_x + 2 * (P));                             \
          Y = *reinterpret_cast<const float2*>(stg + rd_y + 2 * (P));                             \
          Z = *reinterpret_cast<const float2*>(stg + rd_z + 2 * (P));                             \
          D = *reinterpret_cast<const float2*>(stg + rd_d + 2 * (P));                             \
        }
#define MC_GROUP(GI, AV, BV)                                                                      \
        {                                                                                         \
          if ((chg >> (4 * (GI))) & 1ull) {         /* a new sub-brick starts with this group */  \
            const int code = __builtin_amdgcn_readlane(codev, 4 * (GI));                          \
            if (cur_code >= 0) {                                                                  \
              if (in_first && sh_before) part_a = acc;                                            \
              else store_tile(cur_code, acc);                                                     \
              in_first = false;                                                                   \
            }                                                                                     \
            cur_code = code;                                                                      \
            acc = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
          }                                                                                       \
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(AV, BV, acc, 0, 0, 0);                       \
        }
#define MC_DUO(P, X, Y, Z, D)                                                                     \
        {                                                                                         \
          MC_GROUP(2 * (P) + 0, Z.x * D.x, X.x * Y.x)                                             \
          MC_GROUP(2 * (P) + 1, Z.y * D.y, X.y * Y.y)                                             \
        }

        if (g_begin < g_end) { stage_a(g_begin, min(16, g_end - g_begin)); stage_l(); stage_w(); }
        for (int g0 = g_begin; g0 < g_end; g0 += 16) {
          const int ng = min(16, g_end - g0);
#pragma unroll
          for (int o = 0; o < 8 + ZS; ++o) stg[o * MC_CS + wr_p] = rw[o];
          MC_STAMP(9)
#pragma unroll
          for (int c = 0; c < C; ++c) stg[(8 + ZS + c) * MC_CS + wr_p] = rd[c];
          const int codev = rcode;
          // where the sub-brick changes inside the chunk: one ballot per chunk instead of a lane read + compare per group
          // (bit 4 g: group g's code differs from its predecessor's; group 0's predecessor is the running sub-brick)
          const int prevc = __shfl_up(codev, 4);
          const unsigned long long chg = __ballot((lane & 3) == 0 && codev != (lane < 4 ? cur_code : prevc));
          wave_sync_lds();
          MC_STAMP(12)
          const bool has_next = g0 + 16 < g_end;
          // two duos of operands live at a time (ping-pong); the scheduling barriers keep the compiler from hoisting
          // every read to the top (registers)
          float2 x0, y0, z0, d0, x1, y1, z1, d1;
          MC_DREAD(0, x0, y0, z0, d0)
          MC_DREAD(1, x1, y1, z1, d1)
          if (has_next) stage_a(g0 + 16, min(16, g_end - g0 - 16));
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(0, x0, y0, z0, d0)
          MC_DREAD(2, x0, y0, z0, d0)
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(1, x1, y1, z1, d1)
          MC_DREAD(3, x1, y1, z1, d1)
          if (has_next) stage_l();
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(2, x0, y0, z0, d0)
          MC_DREAD(4, x0, y0, z0, d0)
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(3, x1, y1, z1, d1)
          MC_DREAD(5, x1, y1, z1, d1)
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(4, x0, y0, z0, d0)
          MC_DREAD(6, x0, y0, z0, d0)
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(5, x1, y1, z1, d1)
          MC_DREAD(7, x1, y1, z1, d1)
          __builtin_amdgcn_sched_barrier(0);
          MC_DUO(6, x0, y0, z0, d0)
          __builtin_amdgcn_sched_barrier(0);
          MC_STAMP(13)
          if (has_next) stage_w();
          __builtin_amdgcn_sched_barrier(0);
          MC_STAMP(14)
          MC_DUO(7, x1, y1, z1, d1)
          MC_STAMP(15)
          wave_sync_lds();
        }
#undef MC_DREAD
#undef MC_GROUP
#undef MC_DUO
        MC_STAMP(9)
        // the share's last sub-brick
        int a_code = -1, b_code = -1, cont = 0;
        if (cur_code >= 0) {
          if (in_first && sh_before) { part_a = acc; a_code = cur_code; cont = sh_after ? 1 : 0; }
          else if (sh_after) b_code = cur_code;
          else store_tile(cur_code, acc);
          if (!in_first && sh_before) a_code = first_code;
        }
        // partial tiles into this wavefront's own staging area: [0, 256) the first sub-brick's, [256, 512) the last's
        if (a_code >= 0) *reinterpret_cast<float4*>(stg + lane * 4) = make_float4(part_a[0], part_a[1], part_a[2], part_a[3]);
        if (lane == 0) {
          misc[M_A + wave] = a_code; misc[M_B + wave] = b_code; misc[M_CONT + wave] = cont;
          misc[M_EMPTY + wave] = (g_begin >= g_end) ? 1 : 0;
        }
        __syncthreads();
        MC_STAMP(10)
        if (b_code >= 0) {   // a sub-brick that starts here and runs on: sum the pieces in wavefront order
          f32x4 tot = acc;
          for (int v = wave + 1; v < MC_WAVES; ++v) {
            if (__builtin_amdgcn_readfirstlane(misc[M_EMPTY + v])) continue;
            if (__builtin_amdgcn_readfirstlane(misc[M_A + v]) != b_code) break;
            const float4 q = *reinterpret_cast<const float4*>(smem + L::O_STAGE + v * L::STAGE + lane * 4);
            tot[0] += q.x; tot[1] += q.y; tot[2] += q.z; tot[3] += q.w;
            if (!__builtin_amdgcn_readfirstlane(misc[M_CONT + v])) break;
          }
          store_tile(b_code, tot);
        }
        first = false;
        lo = hi; hi = ncand;
        if (lo < ncand) {      // (rare: the pool overflowed) the next range recounts
          __syncthreads();
          for (int i = threadIdx.x; i < MC_ITEMS; i += NT) cnt[i] = 0;
          for (int i = threadIdx.x; i < MC_POOL; i += NT) pool[i] = 0xFFFFFFFFu;
        }
        __syncthreads();
        MC_STAMP(11)
      }
      if (!more) break;
    }
  }
  __syncthreads();
  if (pk.prof && threadIdx.x < 16)
    atomicAdd(&pk.prof[((blockIdx.x + blockIdx.y * 5 + blockIdx.z * 11) & 63) * 16 + threadIdx.x], (unsigned long long)misc[M_PROF + threadIdx.x]);
}

// true when the matrix-core kernel can take the pull of these levels (else grad_pull.hip's kernels)
bool mc_pull_ok(const GridK& g, int C, const int T[3], uint32_t level_mask, int64_t n, int64_t ld) {
  const char* env = getenv("MISO_PULL_MC");          // dev / tests: MISO_PULL_MC=0 keeps the vector kernels of grad_pull.hip
  const bool off = env && atoi(env) == 0;
  if (off || (C != 4 && C != 8)) return false;
  if (n <= 0 || ld <= 0 || n * ld * 4 >= (1ll << 31)) return false;        // d-feat rows are addressed by 32-bit offsets
  int nl = 0;
  for (int l = 0; l < g.n_levels; ++l)
    if ((level_mask >> l) & 1u) {
      const LevelK& lv = g.lv[l];
      const int size[3] = {lv.X, lv.Y, lv.Z};
      if (lv.C != C || lv.sC != 1) return false;
      for (int a = 0; a < 3; ++a) {
        if (3 * size[a] < 2 * T[a]) return false;                // cells much wider than tiles: the sweep (<= 7 tile rows per axis,
                                                                 // 64 in all: the lane table of row bounds) would not cover them
        if ((size[a] + T[a] - 1) / T[a] > PULL_BMAX) return false;
        if ((int64_t)size[a] * T[a] >= (1 << 28)) return false;
      }
      if (++nl > MC_MAXL) return false;
    }
  return nl > 0;
}

hipError_t launch_grad_pull_mc(const GridK& g, int C, const int T[3], const int* tile_off, const float* xn,
                               const float* dfeat, int64_t ld, const int* perm, uint32_t level_mask, int overwrite,
                               int64_t n, hipStream_t s) {
  McK pk;
  memset(&pk, 0, sizeof(pk));
  for (int a = 0; a < 3; ++a) { pk.T[a] = T[a]; pk.nb[a] = (T[a] + 1) / 2; }
  pk.tile_off = tile_off; pk.xn = reinterpret_cast<const float4*>(xn); pk.dfeat = dfeat; pk.perm = perm; pk.ld = ld;
  for (int l = 0; l < g.n_levels; ++l)
    if ((level_mask >> l) & 1u) {
      const LevelK& lv = g.lv[l];
      McLv& o = pk.lv[pk.nl++];
      const int size[3] = {lv.X, lv.Y, lv.Z};
      o.grad = lv.grad; o.touched = lv.touched;
      o.X = lv.X; o.Y = lv.Y; o.Z = lv.Z; o.foff = lv.foff;
      o.sX = lv.sX; o.sY = lv.sY; o.sZ = lv.sZ;
      o.live = ((g.ignore_mask >> l) & 1u) ? 0 : 1;
      for (int a = 0; a < 3; ++a) {
        o.bdiv[a] = (size[a] % T[a] == 0) ? size[a] / T[a] : 0;
        o.inv_size[a] = 1.0f / (float)size[a];
      }
    }
  pk.overwrite = overwrite;
  pk.drow_bytes = (unsigned int)(n * ld * 4);
  const bool small = getenv("MISO_MC_SMALL") != nullptr;      // tests: epochs and halved ranges on ordinary batches
  pk.cand_cap = small ? 192 : MC_CAND;
  pk.pool_cap = small ? 640 : MC_POOL;
  static const int dbg = [] { const char* e = getenv("MISO_DEBUG_PULL"); return e ? atoi(e) : 0; }();   // dev ablation
  pk.debug = dbg;
  // dev: MISO_MC_PROF=1 prints the clocks wavefront 0 of every workgroup spent per phase, averaged, every 64 launches
  static unsigned long long* prof = [] {
    unsigned long long* p = nullptr;
    if (getenv("MISO_MC_PROF") && hipMalloc(&p, 1024 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemset(p, 0, 1024 * 8);
    return p;
  }();
  pk.prof = prof;
  pk.prof_wave = prof ? atoi(getenv("MISO_MC_PROF")) - 1 : 0;
  void (*k)(McK) = nullptr;
  size_t words = 0;
#define PICK(c, n)                                                      \
  if (C == c && ((pk.nl <= 2 && n == 2) || pk.nl == n)) {               \
    k = grad_pull_mc_kernel<c, n>;                                      \
    words = McLds<c, n>::WORDS;                                         \
  }
  PICK(8, 2) PICK(8, 3) PICK(8, 4) PICK(4, 2) PICK(4, 3) PICK(4, 4)
#undef PICK
  if (!k) return hipErrorInvalidValue;
  const size_t lds = words * sizeof(float);
  hipError_t e = allow_dynamic_lds((const void*)k, lds);
  if (e != hipSuccess) return e;
  k<<<dim3((unsigned)pk.nb[0], (unsigned)pk.nb[1], (unsigned)pk.nb[2]), 64 * MC_WAVES, lds, s>>>(pk);
  if (prof) {
    static int calls = 0;
    if (++calls % 64 == 0) {
      static unsigned long long h[1024];
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(h, prof, sizeof(h), hipMemcpyDeviceToHost);
      (void)hipMemset(prof, 0, sizeof(h));
      static const char* nm[16] = {"setup", "sweep", "barrier", "count", "barrier", "prefix+b", "fill", "zero", "barrier",
                                   "m:write-w", "tail+b", "combine+b", "m:write-d+sync", "m:duo0-6+loads", "m:weights", "m:duo7"};
      fprintf(stderr, "[mc prof] clocks per workgroup:");
      for (int i = 0; i < 16; ++i) {
        double t = 0.0;
        for (int b = 0; b < 64; ++b) t += (double)h[b * 16 + i];
        fprintf(stderr, " %s %.0f", nm[i], t / (64.0 * pk.nb[0] * pk.nb[1] * pk.nb[2]));
      }
      fprintf(stderr, "\n");
    }
  }
  return hipGetLastError();
}

}  // namespace miso
