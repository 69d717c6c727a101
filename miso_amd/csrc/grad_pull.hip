// Owner-computes grid gradient ("pull"): the second half of the binned backward.
//
// The scatter formulation of the reference (one atomic per point, corner and channel,
// third_party/cuda_gridsample_grad2/gridsample_cuda.cu:466-481; ATen does the same) is bound
// on MI355X by the L2's fp32 atomic request rate (~21 G requests/s, tools/ubench/atomics.hip)
// and ds_add_f32 in LDS serialises (tools/ubench/lds_atomics.hip).  With the batch binned
// into T^3 spatial tiles (sort.hip) the sum can be turned around:
//
//   every tile OWNS the vertices of every level that lie inside it; one wavefront per tile
//   (1) sweeps the points of its 3x3x3 tile neighbourhood and compacts those whose cell
//       touches an owned vertex into a list (ballot + prefix popcount: deterministic order),
//   (2) bins the list by cell with an LDS counting sort (integer LDS atomics are full rate),
//       staging (frac_x, frac_y, frac_z) and the level's d-feat row next to each other,
//   (3) lets every lane PULL: lane = owned vertex, loop over its 8 adjacent cells' short
//       lists, accumulate w * d-feat in registers -- no atomics, no write conflicts,
//   (4) writes each owned vertex exactly once with plain 16-B stores (dense, coalesced).
//
// So the gradient needs no zero-fill, no atomics, and is bit-reproducible for a given
// sorted order.  d-feat rows come from sdf_bwd_kernel (dfeat_out); normalised coordinates
// from sort_scatter_kernel (xn_sorted), so no division is repeated here.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "common.hpp"
#include "grad_pull.hpp"

namespace miso {

// Owned brick of one (tile, level): wave-uniform scalars.
struct Brick {
  int v0[3], B[3];
  int nverts;
};

// bd: PullK::bdiv of the level.  With T | size the brick is (t * size/T, size/T): no per-tile
// integer divisions (they are emulated, ~25 instructions each, and every wave pays them per tile).
__device__ __forceinline__ Brick make_brick(const LevelK& lv, int ta, int tb, int tc, int T, const int* bd) {
  Brick b;
  if (bd[0] && bd[1] && bd[2]) {
    b.v0[0] = ta * bd[0]; b.v0[1] = tb * bd[1]; b.v0[2] = tc * bd[2];
    b.B[0] = bd[0]; b.B[1] = bd[1]; b.B[2] = bd[2];
    b.nverts = bd[0] * bd[1] * bd[2];
    return b;
  }
  b.v0[0] = ta * lv.X / T; b.v0[1] = tb * lv.Y / T; b.v0[2] = tc * lv.Z / T;
  b.B[0] = (ta + 1) * lv.X / T - b.v0[0]; b.B[1] = (tb + 1) * lv.Y / T - b.v0[1];
  b.B[2] = (tc + 1) * lv.Z / T - b.v0[2];
  b.nverts = b.B[0] * b.B[1] * b.B[2];
  return b;
}

// Bricks of at most 2 x 2 x 2 vertices (the coarsest level of a pyramid over 16 tiles: cfg-2's 32^3) under ~200
// candidates: nothing is binned.  Lane = candidate computes its weight for each of the eight owned vertices (zero for a
// vertex its cell does not touch -- the list is conservative) and fetches its d-feat row; both go to LDS transposed
// ([vertex][candidate], [channel][candidate], rows padded to 68 words: conflict-free 16-B reads); then lane =
// (vertex, channel) -- 64 outputs for C = 8; for C = 4 the two half-waves split the candidates -- sums its row pair
// four candidates per step.  All d-feat rows of up to 256 candidates are requested before the first is used: one
// global round trip per tile instead of one per staged group, no counting sort, no byte counters, no scans.
constexpr int TINY_ROW = 68;
template <int C, int MODE, bool BLK>
__device__ __forceinline__ void pull_tiny(const GridK& g, const PullK& pk, const LevelK& lv, const Brick& b,
                                          float* smem, int o_list, int n, int o_stage, int lane, int add,
                                          const uint16_t* blk_list, const float4* cand) {
  int* ismem = reinterpret_cast<int*>(smem);
  float* wbuf = smem + o_stage;                 // [8][TINY_ROW]
  float* dbuf = wbuf + 8 * TINY_ROW;            // [C][TINY_ROW]
  constexpr int SUPER = (C == 8) ? 2 : 4;       // chunks of 64 candidates whose rows are in flight together
  const int v = (C == 8) ? (lane >> 3) : ((lane >> 2) & 7), ch = (C == 8) ? (lane & 7) : (lane & 3);
  const int half = (C == 8) ? 0 : (lane >> 5);  // C = 4: candidates [32 half, 32 half + 32) of every chunk
  float acc = 0.0f;
  for (int s0 = 0; s0 < max(n, 1); s0 += 64 * SUPER) {
    float4 c4[SUPER];
    float dv[SUPER][C];
#pragma unroll
    for (int u = 0; u < SUPER; ++u) {
      const int i = s0 + u * 64 + lane;
      c4[u] = make_float4(2e30f, 2e30f, 2e30f, 0.f);
#pragma unroll
      for (int c = 0; c < C; ++c) dv[u][c] = 0.0f;
      if (i < n) {
        int p;
        if (BLK) {
          c4[u] = cand[blk_list[i]];
          p = __float_as_int(c4[u].w);
        } else {
          p = ismem[o_list + i];
          c4[u] = pk.xn[p];
        }
        const int row = pk.perm ? pk.perm[p] : p;
        c4[u].w = __int_as_float(row);
        const float* src = pk.dfeat + (int64_t)row * pk.ld + lv.foff;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
          const float4 t = *reinterpret_cast<const float4*>(src + c);
          dv[u][c] = t.x; dv[u][c + 1] = t.y; dv[u][c + 2] = t.z; dv[u][c + 3] = t.w;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SUPER; ++u) {
      if (s0 + u * 64 >= n) break;              // wave-uniform
      // per-axis weight of the two owned vertex planes (and, MODE 1, its derivative)
      int i0[3]; float fr[3];
      cell_of(c4[u].x, lv.X, i0[0], fr[0]); cell_of(c4[u].y, lv.Y, i0[1], fr[1]); cell_of(c4[u].z, lv.Z, i0[2], fr[2]);
      float w[3][2], dw[3][2];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int l = 0; l < 2; ++l) {
          const int V = b.v0[a] + l;
          const bool own = l < b.B[a], lo = V == i0[a], hi = V == i0[a] + 1;
          w[a][l] = own ? (lo ? 1.0f - fr[a] : (hi ? fr[a] : 0.0f)) : 0.0f;
          dw[a][l] = own ? (lo ? -1.0f : (hi ? 1.0f : 0.0f)) : 0.0f;
        }
      float e[3] = {0.f, 0.f, 0.f};
      if (MODE && s0 + u * 64 + lane < n) {
        const float* eg = pk.ggx + (int64_t)__float_as_int(c4[u].w) * 3;
        e[0] = eg[0] * (g.gscale[0] * (0.5f * (float)lv.X));
        e[1] = eg[1] * (g.gscale[1] * (0.5f * (float)lv.Y));
        e[2] = eg[2] * (g.gscale[2] * (0.5f * (float)lv.Z));
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int lx = k & 1, ly = (k >> 1) & 1, lz = k >> 2;
        float wk = (w[0][lx] * w[1][ly]) * w[2][lz];
        if (MODE)
          wk = e[0] * (dw[0][lx] * w[1][ly] * w[2][lz]) + e[1] * (dw[1][ly] * w[0][lx] * w[2][lz]) +
               e[2] * (dw[2][lz] * w[0][lx] * w[1][ly]);
        wbuf[k * TINY_ROW + lane] = wk;
      }
#pragma unroll
      for (int c = 0; c < C; ++c) dbuf[c * TINY_ROW + lane] = dv[u][c];
      wave_sync_lds();
      if (!(pk.debug & 1)) {
        constexpr int STEPS = (C == 8) ? 16 : 8;
#pragma unroll
        for (int q = 0; q < STEPS; ++q) {
          const float4 ww = *reinterpret_cast<const float4*>(wbuf + v * TINY_ROW + 32 * half + 4 * q);
          const float4 dd = *reinterpret_cast<const float4*>(dbuf + ch * TINY_ROW + 32 * half + 4 * q);
          acc += ww.x * dd.x; acc += ww.y * dd.y; acc += ww.z * dd.z; acc += ww.w * dd.w;
        }
      }
      wave_sync_lds();
    }
  }
  if (C == 4) acc += __shfl_xor(acc, 32);
  // ---- store: lane (v, ch); C = 4: the lower half-wave ------------------------------------------------------------
  const int lx = v & 1, ly = (v >> 1) & 1, lz = v >> 2;
  if (!lv.grad || lx >= b.B[0] || ly >= b.B[1] || lz >= b.B[2] || (C == 4 && lane >= 32)) return;
  float* dst = lv.grad + (b.v0[2] + lz) * lv.sZ + (b.v0[1] + ly) * lv.sY + (b.v0[0] + lx) * lv.sX + ch;
  if (lv.touched && acc != 0.0f) lv.touched[(dst - lv.grad) >> ADAM_CHUNK_SHIFT] = 1;
  if (add == 2) {
    if (acc != 0.0f) atomic_add_f32(dst, acc);
  } else {
    *dst = add ? *dst + acc : acc;
  }
}

// Bin `n` listed candidates of one level by cell, pull them into the brick's vertices and store
// the brick (steps 2-4 of the file comment).  `add`: accumulate onto what is already stored.
//
// Per-cell counters are BYTES (a group stages at most PULL_CAP < 256 records, so counts, prefixes
// and cursors all fit): 4 cells per LDS word, atomics on the containing word, plain ds_read_u8 for
// lookups.  Vertices are pulled in blocks of PULL_RB rounds of 64 so that only PULL_RB accumulator
// rows are live.  Both keep a wavefront at <= 128 VGPRs and 10 KB of LDS: 4 blocks per CU, i.e. all
// 4096 tiles of a 16^3 binning resident at once -- the kernel is bound by the serial latency of one
// tile (~55 us whatever the occupancy), so throughput is the number of resident wavefronts.
//
// MODE 0: vertex += w(corner) * row, w the trilinear weight -- the grid gradient of the first
// backward.  MODE 1: vertex += (sum_a e_a * d w(corner) / d ix_a) * row with e = gg_x (.) d ix / d x per
// point -- the grid gradient of the SECOND backward (g_input of gridsample_grad2.grad2_3d,
// third_party/cuda_gridsample_grad2/gridsample_cuda.cu:462-481); records carry e next to the fracs.
// BLK: the candidate list holds 16-bit slots of the block-wide survivor table (grad_pull_block_kernel: `blk_list`,
// `cand_xn`, `cand_p`) instead of point indices in LDS at smem + o_list.
template <int C, int MODE, bool BLK = false>
__device__ __forceinline__ void pull_level(const GridK& g, const PullK& pk, const LevelK& lv, const Brick& b,
                                           float* smem, int o_list, int n, int o_arr, int o_rec, int o_df,
                                           int lane, int add,   // add: 0 store, 1 read-add-store, 2 atomic add
                                           const uint16_t* blk_list = nullptr, const float4* cand = nullptr) {
  if (b.B[0] <= 2 && b.B[1] <= 2 && b.B[2] <= 2 && !(pk.debug & 256)) {
    pull_tiny<C, MODE, BLK>(g, pk, lv, b, smem, o_list, n, o_arr, lane, add, blk_list, cand);
    wave_sync_lds();   // the caller reuses the list and the staging area
    return;
  }
  int* ismem = reinterpret_cast<int*>(smem);
  unsigned* arrw = reinterpret_cast<unsigned*>(smem) + o_arr;
  const unsigned char* arrb = reinterpret_cast<const unsigned char*>(arrw);
  const int vx0 = b.v0[0], vy0 = b.v0[1], vz0 = b.v0[2], Bx = b.B[0], By = b.B[1];
  const int nverts = b.nverts;
  const int ncx = Bx + 1, ncy = By + 1, ncz = b.B[2] + 1, ncells = ncx * ncy * ncz;
  // lane roles.  Bricks of <= 8 vertices: 8 lanes per vertex, one adjacent cell each, reduced
  // at the end; otherwise one lane per vertex in rounds of 64 vertices.
  const bool lpv8 = nverts <= 8;
  const int nrounds = lpv8 ? 1 : (nverts + 63) / 64;
  // exact division of indices < 512 by B <= 8 through a 16-bit reciprocal
  const int inv_bx = (65536 + Bx - 1) / Bx, inv_by = (65536 + By - 1) / By;
  // Loop nest.  Levels whose brick fits one block of rounds (the coarse ones): accumulate over all
  // groups, store once.  Otherwise (the finest level): groups outside, round blocks inside, every
  // (group, block) stored -- overflow groups add onto the first one's store, nothing is restaged.
  constexpr int CAP = pull_cap(C, MODE);
  const int ngroups = (pk.debug & 2) ? 1 : max(1, (n + CAP - 1) / CAP);
  const int nrb = (nrounds + PULL_RB - 1) / PULL_RB;
  const bool single_rb = nrb == 1;
  const int npass = single_rb ? ngroups : ngroups * nrb;
  float acc[PULL_RB][C];

  for (int pass = 0; pass < npass; ++pass) {
    const int gi = single_rb ? pass : pass / nrb;
    const int rb0 = single_rb ? 0 : (pass - gi * nrb) * PULL_RB;
    if (!single_rb || pass == 0) {
#pragma unroll
      for (int r = 0; r < PULL_RB; ++r)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[r][c] = 0.0f;
    }
    {
      const int g0 = gi * CAP;
      const int gn = (pk.debug & 2) ? 0 : max(0, min(CAP, n - g0));
      if (single_rb || rb0 == 0) {
        // ---- stage group gi: count, scan, fill ---------------------------------------------------
        if (pass > 0) wave_sync_lds();     // the previous pass is done reading the staging area
        for (int i = lane; i < PULL_ARRW; i += 64) arrw[i] = 0u;
        wave_sync_lds();
        // (2a) exact cell of every listed candidate (the sweep's box test is conservative), count
        // per cell; the records stay in registers (<= 3 per lane)
        constexpr int RPL = (CAP + 63) / 64;
        int rc[RPL], rp[RPL]; float rfx[RPL], rfy[RPL], rfz[RPL];
        constexpr int REC = pull_rec(MODE);
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
          const int i = u * 64 + lane;
          rc[u] = -1;
          if (i < gn) {
            int p;
            float4 c4;
            if (BLK) {
              c4 = cand[blk_list[g0 + i]];
              p = __float_as_int(c4.w);
            } else {
              p = ismem[o_list + g0 + i];
              c4 = pk.xn[p];
            }
            int i0, j0, k0;
            cell_of(c4.x, lv.X, i0, rfx[u]); cell_of(c4.y, lv.Y, j0, rfy[u]); cell_of(c4.z, lv.Z, k0, rfz[u]);
            const int ci = i0 - (vx0 - 1), cj = j0 - (vy0 - 1), ck = k0 - (vz0 - 1);
            if ((unsigned)ci < (unsigned)ncx && (unsigned)cj < (unsigned)ncy && (unsigned)ck < (unsigned)ncz) {
              rc[u] = (ck * ncy + cj) * ncx + ci;
              rp[u] = p;
              atomicAdd(&arrw[rc[u] >> 2], 1u << (8 * (rc[u] & 3)));
            }
          }
        }
        wave_sync_lds();
        // (2b) exclusive scan of the byte counters in place: 3 words = 12 cells per lane, one wave scan
        {
          unsigned w[3];
          int tot = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            w[k] = arrw[lane * 3 + k];
            tot += (int)((w[k] & 255u) + ((w[k] >> 8) & 255u) + ((w[k] >> 16) & 255u) + (w[k] >> 24));
          }
          int inc = tot;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            int t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
          }
          unsigned run = (unsigned)(inc - tot);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const unsigned c0 = w[k] & 255u, c1 = (w[k] >> 8) & 255u, c2 = (w[k] >> 16) & 255u, c3 = w[k] >> 24;
            const unsigned e0 = run, e1 = e0 + c0, e2 = e1 + c1, e3 = e2 + c2;
            arrw[lane * 3 + k] = e0 | (e1 << 8) | (e2 << 16) | (e3 << 24);
            run = e3 + c3;
          }
        }
        wave_sync_lds();
        // (2c) fill: afterwards arr[c] = end of cell c = start of cell c+1
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
          if (rc[u] >= 0) {
            const int sh = 8 * (rc[u] & 3);
            const int pos = (int)((atomicAdd(&arrw[rc[u] >> 2], 1u << sh) >> sh) & 255u);
            if (!MODE) *reinterpret_cast<float4*>(smem + o_rec + pos * REC) = make_float4(rfx[u], rfy[u], rfz[u], 0.0f);
            const int row = pk.perm ? pk.perm[rp[u]] : rp[u];
            if (MODE) {
              // e_a = gg_x[a] * d ix_a / d x_a = gg_x[a] * (2 / len_a) * (size_a / 2)   (axis_coord's mult)
              const float* e = pk.ggx + (int64_t)row * 3;
              float2* rec2 = reinterpret_cast<float2*>(smem + o_rec + pos * REC);
              const float ex = e[0] * (g.gscale[0] * (0.5f * (float)lv.X)), ey = e[1] * (g.gscale[1] * (0.5f * (float)lv.Y)),
                          ez = e[2] * (g.gscale[2] * (0.5f * (float)lv.Z));
              rec2[0] = make_float2(rfx[u], rfy[u]); rec2[1] = make_float2(rfz[u], ex); rec2[2] = make_float2(ey, ez);
            }
            const float* src = pk.dfeat + (int64_t)row * pk.ld + lv.foff;
#pragma unroll
            for (int c = 0; c < C; c += 4)
              *reinterpret_cast<float4*>(smem + o_df + pos * C + c) = *reinterpret_cast<const float4*>(src + c);
          }
        }
        wave_sync_lds();
      }
      // (3) pull.  Cells c-1 and c (x-neighbours) are adjacent in the cell order, so their records
      // form ONE contiguous range: 4 loops (dy,dz) per vertex instead of 8.
      if (!(pk.debug & 1)) {
#pragma unroll
        for (int r = 0; r < PULL_RB; ++r) {
          if (rb0 + r >= nrounds) continue;
          const int vidx = lpv8 ? (lane >> 3) : (rb0 + r) * 64 + lane;
          if (vidx >= nverts) continue;
          const int t_ = (vidx * inv_bx) >> 16, lx = vidx - t_ * Bx;      // vidx / Bx, vidx % Bx
          const int lz = (t_ * inv_by) >> 16, ly = t_ - lz * By;
          const int c_hi = ((lz + 1) * ncy + (ly + 1)) * ncx + (lx + 1);   // cell whose corner (0,0,0) is this vertex
          if (lpv8) {
            // 8 lanes per vertex, one adjacent cell each: ONE loop per lane over its own cell's
            // records (the general path below would run its four (dy,dz) loops back to back with a
            // quarter of the lanes active in each)
            const int kk = lane & 7, dx = kk & 1, dy = (kk >> 1) & 1, dz = kk >> 2;
            const int cc_ = c_hi - (dz * ncy + dy) * ncx - dx;
            const int q0 = (cc_ >= 1) ? (int)arrb[cc_ - 1] : 0, q1 = (int)arrb[cc_];
            for (int q = q0; q < q1; ++q) {
              constexpr int REC = pull_rec(MODE);
              float4 f, e;
              if (MODE) {
                const float2* rec2 = reinterpret_cast<const float2*>(smem + o_rec + q * REC);
                const float2 r0 = rec2[0], r1 = rec2[1], r2 = rec2[2];
                f = make_float4(r0.x, r0.y, r1.x, 0.0f);
                e = make_float4(r1.y, r2.x, r2.y, 0.0f);
              } else {
                f = *reinterpret_cast<const float4*>(smem + o_rec + q * REC);
              }
              const float ux = dx ? f.x : 1.0f - f.x, uy = dy ? f.y : 1.0f - f.y, uz = dz ? f.z : 1.0f - f.z;
              float w = (ux * uy) * uz;
              if (MODE) {
                w = e.x * ((dx ? 1.0f : -1.0f) * uy * uz) + e.y * ((dy ? 1.0f : -1.0f) * ux * uz) +
                    e.z * ((dz ? 1.0f : -1.0f) * ux * uy);
              }
#pragma unroll
              for (int cc = 0; cc < C; cc += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(smem + o_df + q * C + cc);
                acc[r][cc + 0] += w * dv.x; acc[r][cc + 1] += w * dv.y;
                acc[r][cc + 2] += w * dv.z; acc[r][cc + 3] += w * dv.w;
              }
            }
            continue;
          }
          int b0[4], b1[4], b2[4];   // start of cell c-1, end of c-1 (= start of c), end of c
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int dy = k & 1, dz = k >> 1;
            const int c = c_hi - (dz * ncy + dy) * ncx;        // corner (dx=0, dy, dz); (dx=1) is c-1
            b0[k] = (c >= 2) ? (int)arrb[c - 2] : 0;
            b1[k] = (int)arrb[c - 1];
            b2[k] = (int)arrb[c];
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int dy = k & 1, dz = k >> 1;
            const int q0 = b0[k], q1 = b2[k];
            for (int q = q0; q < q1; ++q) {
              constexpr int REC = pull_rec(MODE);
              float4 f, e;
              if (MODE) {
                const float2* rec2 = reinterpret_cast<const float2*>(smem + o_rec + q * REC);
                const float2 r0 = rec2[0], r1 = rec2[1], r2 = rec2[2];
                f = make_float4(r0.x, r0.y, r1.x, 0.0f);
                e = make_float4(r1.y, r2.x, r2.y, 0.0f);
              } else {
                f = *reinterpret_cast<const float4*>(smem + o_rec + q * REC);
              }
              const bool dx1 = q < b1[k];                         // record of cell c-1 => corner dx = 1
              const float ux = dx1 ? f.x : 1.0f - f.x, uy = dy ? f.y : 1.0f - f.y, uz = dz ? f.z : 1.0f - f.z;
              float w = (ux * uy) * uz;
              if (MODE) {
                w = e.x * ((dx1 ? 1.0f : -1.0f) * uy * uz) + e.y * ((dy ? 1.0f : -1.0f) * ux * uz) +
                    e.z * ((dz ? 1.0f : -1.0f) * ux * uy);
              }
#pragma unroll
              for (int cc = 0; cc < C; cc += 4) {
                const float4 dv = *reinterpret_cast<const float4*>(smem + o_df + q * C + cc);
                acc[r][cc + 0] += w * dv.x; acc[r][cc + 1] += w * dv.y;
                acc[r][cc + 2] += w * dv.z; acc[r][cc + 3] += w * dv.w;
              }
            }
          }
        }
      }
    }
    // ---- (4) store ---------------------------------------------------------------------------------
    if (!lv.grad || (single_rb && pass != npass - 1)) continue;
    const int add_eff = add ? add : ((!single_rb && gi > 0) ? 1 : 0);
    if (lpv8) {   // reduce the 8 lanes of a vertex
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = acc[0][c];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        acc[0][c] = v;
      }
    }
#pragma unroll
    for (int r = 0; r < PULL_RB; ++r) {
      if (rb0 + r >= nrounds) continue;
      const int vidx = lpv8 ? (lane >> 3) : (rb0 + r) * 64 + lane;
      if (vidx >= nverts || (lpv8 && (lane & 7) != 0)) continue;
      const int t_ = (vidx * inv_bx) >> 16, lx = vidx - t_ * Bx;
      const int lz = (t_ * inv_by) >> 16, ly = t_ - lz * By;
      float* dst = lv.grad + (vz0 + lz) * lv.sZ + (vy0 + ly) * lv.sY + (vx0 + lx) * lv.sX;
#pragma unroll
      for (int c = 0; c < C; c += 4) {
        float4 v = make_float4(acc[r][c], acc[r][c + 1], acc[r][c + 2], acc[r][c + 3]);
        if (lv.touched && (v.x != 0.0f || v.y != 0.0f || v.z != 0.0f || v.w != 0.0f))
          lv.touched[(dst + c - lv.grad) >> ADAM_CHUNK_SHIFT] = 1;
        if (add_eff == 2) {   // a queued slice: other wavefronts add to the same brick
          if (v.x != 0.0f) atomic_add_f32(dst + c, v.x);
          if (v.y != 0.0f) atomic_add_f32(dst + c + 1, v.y);
          if (v.z != 0.0f) atomic_add_f32(dst + c + 2, v.z);
          if (v.w != 0.0f) atomic_add_f32(dst + c + 3, v.w);
          continue;
        }
        if (add_eff) {
          const float4 o = *reinterpret_cast<const float4*>(dst + c);
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        *reinterpret_cast<float4*>(dst + c) = v;
      }
    }
  }
  wave_sync_lds();   // the caller reuses the list and the staging area
}

// One wavefront per spatial tile.  The 3x3x3 tile neighbourhood is swept ONCE for all pulled
// levels (a candidate's coordinates are loaded once and tested against every level's box).
// DRAIN = false: one wavefront per tile (slice 0 of a cut tile, queueing the rest); true: the second launch that
// works off the queued slices -- a separate instantiation so that profiles tell the two apart.
// One tile, swept by ONE wavefront: the 3x3x3 tile neighbourhood is read row by row, every candidate tested against
// the catchment box of every pulled level, the hits compacted into per-level LDS lists, the lists binned / pulled /
// stored by pull_level.  The general path: any grid size, heavy tiles cut into slices (sl of ns; DRAIN: a queued
// slice, added atomically).  grad_pull_block_kernel below is the fast path for the common case.
template <int C, int NLV, int MODE, bool DRAIN>
__device__ __forceinline__ void pull_one_tile(const GridK& g, const PullK& pk, float* smem, int o_list, int o_arr,
                                              int o_rec, int o_df, int lane, unsigned long long lt_mask, int tile,
                                              int sl, int ns) {
  constexpr int PULL_NLV = NLV;
  int* ismem = reinterpret_cast<int*>(smem);
  const int T = pk.T;
  const int ta = tile % T, tb = (tile / T) % T, tc = tile / (T * T);
  const int tabc[3] = {ta, tb, tc};
  // catchment box of every level's brick in normalised coordinates: base corner i0 in
  // [v0-1, v1-1] <=> pos in [v0-1, v1) <=> xn in [(2 v0 - 1)/X - 1, (2 v1 + 1)/X - 1), widened by
  // a few ulps; and the union of the tile ranges that can hold such points
  float blo[PULL_NLV][3], bhi[PULL_NLV][3];
  int t_lo[3] = {T, T, T}, t_hi[3] = {-1, -1, -1};
  int nverts_all = 0;
#pragma unroll
  for (int d = 0; d < PULL_NLV; ++d) {
#pragma unroll
    for (int a = 0; a < 3; ++a) { blo[d][a] = 3e30f; bhi[d][a] = -3e30f; }   // empty box
    if (d >= pk.nl) continue;
    const int lvl = pk.lev[d];
    const LevelK& lv = g.lv[lvl];
    const Brick b = make_brick(lv, ta, tb, tc, T, pk.bdiv[d]);
    if (b.nverts == 0 || ((g.ignore_mask >> lvl) & 1u)) continue;
    nverts_all += b.nverts;
    const int size[3] = {lv.X, lv.Y, lv.Z};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      // the reciprocal is within an ulp of the quotient; the box is widened by ~30 ulps
      blo[d][a] = (2.0f * b.v0[a] - 1.0f) * pk.inv_size[d][a] - 1.0f - 8e-6f;
      bhi[d][a] = (2.0f * (b.v0[a] + b.B[a]) + 1.0f) * pk.inv_size[d][a] - 1.0f + 8e-6f;
      int lo, hi;
      if (pk.bdiv[d][a]) {   // size = B T: floor((tB - 1) / B) = t - 1, floor((tB + B + 1) / B) = t + 1 (+1 if B = 1)
        lo = tabc[a] - 1; hi = tabc[a] + (b.B[a] > 1 ? 1 : 2);
      } else {
        lo = floor_div((b.v0[a] - 1) * T, size[a]); hi = floor_div((b.v0[a] + b.B[a] + 1) * T, size[a]);
      }
      t_lo[a] = min(t_lo[a], max(0, lo));
      t_hi[a] = max(t_hi[a], min(T - 1, hi));
    }
  }
  int nlist[PULL_NLV];
#pragma unroll
  for (int d = 0; d < PULL_NLV; ++d) nlist[d] = 0;
  unsigned stored = 0;    // bit d: level d's brick has been stored once already
  bool sweeping = (t_hi[0] >= t_lo[0]) && !(pk.debug & 4);
  int tz = t_lo[2], ty = t_lo[1];
  int p_cur = 0, p_end = 0;
  // bounds of every row of tiles (fixed ty, tz), fetched lane-parallel up front: one memory
  // round trip per tile instead of a dependent scalar load in front of every row's sweep
  const int ny = t_hi[1] - t_lo[1] + 1;
  const int nrows = sweeping ? ny * (t_hi[2] - t_lo[2] + 1) : 0;
  const bool tabled = nrows <= 64;
  int rs = 0, re = 0;
  if (tabled && lane < nrows) {
    const int ry = t_lo[1] + lane % ny, rz = t_lo[2] + lane / ny;
    rs = pk.tile_off[(rz * T + ry) * T + t_lo[0]];
    re = pk.tile_off[(rz * T + ry) * T + t_hi[0] + 1];
  }
  if (pk.queue && tabled && sweeping) {
    if (!DRAIN) {
      int work = re - rs;
      for (int o = 32; o > 0; o >>= 1) work += __shfl_xor(work, o);
      const int slice = max(pk.work0, 16 * nverts_all);
      ns = __builtin_amdgcn_readfirstlane(min(PULL_NS_MAX, (work + slice - 1) / slice));
      if (ns > 1) {
        int pos = 0;
        if (lane == 0) pos = atomicAdd(&pk.queue[0], ns - 1);
        pos = __builtin_amdgcn_readfirstlane(pos);
        const bool fits = pos + ns - 1 <= pk.qcap;     // else: blank what was reserved, keep the tile whole
        for (int i = lane; i < ns - 1; i += 64)
          if (pos + i < pk.qcap) pk.queue[PULL_QHDR + pos + i] = fits ? (tile | ((i + 1) << 12) | (ns << 20)) : -1;
        if (!fits) ns = 1;
      }
    }
    if (ns > 1) {   // this wavefront's slice of every row range
      const int len = re - rs;
      const int a = rs + (int)(((int64_t)len * sl) / ns), b = rs + (int)(((int64_t)len * (sl + 1)) / ns);
      rs = a; re = b;
    }
  }
  const bool atomic = sl > 0;
  int ridx = 0;
  if (sweeping) {
    if (tabled) {
      p_cur = __builtin_amdgcn_readlane(rs, 0); p_end = __builtin_amdgcn_readlane(re, 0);
    } else {
      p_cur = pk.tile_off[(tz * T + ty) * T + t_lo[0]];
      p_end = pk.tile_off[(tz * T + ty) * T + t_hi[0] + 1];
    }
  }
  bool more = true;
  while (more) {
    // ---- (1) sweep: compact the contributing candidates per level (deterministic order) ------
    while (sweeping) {
      int room = PULL_LIST;
#pragma unroll
      for (int d = 0; d < PULL_NLV; ++d) room = min(room, PULL_LIST - nlist[d]);
      if (room < 64) break;                     // a list is nearly full: flush first
      if (p_cur >= p_end) {                     // next (ty, tz) row of tiles
        if (++ty > t_hi[1]) { ty = t_lo[1]; ++tz; }
        if (tz > t_hi[2]) { sweeping = false; break; }
        ++ridx;
        if (tabled) {
          p_cur = __builtin_amdgcn_readlane(rs, ridx); p_end = __builtin_amdgcn_readlane(re, ridx);
        } else {
          p_cur = pk.tile_off[(tz * T + ty) * T + t_lo[0]];
          p_end = pk.tile_off[(tz * T + ty) * T + t_hi[0] + 1];
        }
        continue;
      }
      // up to 4 steps (256 candidates) per trip, fixed BEFORE the loads so that all coordinate
      // loads are in flight together: a row of tiles costs one memory round trip
      constexpr int UN = 4;
      const int nstep = __builtin_amdgcn_readfirstlane(min(min(UN, (p_end - p_cur + 63) / 64), room / 64));
      float4 c4[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int p = p_cur + u * 64 + lane;
        c4[u] = make_float4(2e30f, 2e30f, 2e30f, 0.f);   // outside every box
        if (u < nstep && p < p_end) c4[u] = pk.xn[p];
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (u >= nstep) continue;    // wave-uniform
#pragma unroll
        for (int d = 0; d < PULL_NLV; ++d) {
          const bool inside = c4[u].x >= blo[d][0] && c4[u].x < bhi[d][0] && c4[u].y >= blo[d][1] &&
                              c4[u].y < bhi[d][1] && c4[u].z >= blo[d][2] && c4[u].z < bhi[d][2];
          const unsigned long long m = __ballot(inside);
          if (inside) ismem[o_list + d * PULL_LIST + nlist[d] + __popcll(m & lt_mask)] = p_cur + u * 64 + lane;
          nlist[d] = __builtin_amdgcn_readfirstlane(nlist[d] + (int)__popcll(m));
        }
      }
      p_cur += 64 * nstep;
    }
    wave_sync_lds();
    // ---- (2-4) per level: bin, pull, store --------------------------------------------------------
#pragma unroll 1
    for (int d = 0; d < pk.nl; ++d) {
      const int lvl = pk.lev[d];
      const LevelK& lv = g.lv[lvl];
      const Brick b = make_brick(lv, ta, tb, tc, T, pk.bdiv[d]);
      if (b.nverts == 0) continue;
      int n = 0;
#pragma unroll
      for (int e = 0; e < PULL_NLV; ++e) n = (e == d) ? nlist[e] : n;
      // mid-sweep flush: only the level whose list is (nearly) full is processed; the others keep
      // collecting, so a fine level's brick is normally pulled and stored once per tile
      if (sweeping && PULL_LIST - n >= 64) continue;
      const bool first = !((stored >> d) & 1u);
      if (n == 0 && (!first || atomic)) continue;
      pull_level<C, MODE>(g, pk, lv, b, smem, o_list + d * PULL_LIST, n, o_arr, o_rec, o_df, lane,
                    atomic ? 2 : ((first && pk.overwrite) ? 0 : 1));
      stored |= 1u << d;
#pragma unroll
      for (int e = 0; e < PULL_NLV; ++e) nlist[e] = (e == d) ? 0 : nlist[e];
    }
    more = sweeping;
  }
  (void)tabc;
}

// One wavefront per spatial tile.  The 3x3x3 tile neighbourhood is swept ONCE for all pulled
// levels (a candidate's coordinates are loaded once and tested against every level's box).
// DRAIN = false: one wavefront per tile (slice 0 of a cut tile, queueing the rest); true: the second launch that
// works off the queued slices -- a separate instantiation so that profiles tell the two apart.
template <int C, int NLV, int MODE, bool DRAIN>
__global__ __launch_bounds__(256) void grad_pull_kernel(GridK g, PullK pk) {
  constexpr int PULL_NLV = NLV;   // levels swept together
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // the wave index is uniform, but derived from threadIdx the compiler treats everything that
  // depends on it (tile, bricks, loop counters) as divergent: VGPRs, exec-mask juggling, no scalar
  // loads.  readfirstlane pins it (and the list lengths below) to SGPRs.
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int REC = pull_rec(MODE);
  constexpr int CAP = pull_cap(C, MODE);
  constexpr int PER_WAVE = PULL_NLV * PULL_LIST + PULL_ARRW + CAP * REC + CAP * C;
  const int o_list = wave * PER_WAVE, o_arr = o_list + PULL_NLV * PULL_LIST, o_rec = o_arr + PULL_ARRW,
            o_df = o_rec + CAP * REC;
  const int T = pk.T, ntiles = T * T * T;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  int qn = 0;
  if (DRAIN) {
    qn = min(pk.queue[0], pk.qcap);
    if (qn <= 0) return;          // nothing was split (the uniform case): no atomics, counters stay zero
  }
  for (int iter = 0;; ++iter) {
    int tile, sl = 0, ns = 1;
    if (!DRAIN) {
      tile = blockIdx.x * 4 + wave + iter * (int)gridDim.x * 4;
      if (tile >= ntiles) break;
    } else {
      // static round-robin over the queued slices (they are of similar size); grabbing them with an atomic
      // counter would serialise on one address at ~13 ns per grab -- 0.2 ms for 14 K slices
      const int i = blockIdx.x * 4 + wave + iter * (int)gridDim.x * 4;
      if (i >= qn) break;
      const int it = pk.queue[PULL_QHDR + i];
      if (it < 0) continue;
      tile = it & 0xfff; sl = (it >> 12) & 0xff; ns = (it >> 20) & 0xff;   // sl < ns <= 255
    }
    pull_one_tile<C, NLV, MODE, DRAIN>(g, pk, smem, o_list, o_arr, o_rec, o_df, lane, lt_mask, tile, sl, ns);
  }
  if (DRAIN) {   // the last workgroup out rewinds the queue for the next launch
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(&pk.queue[2], 1) == (int)gridDim.x - 1) { pk.queue[0] = 0; pk.queue[2] = 0; }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path: one WORKGROUP (8 wavefronts) per 2x2x2 block of tiles.
//
// The per-tile kernel above is bound by the serial chain of one tile: ten dependent memory round trips of the sweep
// (row table, then nine tile rows one after the other), and 27 x 64 candidates tested against every level's box
// to find the ~430 (point, level) pairs that matter.  Eight neighbouring tiles share most of their neighbourhoods:
// the union is 4x4x4 tiles (8 tile-loads per tile instead of 27), so the workgroup
//   (1) sweeps the union ONCE, cooperatively: 16 rows of tiles, two per wavefront, all coordinate loads of a wave
//       in flight together (one round trip); a candidate outside the block's widest catchment box (3/4 of them)
//       is dropped after six compares, the survivors (~1000) are compacted into an LDS table {xn, index};
//   (2) routes the survivors: densely re-read from the table (125 per wavefront), for every level the exact cell,
//       from it the one-to-eight tiles of the block whose bricks it touches (integer compares against the block's
//       first vertex: no box test per tile, no division), appended to that (tile, level)'s list of 16-bit table slots;
//   (3) lets wavefront w bin / pull / store tile w from its three lists (pull_level, reading the coordinates from
//       the LDS table instead of global memory).
// Taken when 2 | T and T divides every pulled level's size (bricks are (size/T)^3; else the per-tile kernel).  A block
// whose union holds more than BLK_MAX_UNION candidates, or whose survivor table or a list overflows (a crowded
// batch), runs the per-tile routine for its eight tiles -- heavy-tile slicing included -- in the same launch.
constexpr int BLK_WAVES = 8;
constexpr int BLK_CAND = 1152;        // survivor table (uniform cfg-2 batch: ~1000 per block, sigma ~31)
constexpr int BLK_POOL = 704;         // 16-bit list slots per tile, split over the levels by the launcher in
                                      // proportion to their catchment volumes (uniform cfg-2: 91 / 125 / 218 used)
constexpr int BLK_UN = 8;             // 64-candidate steps in flight per wavefront

template <int C, int NLV, int MODE>
__global__ __launch_bounds__(512, 4) void grad_pull_block_kernel(GridK g, PullK pk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int REC = pull_rec(MODE);
  // LDS (words): survivor table {xn.x, xn.y, xn.z, index} | lists | counters | per-wave staging.  The per-tile
  // fallback reuses table + lists as its per-wave index lists (NLV * PULL_LIST ints per wave).
  constexpr int O_LISTS = BLK_CAND * 4;
  constexpr int O_CNT = O_LISTS + BLK_WAVES * BLK_POOL / 2;
  constexpr int O_STAGE = O_CNT + 32;
  constexpr int CAP = pull_cap(C, MODE);
  constexpr int STAGE = PULL_ARRW + CAP * REC + CAP * C;
  static_assert(BLK_WAVES * NLV * PULL_LIST <= O_CNT, "fallback lists must fit the table + list area");
  float4* cand = reinterpret_cast<float4*>(smem);        // .w = the point's sorted index (bits)
  uint16_t* lists = reinterpret_cast<uint16_t*>(reinterpret_cast<int*>(smem) + O_LISTS);
  int* cnt = reinterpret_cast<int*>(smem) + O_CNT;       // [tile * NLV + level] list lengths, [24] survivors, [25] overflow
  const int o_arr = O_STAGE + wave * STAGE, o_rec = o_arr + PULL_ARRW, o_df = o_rec + CAP * REC;
  const int T = pk.T, nb = T / 2;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

  for (int blk = blockIdx.x; blk < nb * nb * nb; blk += gridDim.x) {
    const int bx = blk % nb, by = (blk / nb) % nb, bz = blk / (nb * nb);
    if (threadIdx.x < 32) cnt[threadIdx.x] = 0;
    // ---- (1) cooperative sweep of the 4x4x4 tile union, (2) routing -------------------------------------------
    const int tx0 = max(2 * bx - 1, 0), tx1 = min(2 * bx + 2, T - 1);
    const int ty0 = max(2 * by - 1, 0), ny = min(2 * by + 2, T - 1) - ty0 + 1;
    const int tz0 = max(2 * bz - 1, 0), nz = min(2 * bz + 2, T - 1) - tz0 + 1;
    const int nrows = ny * nz;
    // the widest catchment box of the block over the pulled levels (axis-wise), cf. pull_one_tile
    float ulo[3] = {3e30f, 3e30f, 3e30f}, uhi[3] = {-3e30f, -3e30f, -3e30f};
    const int b3[3] = {bx, by, bz};
#pragma unroll
    for (int d = 0; d < NLV; ++d) {
      if (d >= pk.nl || ((g.ignore_mask >> pk.lev[d]) & 1u)) continue;
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int B = pk.bdiv[d][a], v0 = 2 * b3[a] * B;
        ulo[a] = fminf(ulo[a], (2.0f * v0 - 1.0f) * pk.inv_size[d][a] - 1.0f - 8e-6f);
        uhi[a] = fmaxf(uhi[a], (2.0f * (v0 + 2 * B) + 1.0f) * pk.inv_size[d][a] - 1.0f + 8e-6f);
      }
    }
    int rs[2], re[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = wave + r * BLK_WAVES;
      rs[r] = re[r] = 0;
      if (row < nrows) {
        const int ry = ty0 + row % ny, rz = tz0 + row / ny;
        rs[r] = pk.tile_off[(rz * T + ry) * T + tx0];
        re[r] = pk.tile_off[(rz * T + ry) * T + tx1 + 1];
      }
    }
    __syncthreads();       // counters are zero
    const bool forced = (pk.debug & 8) != 0;
#pragma unroll 1
    for (int r = 0; r < 2 && !forced; ++r) {
#pragma unroll 1
      for (int p0 = rs[r]; p0 < re[r]; p0 += 64 * BLK_UN) {
        if (cnt[25]) break;            // a crowded block: the table or a list is full already
        const int nstep = __builtin_amdgcn_readfirstlane(min(BLK_UN, (re[r] - p0 + 63) / 64));
        float4 c4[BLK_UN];
#pragma unroll
        for (int u = 0; u < BLK_UN; ++u) {
          const int p = p0 + u * 64 + lane;
          c4[u] = make_float4(2e30f, 2e30f, 2e30f, 0.f);
          if (u < nstep && p < re[r]) c4[u] = pk.xn[p];
        }
#pragma unroll
        for (int u = 0; u < BLK_UN; ++u) {
          if (u >= nstep) continue;
          const bool in = c4[u].x >= ulo[0] && c4[u].x < uhi[0] && c4[u].y >= ulo[1] && c4[u].y < uhi[1] &&
                          c4[u].z >= ulo[2] && c4[u].z < uhi[2];
          const unsigned long long m = __ballot(in);
          if (m == 0ull) continue;
          int base = 0;
          if (lane == 0) base = atomicAdd(&cnt[24], (int)__popcll(m));
          base = __builtin_amdgcn_readfirstlane(base);
          const int slot = base + (int)__popcll(m & lt_mask);
          if (in) {
            if (slot < BLK_CAND) cand[slot] = make_float4(c4[u].x, c4[u].y, c4[u].z, __int_as_float(p0 + u * 64 + lane));
            else cnt[25] = 1;
          }
        }
      }
    }
    __syncthreads();
    // ---- (2) route the survivors, densely re-read from the table: for every level the exact cell, from it the tiles
    // of the block whose bricks it touches ----------------------------------------------------------------------
    if (!forced && cnt[25] == 0 && !(pk.debug & 32)) {
      const int ncand = cnt[24];
#pragma unroll 1
      for (int s0 = wave * 64; s0 < ncand; s0 += 64 * BLK_WAVES) {
        const int slot = s0 + lane;
        const bool live = slot < ncand;
        const float4 c4 = live ? cand[slot] : make_float4(2e30f, 2e30f, 2e30f, 0.f);
#pragma unroll
        for (int d = 0; d < NLV; ++d) {
          if (d >= pk.nl || ((g.ignore_mask >> pk.lev[d]) & 1u)) continue;
          const LevelK& lv = g.lv[pk.lev[d]];
          int i0[3]; float fr;
          cell_of(c4.x, lv.X, i0[0], fr); cell_of(c4.y, lv.Y, i0[1], fr); cell_of(c4.z, lv.Z, i0[2], fr);
          // per axis: which of the block's two tiles own vertex i0 / i0 + 1 (brick of tile l: [v0 + l B, v0 + (l+1) B))
          unsigned m3[3];
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const int B = pk.bdiv[d][a], rel = i0[a] - 2 * b3[a] * B;      // vertex i0 relative to the block's first
            unsigned mm = 0;
            if (rel >= 0 && rel < 2 * B) mm |= (rel >= B) ? 2u : 1u;
            if (rel + 1 >= 0 && rel + 1 < 2 * B) mm |= (rel + 1 >= B) ? 2u : 1u;
            m3[a] = mm;
          }
          unsigned m8 = ((m3[2] & 1u ? 0x0Fu : 0u) | (m3[2] & 2u ? 0xF0u : 0u)) &
                        ((m3[1] & 1u ? 0x33u : 0u) | (m3[1] & 2u ? 0xCCu : 0u)) &
                        ((m3[0] & 1u ? 0x55u : 0u) | (m3[0] & 2u ? 0xAAu : 0u));
          if (!live) m8 = 0;
          while (m8) {
            const int t = __ffs((int)m8) - 1;
            m8 &= m8 - 1;
            const int pos = atomicAdd(&cnt[t * NLV + d], 1);
            if (pos < pk.blk_cap[d]) lists[t * BLK_POOL + pk.blk_off[d] + pos] = (uint16_t)slot;
            else cnt[25] = 1;
          }
        }
      }
    }
    __syncthreads();
    const bool fallback = forced || cnt[25] != 0;
    const int ta = 2 * bx + (wave & 1), tb = 2 * by + ((wave >> 1) & 1), tc = 2 * bz + (wave >> 2);
    if (fallback) {
      // crowded block: the per-tile routine (own sweep, slicing) for this wavefront's tile
      pull_one_tile<C, NLV, MODE, false>(g, pk, smem, wave * (NLV * PULL_LIST), o_arr, o_rec, o_df, lane, lt_mask,
                                         (tc * T + tb) * T + ta, 0, 1);
    } else if (!(pk.debug & 16)) {
      // ---- (3) wavefront w: tile w of the block -----------------------------------------------------------------
      // finest level first: its brick is most of the bytes this tile writes (64 of 73 MB at cfg-2), and stores need
      // no wait -- they drain to HBM while the coarser levels are binned and pulled
#pragma unroll 1
      for (int d = pk.nl - 1; d >= 0; --d) {
        const LevelK& lv = g.lv[pk.lev[d]];
        const Brick b = make_brick(lv, ta, tb, tc, T, pk.bdiv[d]);
        if (b.nverts == 0) continue;
        const int n = __builtin_amdgcn_readfirstlane(cnt[wave * NLV + d]);
        pull_level<C, MODE, true>(g, pk, lv, b, smem, 0, n, o_arr, o_rec, o_df, lane, pk.overwrite ? 0 : 1,
                                  lists + wave * BLK_POOL + pk.blk_off[d], cand);
      }
    }
    __syncthreads();   // the next block reuses the table, the lists and the counters
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Coarse levels under a crowd, pushed through the fp32 matrix cores.  A coarse level (ScanNet's 40 x 20 x 40 over 16
// tiles) of a batch that piles hundreds of samples on every occupied tile: all samples of a tile touch the same
// <= 5 x 5 x 5 vertices (its REGION), so the tile's contribution is a small dense product
//     G[(vx,vy)][(vz,c)] = sum_p  (wx_p[vx] wy_p[vy]) * (wz_p[vz] d_p[c])
// with the per-axis trilinear weights w (two non-zeros among five per axis) and the sample's d-feat row d.  That is
// A^T B with A = (samples x 25) and B = (samples x 5 C): v_mfma_f32_32x32x2_f32 adds two samples per instruction into
// a 32 x 32 accumulator tile (16 registers), exact fp32 FMA chains.  Nothing is binned, swept or reduced across lanes:
// a wavefront takes a run of PUSH_RUN tile-sorted samples, stages 64 at a time in LDS (lane = sample: three cell_of, 15
// weights, the d-feat row), issues 32 MFMAs over them (lane = (sample parity, matrix row / column): two LDS reads and a
// product per operand), and at the end of a tile's samples adds the non-zero accumulators to the gradient with
// atomics (the regions of neighbouring tiles overlap).  The level's gradient is zero-filled by the caller.
// Measured at the ScanNet shape (540 000 samples): the dense wave walk this replaces cost ~85 us of the backward pass
// and a workgroup-per-tile owner-computes kernel with register accumulators 187 us (27 x candidate sweep).
constexpr int PUSH_R = 5;          // region vertices per axis
constexpr int PUSH_WAVES = 4;

struct PushRegion { int r0[3]; };

// first region vertex of tile coordinate t along an axis of `size` vertices: the smallest base corner of a sample of
// the tile, pos >= t size / T - 1/2 (the -1: a tile boundary that is an integer position keeps the vertex below it, a
// sample may sit an ulp on the other side of it than its tile says)
__host__ __device__ inline int push_r0(int t, int size, int T) {
  const int a = 2 * t * size - T - 1, b = 2 * T;
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}
// last region vertex: the largest base corner (pos < (t + 1) size / T - 1/2, the bound included) plus one
__host__ __device__ inline int push_r1(int t, int size, int T) {
  const int a = 2 * (t + 1) * size - T, b = 2 * T;
  int q = a / b;
  return ((a % b != 0 && a < 0) ? q - 1 : q) + 1;
}

template <int C>
__global__ __launch_bounds__(64 * PUSH_WAVES) void grad_push_mfma_kernel(GridK g, PullK pk, int64_t n, int run) {
  constexpr int NB = C / 4;                                  // 32-column blocks of B: (5 vz) x (4 channels) each
  __shared__ float s_w[PUSH_WAVES][64][16];                  // per sample: wx[5], wy[5], wz[5], 0
  __shared__ float s_d[PUSH_WAVES][64][C];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lvl = pk.lev[0];
  const LevelK& lv = g.lv[lvl];
  if ((g.ignore_mask >> lvl) & 1u) return;
  const int T = pk.T, ntiles = T * T * T;
  const int64_t w_id = (int64_t)blockIdx.x * PUSH_WAVES + wave;
  const int p_begin = (int)min(n, w_id * run), p_end = (int)min(n, (w_id + 1) * run);
  if (p_begin >= p_end) return;
  const int size[3] = {lv.X, lv.Y, lv.Z};

  // the tile of the run's first sample: tile_off is non-decreasing, two lane-parallel probes (stride 64, then 1)
  int tile;
  {
    const int nblk = (ntiles + 63) >> 6;
    int cnt = 0;
    for (int b0 = 0; b0 < nblk; b0 += 64) {                  // T = 16: one round
      const int i = b0 + lane;
      const bool le = i < nblk && pk.tile_off[i << 6] <= p_begin;
      cnt += (int)__popcll(__ballot(le));
    }
    const int blk = max(cnt - 1, 0);
    const int i = (blk << 6) + lane;
    const bool le = i < ntiles && pk.tile_off[i] <= p_begin;
    tile = (blk << 6) + max((int)__popcll(__ballot(le)) - 1, 0);
  }
  // lane constants of the operand reads
  const int half = lane >> 5, ij = lane & 31;
  const int a_x = ij < 25 ? ij % 5 : 15, a_y = ij < 25 ? 5 + ij / 5 : 15;
  const int b_z = ij < 20 ? 10 + (ij >> 2) : 15, b_c = ij & 3;

  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nb][r] = 0.0f;

  int p = p_begin;
  while (p < p_end) {
    // skip to the tile that holds sample p (empty tiles in between), 64 offsets per probe
    int t_end = __builtin_amdgcn_readfirstlane(pk.tile_off[tile + 1]);
    while (t_end <= p) {
      const int i = tile + 1 + lane;
      const bool le = i < ntiles && pk.tile_off[i + 1] <= p;        // tile i ends at or before p: not it
      const unsigned long long m = __ballot(!le);
      tile += 1 + (m ? __builtin_ctzll(m) : 64);
      tile = min(tile, ntiles - 1);
      t_end = __builtin_amdgcn_readfirstlane(pk.tile_off[tile + 1]);
      if (tile == ntiles - 1) break;
    }
    const int seg_end = min(p_end, t_end);
    const int tabc[3] = {tile % T, (tile / T) % T, tile / (T * T)};
    int r0[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) r0[a] = push_r0(tabc[a], size[a], T);

    for (; p < seg_end; p += 64) {
      const int cnt = min(64, seg_end - p);
      // ---- stage: lane = sample ----------------------------------------------------------------------------------
      {
        const bool live = lane < cnt;
        float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) c4 = pk.xn[p + lane];
        const float xs[3] = {c4.x, c4.y, c4.z};
        float w[16];
        bool ok = live;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          int i0; float fr;
          cell_of(xs[a], size[a], i0, fr);
          const int rel = i0 - r0[a];                       // base corner -> region slot rel, next corner -> rel + 1
          ok = ok && rel >= -1 && rel < PUSH_R;
#pragma unroll
          for (int o = 0; o < PUSH_R; ++o) w[a * 5 + o] = (o == rel) ? 1.0f - fr : ((o == rel + 1) ? fr : 0.0f);
        }
        w[15] = 0.0f;
        float dv[C];
#pragma unroll
        for (int c = 0; c < C; ++c) dv[c] = 0.0f;
        if (ok) {
          const int rowi = pk.perm ? pk.perm[p + lane] : p + lane;
          const float* src = pk.dfeat + (int64_t)rowi * pk.ld + lv.foff;
#pragma unroll
          for (int c = 0; c < C; c += 4) {
            const float4 t4 = *reinterpret_cast<const float4*>(src + c);
            dv[c] = t4.x; dv[c + 1] = t4.y; dv[c + 2] = t4.z; dv[c + 3] = t4.w;
          }
        } else {
          // a sample that reaches no region vertex (outside the bound, NaN): all-zero operands, nothing of it can
          // reach an accumulator (0 * NaN would)
#pragma unroll
          for (int o = 0; o < 15; ++o) w[o] = 0.0f;
        }
#pragma unroll
        for (int o = 0; o < 16; o += 4)
          *reinterpret_cast<float4*>(&s_w[wave][lane][o]) = make_float4(w[o], w[o + 1], w[o + 2], w[o + 3]);
#pragma unroll
        for (int c = 0; c < C; c += 4)
          *reinterpret_cast<float4*>(&s_d[wave][lane][c]) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
      }
      wave_sync_lds();
      // ---- 32 x 32 x 2 products: lanes 0-31 feed sample 2 s, lanes 32-63 sample 2 s + 1 --------------------------
      // (in groups of four steps: rows past `cnt` are staged as zeros, so running over them adds nothing)
      const int nstep = (pk.debug & 128) ? 0 : (cnt + 1) >> 1;
      for (int st0 = 0; st0 < nstep; st0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = 2 * (st0 + u) + half;
          const float av = s_w[wave][k][a_x] * s_w[wave][k][a_y];
          const float wz = s_w[wave][k][b_z];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wz * s_d[wave][k][nb * 4 + b_c], acc[nb], 0, 0, 0);
        }
      }
      wave_sync_lds();                                       // the next 64 overwrite the staging rows
    }
    p = seg_end;
    // ---- the tile's samples of this run are in: add the region to the gradient ---------------------------------------
    // accumulator register r of lane l: row (vx,vy) = 8 (r / 4) + 4 (l / 32) + r % 4, column (vz,c) = l % 32
    const int vz = r0[2] + (ij >> 2);
    const bool col_ok = ij < 20 && vz >= 0 && vz < lv.Z;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float* colp = lv.grad + (int64_t)vz * lv.sZ + (int64_t)(nb * 4 + b_c) * lv.sC;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * half + (r & 3);
        const int vx = r0[0] + row % 5, vy = r0[1] + row / 5;
        const float v = acc[nb][r];
        if (col_ok && row < 25 && v != 0.0f && vx >= 0 && vx < lv.X && vy >= 0 && vy < lv.Y && !(pk.debug & 64)) {
          float* dst = colp + (int64_t)vy * lv.sY + (int64_t)vx * lv.sX;
          atomic_add_f32(dst, v);
          touch_chunk(lv, (int64_t)(dst - lv.grad));
        }
        acc[nb][r] = 0.0f;
      }
    }
  }
}

// Levels the pull kernels can own: default sampling convention, a gradient requested, and a brick of at most
// PULL_BMAX^3 vertices per tile (swept together, at most PULL_MAXL of them).  Levels with larger bricks (ScanNet's
// 200 x 100 x 200 over 16 tiles: 13 x 7 x 13) are scattered with float atomics from the backward kernel: owner-computes
// was built for them four ways over two rounds (tools/experiments/README.md) and every version cost at least what the
// 4.3 M atomic requests cost.
uint32_t plan_grad_pull(const GridK& g, int tiles) {
  if (g.flags & (MISO_F_ALIGN_CORNERS | MISO_F_PAD_BORDER)) return 0;
  int T3[3];
  if (!tiles_xyz(tiles, T3)) return 0;
  // a per-axis or finer-than-16 binning is served by the matrix-core pull only (grad_pull_mc.hip): its levels must be
  // at least as fine as the binning and share the channel count it is compiled for
  const bool mc_only = !tiles_cubic16(tiles);
  uint32_t mask = 0;
  int cnt = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if (!lv.grad) continue;
    const int size[3] = {lv.X, lv.Y, lv.Z};
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
      if ((size[a] + T3[a] - 1) / T3[a] > PULL_BMAX) ok = false;
      if ((int64_t)size[a] * T3[a] >= (1 << 28)) ok = false;
      if (mc_only && 3 * size[a] < 2 * T3[a]) ok = false;
    }
    if (mc_only && ((lv.C != 4 && lv.C != 8) || lv.sC != 1 || lv.C != g.lv[0].C)) ok = false;
    if (!ok) continue;
    if (cnt++ >= PULL_MAXL) continue;
    mask |= 1u << l;
  }
  return mask;
}

// levels (subset of `pull`) for the matrix-core push: a crowd of at least MISO_DENSE_MIN (100) samples per tile on
// average, 4 or 8 channels, and every tile's region within PUSH_R vertices per axis
uint32_t plan_push(const GridK& g, int tiles, int64_t n, uint32_t pull) {
  if (!tiles_cubic16(tiles)) return 0;      // (a finer binning: the matrix-core pull takes crowded levels as they come)
  const int T = tiles;
  static const int dense_min = [] { const char* e = getenv("MISO_DENSE_MIN"); return e ? atoi(e) : 100; }();
  // MISO_F_CROWDED: the caller knows the batch piles up on a few tiles (ray samples: a 54 000-sample batch of the
  // synthetic RGB-D demo put 112 us of pull + drain launches on the 40 x 20 x 40 level, the push 46 us), where the
  // average density says nothing; a uniform batch below the threshold is better off pulled (cfg-2: 107 -> 180 us)
  const bool crowded = (g.flags & MISO_F_CROWDED) != 0 && n >= 4096;
  if (dense_min <= 0 || (!crowded && n < (int64_t)dense_min * T * T * T) || n >= (1ll << 31)) return 0;
  uint32_t push = 0;
  for (int l = 0; l < g.n_levels && l < 16; ++l) {
    const LevelK& lv = g.lv[l];
    if (!((pull >> l) & 1u) || (lv.C != 4 && lv.C != 8)) continue;
    const int size[3] = {lv.X, lv.Y, lv.Z};
    bool ok = true;
    for (int a = 0; a < 3 && ok; ++a)
      for (int t = 0; t < T && ok; ++t) ok = push_r1(t, size[a], T) - push_r0(t, size[a], T) + 1 <= PUSH_R;
    if (ok) push |= 1u << l;
  }
  return push;
}

bool mc_pull_ok(const GridK& g, int C, const int T[3], uint32_t level_mask, int64_t n, int64_t ld);
hipError_t launch_grad_pull_mc(const GridK& g, int C, const int T[3], const int* tile_off, const float* xn,
                               const float* dfeat, int64_t ld, const int* perm, uint32_t level_mask, int overwrite,
                               int64_t n, hipStream_t s);

static hipError_t launch_push(const GridK& g, int C, int T, const int* tile_off, const float* xn, const float* dfeat,
                              int64_t ld, const int* perm, int level, int64_t n, hipStream_t s) {
  // samples per wavefront: 512 (fewest region flushes per sample) once that still makes >= 2048 wavefronts -- two per
  // SIMD, one staging while the other multiplies (540 000 samples: 512 / 384 / 256 / 128 per wavefront give a trainer
  // step of 360 / 356 / 353 / 366 us); a small batch is cut finer so that the chip is not left to a few dozen
  // wavefronts (54 000 samples: 105 of them, 46 us; 844 of 64 samples, 26 us)
  const int run = (int)min((int64_t)512, max((int64_t)64, ((n / 2048 + 63) / 64) * 64));
  PullK pk;
  memset(&pk, 0, sizeof(pk));
  pk.T = T; pk.tile_off = tile_off; pk.xn = reinterpret_cast<const float4*>(xn); pk.dfeat = dfeat;
  pk.ld = ld; pk.perm = perm;
  pk.lev[0] = level; pk.nl = 1;
  static const int dbg = [] { const char* e = getenv("MISO_DEBUG_PULL"); return e ? atoi(e) : 0; }();   // dev ablation
  pk.debug = dbg;
  const int64_t waves = (n + run - 1) / run;
  const unsigned blocks = (unsigned)((waves + PUSH_WAVES - 1) / PUSH_WAVES);
  if (!blocks) return hipSuccess;
  if (C == 8) grad_push_mfma_kernel<8><<<blocks, 64 * PUSH_WAVES, 0, s>>>(g, pk, n, run);
  else grad_push_mfma_kernel<4><<<blocks, 64 * PUSH_WAVES, 0, s>>>(g, pk, n, run);
  return hipGetLastError();
}

hipError_t launch_zero_fill(float*, int64_t, hipStream_t);
hipError_t launch_grad_pull(const GridK& g, int C, int tiles, const int* tile_off, const float* xn,
                            const float* dfeat, int64_t ld, const int* perm, uint32_t level_mask,
                            int overwrite, const float* ggx, int32_t* queue, int64_t queue_ints, hipStream_t s,
                            uint32_t push_mask, int64_t n) {
  int T3[3];
  if (!tiles_xyz(tiles, T3)) return hipErrorInvalidValue;
  const int T = T3[0];       // (the vector kernels below: cubic binning only)
  // push_mask: levels of plan_push the caller has zero-filled (overwrite) -- added to with atomics
  push_mask &= level_mask;
  if (push_mask && !ggx) {
    for (int l = 0; l < g.n_levels; ++l)
      if ((push_mask >> l) & 1u) {
        hipError_t e = launch_push(g, C, T, tile_off, xn, dfeat, ld, perm, l, n, s);
        if (e != hipSuccess) return e;
      }
    level_mask &= ~push_mask;
  }
  if (!level_mask) return hipSuccess;
  {
    // first-order gradients on grids at least as fine as the binning: the matrix-core pull (grad_pull_mc.hip)
    if (!ggx && mc_pull_ok(g, C, T3, level_mask, n, ld))
      return launch_grad_pull_mc(g, C, T3, tile_off, xn, dfeat, ld, perm, level_mask, overwrite, n, s);
    if (!tiles_cubic16(tiles)) {
      // plan_grad_pull admits a per-axis binning for the matrix-core kernel only.  What it cannot serve: an empty batch
      // (the gradient of nothing is zero -- written here when the caller asked for overwrite, as the vector
      // kernels of a cubic binning do) and d-feat rows past its 32-bit offsets (pull_serviceable: refused by the entry
      // points before anything is launched)
      if (n > 0) return hipErrorInvalidValue;
      for (int l = 0; overwrite && l < g.n_levels; ++l)
        if ((level_mask >> l) & 1u) {
          const LevelK& lv = g.lv[l];
          const size_t span = (size_t)(lv.C - 1) * lv.sC + (size_t)(lv.X - 1) * lv.sX + (size_t)(lv.Y - 1) * lv.sY +
                              (size_t)(lv.Z - 1) * lv.sZ + 1;
          hipError_t e = launch_zero_fill(lv.grad, (int64_t)span, s);
          if (e != hipSuccess) return e;
        }
      return hipSuccess;
    }
  }
  PullK pk;
  memset(&pk, 0, sizeof(pk));
  pk.T = T; pk.tile_off = tile_off; pk.xn = reinterpret_cast<const float4*>(xn); pk.dfeat = dfeat;
  pk.ld = ld; pk.perm = perm; pk.ggx = ggx;
  for (int l = 0; l < g.n_levels; ++l)
    if ((level_mask >> l) & 1u) {
      const int size[3] = {g.lv[l].X, g.lv[l].Y, g.lv[l].Z};
      for (int a = 0; a < 3; ++a) {
        pk.bdiv[pk.nl][a] = (size[a] % T == 0) ? size[a] / T : 0;
        pk.inv_size[pk.nl][a] = 1.0f / (float)size[a];
      }
      pk.lev[pk.nl++] = l;
    }
  pk.overwrite = overwrite;
  if (queue && queue_ints > PULL_QHDR && T <= 16 && !getenv("MISO_PULL_NO_SPLIT")) {
    pk.queue = queue;
    pk.qcap = (int)((queue_ints - PULL_QHDR) < (1 << 30) ? (queue_ints - PULL_QHDR) : (1 << 30));
  }
  pk.work0 = PULL_WORK0;
  if (const char* d = getenv("MISO_DEBUG_PULL")) pk.debug = atoi(d);
  const int cap = pull_cap(C, ggx ? 1 : 0);
  const int per_wave = pk.nl * PULL_LIST + PULL_ARRW + cap * pull_rec(ggx ? 1 : 0) + cap * C;
  size_t lds = (size_t)per_wave * 4 * sizeof(float);
  const int ntiles = T * T * T;
  unsigned blocks = (unsigned)((ntiles + 3) / 4);
  if (blocks > 2048u) blocks = 2048u;
  void (*k)(GridK, PullK) = nullptr;
  void (*kd)(GridK, PullK) = nullptr;
  void (*kb)(GridK, PullK) = nullptr;
#define PICK(c, n)                                                                            \
  if (C == c && pk.nl == n) {                                                                 \
    k = ggx ? grad_pull_kernel<c, n, 1, false> : grad_pull_kernel<c, n, 0, false>;            \
    kd = ggx ? grad_pull_kernel<c, n, 1, true> : grad_pull_kernel<c, n, 0, true>;             \
  }
  PICK(8, 1) PICK(8, 2) PICK(8, 3) PICK(8, 4) PICK(4, 1) PICK(4, 2) PICK(4, 3) PICK(4, 4)
#undef PICK
#define PICKB(c, n) \
  if (C == c && pk.nl == n) kb = ggx ? grad_pull_block_kernel<c, n, 1> : grad_pull_block_kernel<c, n, 0>;
  PICKB(8, 1) PICKB(8, 2) PICKB(8, 3) PICKB(4, 1) PICKB(4, 2) PICKB(4, 3)
#undef PICKB
  if (!k) return hipErrorInvalidValue;
  // block kernel (one workgroup per 2x2x2 tiles): bricks must be (size / T)^3 for every pulled level
  bool block_ok = (T % 2 == 0) && kb != nullptr;
  for (int d = 0; d < pk.nl; ++d)
    for (int a = 0; a < 3; ++a) block_ok = block_ok && pk.bdiv[d][a] > 0;
  if (block_ok) {
    const int rec = pull_rec(ggx ? 1 : 0);
    // a tile's list pool, split in proportion to the catchment volume ((B+1)/B)^3 of each level's brick
    double w[PULL_MAXL], wsum = 0.0;
    for (int d = 0; d < pk.nl; ++d) {
      w[d] = 1.0;
      for (int a = 0; a < 3; ++a) w[d] *= (double)(pk.bdiv[d][a] + 1) / (double)pk.bdiv[d][a];
      wsum += w[d];
    }
    int off = 0;
    for (int d = 0; d < pk.nl; ++d) {
      pk.blk_off[d] = off;
      pk.blk_cap[d] = (d + 1 == pk.nl) ? BLK_POOL - off : ((int)(BLK_POOL * w[d] / wsum) & ~1);
      off += pk.blk_cap[d];
    }
    const size_t words = (size_t)BLK_CAND * 4 + (size_t)BLK_WAVES * BLK_POOL / 2 + 32 +
                         (size_t)BLK_WAVES * (PULL_ARRW + cap * rec + cap * C);
    const size_t blds = words * sizeof(float);
    hipError_t e = allow_dynamic_lds((const void*)kb, blds);
    if (e != hipSuccess) return e;
    const int nb = T / 2;
    kb<<<(unsigned)(nb * nb * nb), 512, blds, s>>>(g, pk);
  } else {
    {
      hipError_t e = allow_dynamic_lds((const void*)k, lds);
      if (e != hipSuccess) return e;
    }
    k<<<blocks, 256, lds, s>>>(g, pk);
  }
  if (pk.queue) {
    pk.drain = 1;
    const unsigned dblocks = 1024;
    {
      hipError_t e = allow_dynamic_lds((const void*)kd, lds);
      if (e != hipSuccess) return e;
    }
    kd<<<dblocks, 256, lds, s>>>(g, pk);
  }
  return hipGetLastError();
}

// ints a slice queue for a batch of n points needs: every point is swept by at most 64 tiles, a slice is cut
// per PULL_WORK0 swept candidates
int64_t pull_queue_ints(int64_t n) { return PULL_QHDR + 64 + (n * 64) / PULL_WORK0; }

}  // namespace miso
