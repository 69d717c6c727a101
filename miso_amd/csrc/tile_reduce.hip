// Per-tile pre-reduction of the coarse levels' gradient (second half of the binned backward).
//
// Why: the L2 executes fp32 atomics at ~21 G requests/s however local they are
// (tools/ubench/atomics.hip), and a point batch issues N * 4 rows * 1.5 requests per level.
// On the COARSE levels many points share vertices, so summing a spatial tile's contributions
// on chip first cuts the requests of a level from ~1.6 M to (touched 2C-float runs) ~0.1-0.4 M.
// ds_add_f32 is no help (it serialises: ~0.4 lane-ops/clk/CU, tools/ubench/lds_atomics.hip),
// so the accumulation is a plain LDS read-add-write made conflict-free by construction:
// one wavefront owns one tile and its private LDS copy of the tile's vertex region, and one
// instruction handles ONE point = 8 corners x C channels = 8C distinct addresses.
//
// Input: points in tile-sorted order (sort.hip) and the rows of d(feats) written by
// sdf_bwd_kernel (dfeat_out).  Corners outside the wave's region (numerical edge cases,
// points clamped into border tiles) fall back to direct atomics, so the result never depends
// on the region estimate.  No reference counterpart (ATen scatters one atomic per corner per
// channel, third_party/cuda_gridsample_grad2/gridsample_cuda.cu:466-481).
#include <string.h>

#include "common.hpp"

namespace miso {

struct TileRed {
  int T;
  const int* tile_off;   // T^3 + 1
  uint32_t level_mask;   // levels handled here
  int nd;                // how many
  int lev[MISO_MAX_LEVELS];   // their indices
  int W[MISO_MAX_LEVELS][3];  // region extent per handled level (x even-padded, y, z)
  int acc_off[MISO_MAX_LEVELS];
  int acc_total;         // floats per wave
};

// first vertex a tile's points can touch: floor(ix) for ix = u*size - 0.5 at the tile's low edge
__device__ __forceinline__ int region_lo(int a, int size, int T) {
  return (int)floorf((float)a * (float)size / (float)T - 0.5f);
}

template <int C>
__global__ __launch_bounds__(256) void tile_reduce_kernel(GridK g, TileRed tr, const float* __restrict__ xs,
                                                         const float* __restrict__ dfeat, int F) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int REC = 4;
  const int nd = tr.nd;
  // per wave: accumulators | cell records [64][nd][4] | d-feat staging [64][nd*C]
  const int per_wave = tr.acc_total + 64 * nd * REC + 64 * nd * C;
  const int acc_base = wave * per_wave;
  const int rec_base = acc_base + tr.acc_total;
  const int stg_base = rec_base + 64 * nd * REC;
  const int ntiles = tr.T * tr.T * tr.T;
  // lane role while accumulating: corner k, channel ch (C == 4: lanes 32..63 idle)
  const int k = lane / C, ch = lane % C;
  const bool role = k < 8;
  const int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;

  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int t0 = tr.tile_off[tile], t1 = tr.tile_off[tile + 1];
    if (t1 <= t0) continue;
    const int ta = tile % tr.T, tb = (tile / tr.T) % tr.T, tc = tile / (tr.T * tr.T);
    for (int i = lane * 4; i < tr.acc_total; i += 256)
      *reinterpret_cast<float4*>(smem + acc_base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = t0; base < t1; base += 64) {
      const int cnt = min(64, t1 - base);
      // ---- stage this batch: lane = point ------------------------------------------------
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane < cnt) {
        const int64_t p = base + lane;
        const float px = xs[p * 3 + 0], py = xs[p * 3 + 1], pz = xs[p * 3 + 2];
        for (int d = 0; d < nd; ++d) {
          const LevelK& lv = g.lv[tr.lev[d]];
          Axis ax = axis_coord(px, g.bmin[0], g.bmax[0], lv.X, g.flags);
          Axis ay = axis_coord(py, g.bmin[1], g.bmax[1], lv.Y, g.flags);
          Axis az = axis_coord(pz, g.bmin[2], g.bmax[2], lv.Z, g.flags);
          Cell c = make_cell(ax, ay, az, lv);
          *reinterpret_cast<int4*>(smem + rec_base + (lane * nd + d) * REC) =
              make_int4((c.i0 + 2) | ((c.j0 + 2) << 10) | ((c.k0 + 2) << 20), __float_as_int(c.wx[1]),
                        __float_as_int(c.wy[1]), __float_as_int(c.wz[1]));
#pragma unroll
          for (int f = 0; f < C; f += 4)
            *reinterpret_cast<float4*>(smem + stg_base + (lane * nd + d) * C + f) =
                *reinterpret_cast<const float4*>(dfeat + p * F + lv.foff + f);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // ---- accumulate: one point per instruction, lane = (corner, channel) -----------------
      for (int d = 0; d < nd; ++d) {
        const LevelK& lv = g.lv[tr.lev[d]];
        const int lox = region_lo(ta, lv.X, tr.T) & ~1, loy = region_lo(tb, lv.Y, tr.T),
                  loz = region_lo(tc, lv.Z, tr.T);
        const int W0 = tr.W[d][0], W1 = tr.W[d][1], W2 = tr.W[d][2];
        const int acc_l = acc_base + tr.acc_off[d];
#pragma unroll 2
        for (int q = 0; q < cnt; ++q) {
          const int4 r = *reinterpret_cast<const int4*>(smem + rec_base + (q * nd + d) * REC);
          if (!role) continue;
          const int i = (r.x & 1023) - 2 + dx, j = ((r.x >> 10) & 1023) - 2 + dy, kk = ((r.x >> 20) & 1023) - 2 + dz;
          if ((unsigned)i >= (unsigned)lv.X || (unsigned)j >= (unsigned)lv.Y || (unsigned)kk >= (unsigned)lv.Z)
            continue;
          const float w1x = __int_as_float(r.y), w1y = __int_as_float(r.z), w1z = __int_as_float(r.w);
          const float w = ((dx ? w1x : 1.0f - w1x) * (dy ? w1y : 1.0f - w1y)) * (dz ? w1z : 1.0f - w1z);
          const float val = w * smem[stg_base + (q * nd + d) * C + ch];
          const int li = i - lox, lj = j - loy, lk = kk - loz;
          if ((unsigned)li < (unsigned)W0 && (unsigned)lj < (unsigned)W1 && (unsigned)lk < (unsigned)W2) {
            const int a = acc_l + ((lk * W1 + lj) * W0 + li) * C + ch;
            smem[a] = smem[a] + val;
          } else {
            atomic_add_f32(lv.grad + kk * lv.sZ + j * lv.sY + i * lv.sX + ch, val);
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- flush: 2C lanes per even-aligned x pair, zero runs skipped ---------------------------
    {
      constexpr int LPR = 2 * C;
      const int grp = lane / LPR, within = lane % LPR, dxv = within / C, fch = within % C;
      for (int d = 0; d < nd; ++d) {
        const LevelK& lv = g.lv[tr.lev[d]];
        const int lox = region_lo(ta, lv.X, tr.T) & ~1, loy = region_lo(tb, lv.Y, tr.T),
                  loz = region_lo(tc, lv.Z, tr.T);
        const int W0 = tr.W[d][0], W1 = tr.W[d][1], W2 = tr.W[d][2];
        const int acc_l = acc_base + tr.acc_off[d];
        const int npairs = (W0 / 2) * W1 * W2;
        for (int pr = grp; pr < npairs; pr += 64 / LPR) {
          const int ip = pr % (W0 / 2), jy = (pr / (W0 / 2)) % W1, kz = pr / ((W0 / 2) * W1);
          const int li = 2 * ip + dxv;
          const float val = smem[acc_l + ((kz * W1 + jy) * W0 + li) * C + fch];
          const int gi = lox + li, gj = loy + jy, gk = loz + kz;
          if (val != 0.0f && (unsigned)gi < (unsigned)lv.X && (unsigned)gj < (unsigned)lv.Y &&
              (unsigned)gk < (unsigned)lv.Z)
            atomic_add_f32(lv.grad + gk * lv.sZ + gj * lv.sY + gi * lv.sX + fch, val);
        }
      }
    }
  }
}

// Levels whose per-tile vertex region fits the per-wave LDS budget are handled by the
// tile reduction; returns their mask and fills the plan.
uint32_t plan_tile_reduce(const GridK& g, int T, TileRed* tr, int budget_floats) {
  memset(tr, 0, sizeof(*tr));
  tr->T = T;
  int total = 0;
  uint32_t mask = 0;
  for (int l = 0; l < g.n_levels; ++l) {
    const LevelK& lv = g.lv[l];
    if (!lv.grad || ((g.ignore_mask >> l) & 1u)) continue;
    if (lv.X > 1020 || lv.Y > 1020 || lv.Z > 1020) continue;   // 10-bit packed cell index
    int w0 = ((lv.X + T - 1) / T + 2 + 1 + 1) & ~1, w1 = (lv.Y + T - 1) / T + 2, w2 = (lv.Z + T - 1) / T + 2;
    int need = w0 * w1 * w2 * lv.C;
    if (total + need > budget_floats) continue;
    const int d = tr->nd++;
    tr->lev[d] = l;
    tr->W[d][0] = w0; tr->W[d][1] = w1; tr->W[d][2] = w2;
    tr->acc_off[d] = total;
    total += need;
    mask |= 1u << l;
  }
  tr->acc_total = (total + 3) & ~3;
  tr->level_mask = mask;
  return mask;
}

hipError_t launch_tile_reduce(const GridK& g, TileRed tr, const int* tile_off, const float* xs,
                              const float* dfeat, int C, hipStream_t s) {
  if (tr.nd == 0) return hipSuccess;
  tr.tile_off = tile_off;
  const int per_wave = tr.acc_total + 64 * tr.nd * 4 + 64 * tr.nd * C;
  size_t lds = (size_t)per_wave * 4 * sizeof(float);
  const int ntiles = tr.T * tr.T * tr.T;
  unsigned blocks = (unsigned)((ntiles + 3) / 4);
  if (blocks > 1024u) blocks = 1024u;
  void (*k)(GridK, TileRed, const float*, const float*, int) =
      (C == 8) ? tile_reduce_kernel<8> : tile_reduce_kernel<4>;
  if (lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  k<<<blocks, 256, lds, s>>>(g, tr, xs, dfeat, g.F);
  return hipGetLastError();
}

}  // namespace miso
