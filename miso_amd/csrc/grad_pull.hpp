// Shared definitions of the owner-computes grid gradient kernels (grad_pull.hip, grad_pull_mc.hip).
#pragma once
#include "common.hpp"

namespace miso {

constexpr int PULL_BMAX = 8;      // owned vertices per axis per tile
constexpr int PULL_ARRW = 192;    // words of byte counters: (8+1)^3 = 729 cells, 3 words per lane
constexpr int PULL_LIST = 272;    // compacted candidate ids per level
constexpr int PULL_CAP = 112;     // staged records per group (<= 2 per lane); < 256: counters are bytes.  112: the
                                  // block kernel fits two workgroups per CU (2 x 79 KB of LDS)
// MODE 1 records are 6 words ({fx, fy, fz, ex, ey, ez}, read as three 8-byte pairs) instead of 4: fewer of them per
// group so that a wavefront's staging area is the size of MODE 0's and the block kernel still fits two workgroups per
// CU (at 112 records of 8 words it fitted one, and ran at half speed).  96 records for C = 8: the finest level of a
// uniform cfg-2 batch (~91 candidates per tile) still goes in one group.
constexpr int pull_rec(int mode) { return mode ? 6 : 4; }
constexpr int pull_cap(int C, int mode) { return mode ? ((PULL_CAP * (4 + C)) / (6 + C)) & ~1 : PULL_CAP; }
constexpr int PULL_RB = 4;        // rounds of 64 vertices pulled per pass over the staged records
constexpr int PULL_MAXL = 4;      // levels swept together (the fused kernels cover <= 4 levels)
// Heavy tiles.  One wavefront drains one tile serially (~50 ns per swept candidate), so a batch that piles its
// points into a few tiles -- depth samples crowd around the cameras and hug surfaces; a uniform batch has ~1700
// swept candidates per tile -- would be bound by its heaviest tile: 1.5 ms instead of 0.1 ms at the ScanNet
// shapes.  A tile whose sweep exceeds its slice size is cut into ns slices of every row range; the owning
// wavefront keeps slice 0 (plain stores, as always) and queues the others, which a second launch of the same
// kernel ("drain") spreads round-robin over the whole chip and adds atomically.  The slice size is at least
// PULL_WORK0 candidates and at least 16 per owned vertex: every extra slice costs one atomic per vertex and
// channel, which pays for a coarse brick under a crowd of points and not for a fine brick in a uniform batch
// (cfg-2 with every tile cut in two: 73 -> 396 us).
constexpr int PULL_WORK0 = 1024;
constexpr int PULL_NS_MAX = 255;
constexpr int PULL_QHDR = 4;      // queue header: {tail, _, workgroups done, _}; items follow

struct PullK {
  int T;
  const int* tile_off;   // T^3 + 1
  const float4* xn;      // (N) normalised coordinates {x,y,z,_}, tile-sorted
  const float* dfeat;    // d-feat rows: row p (tile-sorted order) or, with perm, row perm[p]
  const int* perm;       // NULL: rows are in tile-sorted order
  const float* ggx;      // MODE 1 only: (N,3) cotangent of the coordinate gradient, caller order (see below)
  int64_t ld;            // row pitch in floats (multiple of 4)
  int nl;                // number of pulled levels
  int lev[PULL_MAXL];    // their indices
  int overwrite;         // 1: grad = sum (no zero-fill needed), 0: grad += sum
  int bdiv[PULL_MAXL][3];   // size / T per pulled level and axis where T divides the size, else 0
  float inv_size[PULL_MAXL][3];
  int debug;             // ablation (MISO_DEBUG_PULL, dev only): 1 no pull loop, 2 no groups, 4 no sweep
  int32_t* queue;        // slice queue (see PULL_WORK0) or nullptr: tiles are never split
  int qcap;              // item capacity
  int drain;             // 0: one wavefront per tile, slice 0 + queueing; 1: process the queued slices
  int work0;             // swept candidates per slice
  int blk_off[PULL_MAXL], blk_cap[PULL_MAXL];   // block kernel: partition of a tile's list pool over the levels
};

__device__ __forceinline__ int floor_div(int a, int b) {   // b > 0
  int q = a / b;
  return (a % b != 0 && a < 0) ? q - 1 : q;
}

// continuous index from the normalised coordinate, op for op as common.hpp:axis_coord
__device__ __forceinline__ void cell_of(float xn, int size, int& i0, float& frac) {
  float pos = __fmul_rn(__fsub_rn(__fmul_rn(__fadd_rn(xn, 1.0f), (float)size), 1.0f), 0.5f);  // == /2 exactly
  float f = fminf(fmaxf(floorf(pos), -2.0f), (float)size + 1.0f);
  i0 = (pos == pos) ? (int)f : -2;
  frac = __fsub_rn(pos, f);
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace miso
