// Dense Adam step over one tensor (torch.optim.Adam semantics, amsgrad=False,
// weight_decay=0, maximize=False), as driven by grid_opt/trainer.py:196-228.
// Dense on purpose: moments decay and parameters keep moving for voxels the
// batch did not touch, exactly like the reference.  HBM-bound: 16 B read +
// 12 B written per element (+4 B when the gradient is cleared in the same pass).
#include <math.h>
#include <string.h>

#include "common.hpp"

namespace miso {

// Step scalars from the device, for a step captured in a HIP graph (the bias corrections change every step, kernel
// arguments of a captured launch do not): table[t - 1] = the AdamScalars of step t as the host computes them
// (miso_adam_scalars_table: same code, so the captured step is bit-identical to the launch-by-launch one), step[0] =
// the 1-based step count, advanced by adam_bump_kernel in front of the launches of a step.  table == nullptr: the
// by-value scalars are used.
struct AdamDevK {
  const AdamScalars* table;
  const int32_t* step;
  int32_t table_len;
};

// step += 1 unless the step's loss is NaN (the reference skips optimizer.step() then: its step count does not move)
__global__ void adam_bump_kernel(int32_t* step, const float* __restrict__ guard) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && !(guard != nullptr && !(guard[0] == guard[0]))) step[0] += 1;
}

// The step's loss from its per-workgroup slots and the step count in one launch (a captured trainer step needs both
// between the backward and the Adam launches): total = sum of n floats (pairwise in LDS, a fixed order), then the bump.
// host_ring (pinned host memory, ring_len slots of two words, ring_len a power of two) also receives the total, in slot
// step[1] % ring_len with step[1] counting the launches: the host reads the NaN guard from there without a copy on the
// stream.
__global__ __launch_bounds__(256) void loss_total_bump_kernel(const float* __restrict__ slots, int n,
                                                             float* __restrict__ total, int32_t* step,
                                                             float* host_ring, int ring_len) {
  __shared__ float red[256];
  float acc = 0.0f;
  for (int i = threadIdx.x; i < n; i += 256) acc += slots[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float t = red[0];
    total[0] = t;
    if (step && t == t) step[0] += 1;
    if (host_ring) {
      // slot = {total, launch number}: the number is stored behind a system-scope fence, so a host that finds its
      // launch's number in the slot finds the total there too -- it polls the slot, no event on the stream (an event
      // record is a barrier packet: ~5.6 us between this step's last kernel and the next step's first)
      const int it = step[1];
      step[1] = it + 1;
      float* slot = host_ring + 2 * (it & (ring_len - 1));
      __hip_atomic_store(slot, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __hip_atomic_store(reinterpret_cast<int32_t*>(slot) + 1, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Streaming accesses (the `nt` bit) for the gradient and the moments: they are read and written once per step, 460 of
// the step's 536 MB at cfg-2, and as ordinary accesses they pushed the parameters -- which the next step's forward
// gathers -- out of the 256 MB Infinity Cache.  Measured on the cfg-2 trainer step (rocprofv3, same box): the next
// sdf_train_kernel 83.3 -> 77.1 us, grad_pull_mc_kernel 65.9 -> 61.5 us, the three Adam launches 95.0 -> 90.6 us.
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* q) {
  const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(q));
  return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ void st4_stream(float* q, const float4& v) {
  const f4v t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(q));
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v,
                                                  int64_t n4, int64_t n, AdamScalars a) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = ld4_stream(g + 4 * i);
    float4 mm = ld4_stream(m + 4 * i), vv = ld4_stream(v + 4 * i);
    adam_one(pp.x, gg.x, mm.x, vv.x, a); adam_one(pp.y, gg.y, mm.y, vv.y, a);
    adam_one(pp.z, gg.z, mm.z, vv.z, a); adam_one(pp.w, gg.w, mm.w, vv.w, a);
    reinterpret_cast<float4*>(p)[i] = pp;
    st4_stream(m + 4 * i, mm);
    st4_stream(v + 4 * i, vv);
    if (ZERO) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // scalar tail (numel % 4), handled by the first threads of block 0
  int64_t t = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    adam_one(p[t], g[t], m[t], v[t], a);
    if (ZERO) g[t] = 0.0f;
  }
}

// Same step, skipping what cannot move.  An element whose gradient has been zero in every step so far has
// m = v = 0 and its update is lr * 0 / (0 + eps) = 0: the dense kernel reads 16 B and writes 12 B per element to
// change nothing.  For a submap whose bound is mostly empty space (Newer College: a 120 x 120 x 20 m bound,
// 144 M floats in the fine level, of which a 6144-point batch touches a few thousand) that is the whole cost of
// a training step.  A chunk is 64 floats (256 B: sixteen cells of an x-row at C = 4 -- a sample's corners touch two of
// them; with the 256-float chunks of rounds 1-2 the step of such a level moved 4x the bytes, 27 us of the 92 us
// Newer College trainer step); a chunk is updated if any of its gradients is non-zero or if it has ever been updated
// (`active`, one byte per chunk: once touched, moments decay and the parameters keep moving, exactly as in the dense
// step).  One wavefront takes a SLAB of four chunks at a time (one float4 per lane, a quarter wavefront per chunk) and
// the lanes of chunks that cannot move sit the loads and stores out.  Results are bit-identical to the dense kernel;
// traffic is 4 B per element plus 28 B per active element, and because every non-zero gradient belongs to an active
// chunk, ZERO clears exactly what needs clearing.
constexpr int ADAM_CHUNK = MISO_ADAM_CHUNK;   // floats
constexpr int ADAM_SLAB = 4 * ADAM_CHUNK;     // floats a wavefront takes per slot: one float4 per lane
constexpr int ADAM_UN = 4;                    // slabs in flight per wavefront
static_assert(ADAM_CHUNK == 64, "a chunk is the sixteen float4s of a quarter wavefront");

// one tensor's step by the whole grid (wavefront `wave` of `nwaves`)
template <bool ZERO>
__device__ __forceinline__ void adam_active_body(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, unsigned char* __restrict__ active, int64_t n,
                                                 const AdamScalars& a, bool skip, int64_t wave, int64_t nwaves) {
  const int lane = threadIdx.x & 63, q = lane >> 4;      // q: this lane's chunk within the slab
  const int64_t nslab = n / ADAM_SLAB;             // whole slabs: float4 path
  for (int64_t s0 = wave * ADAM_UN; s0 < nslab; s0 += nwaves * ADAM_UN) {
    float4 gg[ADAM_UN], pp[ADAM_UN], mm[ADAM_UN], vv[ADAM_UN];
    bool act[ADAM_UN];
    // chunks already moving: everything is loaded at once, as in the dense kernel; the others show their
    // gradient first
#pragma unroll
    for (int u = 0; u < ADAM_UN; ++u) {
      const int64_t sl = s0 + u, i = sl * ADAM_SLAB + lane * 4;
      act[u] = false;
      gg[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (sl < nslab) {
        act[u] = active[sl * 4 + q] != 0;
        gg[u] = ld4_stream(g + i);
        if (act[u]) {
          pp[u] = *reinterpret_cast<float4*>(p + i); mm[u] = ld4_stream(m + i);
          vv[u] = ld4_stream(v + i);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < ADAM_UN; ++u) {
      const int64_t sl = s0 + u, i = sl * ADAM_SLAB + lane * 4;
      const bool nz = gg[u].x != 0.f || gg[u].y != 0.f || gg[u].z != 0.f || gg[u].w != 0.f;   // NaN counts
      const bool any = ((__ballot(nz) >> (16 * q)) & 0xffffull) != 0ull;      // a non-zero in this lane's chunk
      const bool go = sl < nslab && (any || act[u]);
      if (__ballot(go) == 0ull) continue;                // wave-uniform: the whole slab stays where it is
      if (!go) continue;
      if (skip) {
        if (ZERO && any) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      if (!act[u]) {
        pp[u] = *reinterpret_cast<float4*>(p + i); mm[u] = ld4_stream(m + i);
        vv[u] = ld4_stream(v + i);
        if ((lane & 15) == 0) active[sl * 4 + q] = 1;
      }
      adam_one(pp[u].x, gg[u].x, mm[u].x, vv[u].x, a); adam_one(pp[u].y, gg[u].y, mm[u].y, vv[u].y, a);
      adam_one(pp[u].z, gg[u].z, mm[u].z, vv[u].z, a); adam_one(pp[u].w, gg[u].w, mm[u].w, vv[u].w, a);
      *reinterpret_cast<float4*>(p + i) = pp[u];
      st4_stream(m + i, mm[u]);
      st4_stream(v + i, vv[u]);
      if (ZERO && any) *reinterpret_cast<float4*>(g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // the ragged end (numel % 256 elements: up to four chunks, the last one partial), element-wise by the first wavefront
  if (wave == 0) {
    for (int64_t c = nslab * 4; c * ADAM_CHUNK < n; ++c) {
      const int64_t i = c * ADAM_CHUNK + lane;
      const bool in = i < n;
      const bool nz = in && g[i] != 0.f;
      const bool was = active[c] != 0;
      if (__ballot(nz) != 0ull || was) {
        if (in) {
          if (!skip) adam_one(p[i], g[i], m[i], v[i], a);
          if (ZERO) g[i] = 0.0f;
        }
        if (!skip && !was && lane == 0) active[c] = 1;
      }
    }
  }
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_active_kernel(float* __restrict__ p, float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         unsigned char* __restrict__ active, int64_t n,
                                                         AdamScalars a, const float* __restrict__ guard,
                                                         AdamDevK dev) {
  if (dev.table) a = dev.table[min(max(dev.step[0], 1), dev.table_len) - 1];     // a captured step: see AdamDevK
  // guard (optional, device): the step's loss.  NaN => the reference skips backward and optimizer step
  // (grid_opt/trainer.py:213-219); here the launch leaves parameters, moments and flags alone (and still clears
  // the consumed gradients when asked), so the host need not read the loss back before launching.
  const bool skip = guard != nullptr && !(guard[0] == guard[0]);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  adam_active_body<ZERO>(p, g, m, v, active, n, a, skip, wave, nwaves);
}

// The same step driven by flags instead of by reading the gradient (miso_adam_touched): the scatter kernels set
// touched[c] when they put a non-zero into chunk c (common.hpp:touch_chunk), so a chunk is stepped iff
// active[c] | touched[c].  A sparse level is latency, not bytes -- the Newer College fine level has 2.3 M flag bytes of
// which a batch sets a few per cent, all in the rows around the sensor:
//  * a lane reads the four flag bytes of one slab as ONE 32-bit word (the arrays are 4-byte aligned);
//  * a wavefront looks at ADAM_SLOTS groups of 16 consecutive slabs per round, the groups far apart in the level
//    (slot t of wavefront w: group t * nwaves + w), all flag words loaded before any is used -- one round trip decides
//    about 128 slabs, and the busy rows are dealt over all wavefronts instead of landing on the few that own them
//    (256 consecutive slabs per wavefront: 64 us, the owners of the busy rows walking 54 slabs each; a lane per flag
//    byte and 16 consecutive chunks per wavefront, rounds 1-2: 27 us, ~9 dependent rounds per resident wavefront);
//  * then it steps the slabs that have a set byte, up to ADAM_UN of them in flight whatever their number (the lanes of
//    a slab's idle chunks sit out).
// Such a level costs its flags instead of a 576 MB gradient scan.
constexpr int ADAM_GROUP = 16;                // consecutive slabs (flag words) per group: a quarter wavefront
constexpr int ADAM_LOADS = 2;                 // flag words per lane and round
constexpr int ADAM_SLOTS = 4 * ADAM_LOADS;    // groups per wavefront and round
constexpr int ADAM_TUN = 4;                   // slabs in flight per wavefront (8: 202 VGPRs and slower, 31 vs 25 us)

template <bool ZERO>
__device__ __forceinline__ void adam_touched_body(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, unsigned char* __restrict__ active,
                                                  unsigned char* __restrict__ touched, int64_t n, const AdamScalars& a,
                                                  bool skip, int64_t wave, int64_t nwaves) {
  const int lane = threadIdx.x & 63, q = lane >> 4;
  const int64_t nchunks = (n + ADAM_CHUNK - 1) / ADAM_CHUNK, nslab = n / ADAM_SLAB;
  const int64_t nwords = (nchunks + 3) / 4;                // slabs, a ragged last one included
  const int64_t ngroups = (nwords + ADAM_GROUP - 1) / ADAM_GROUP;
  uint32_t* active32 = reinterpret_cast<uint32_t*>(active);
  uint32_t* touched32 = reinterpret_cast<uint32_t*>(touched);
  for (int64_t base = 0; base < ngroups; base += nwaves * ADAM_SLOTS) {
    uint32_t w[ADAM_LOADS], tw[ADAM_LOADS];                 // byte c of word k = chunk c of this lane's slab of load k
#pragma unroll
    for (int k = 0; k < ADAM_LOADS; ++k) {
      const int64_t grp = base + (int64_t)(k * 4 + q) * nwaves + wave;
      const int64_t mine = grp * ADAM_GROUP + (lane & 15);
      uint32_t aw = 0;
      tw[k] = 0;
      const bool whole = grp < ngroups && mine * 4 + 4 <= nchunks;      // all four flag bytes exist
      if (whole) {
        aw = active32[mine]; tw[k] = touched32[mine];
      } else if (grp < ngroups && mine < nwords) {
        for (int c = 0; mine * 4 + c < nchunks; ++c) {
          aw |= (uint32_t)active[mine * 4 + c] << (8 * c);
          tw[k] |= (uint32_t)touched[mine * 4 + c] << (8 * c);
        }
      }
      if (tw[k]) {
        // flag bytes are 0 / 1: the chunks that wake up are the touched ones
        if (whole) {
          touched32[mine] = 0;
          if (!skip && (tw[k] & ~aw)) active32[mine] = aw | tw[k];
        } else {
          for (int c = 0; mine * 4 + c < nchunks; ++c)
            if ((tw[k] >> (8 * c)) & 0xffu) {
              touched[mine * 4 + c] = 0;
              if (!skip) active[mine * 4 + c] = 1;
            }
        }
      }
      w[k] = aw | tw[k];
    }
#pragma unroll
    for (int k = 0; k < ADAM_LOADS; ++k) {
      unsigned long long slabs = __ballot(w[k] != 0u);      // bit b: the slab of lane b (load k) has a chunk to step
      while (slabs) {
        // up to ADAM_TUN slabs in flight whatever their number (all loads issued before the first use)
        int64_t ii[ADAM_TUN];
        bool on[ADAM_TUN], tt[ADAM_TUN];
        float4 gg[ADAM_TUN], pp[ADAM_TUN], mm[ADAM_TUN], vv[ADAM_TUN];
#pragma unroll
        for (int u = 0; u < ADAM_TUN; ++u) {
          on[u] = tt[u] = false;
          ii[u] = 0;
          if (!slabs) continue;                             // wave-uniform
          const int b = __builtin_ctzll(slabs);
          slabs &= slabs - 1;
          const int64_t sl = (base + (int64_t)(k * 4 + (b >> 4)) * nwaves + wave) * ADAM_GROUP + (b & 15);
          const uint32_t wb = __shfl(w[k], b), tb = __shfl(tw[k], b);
          if (sl >= nslab) {
            // the ragged end (numel % 256 elements: the one slab that is not whole), element-wise, chunk by chunk
            for (int c = 0; c < 4; ++c) {
              if (!((wb >> (8 * c)) & 0xffu)) continue;
              const int64_t i = (sl * 4 + c) * ADAM_CHUNK + lane;
              if (i < n) {
                if (!skip) adam_one(p[i], g[i], m[i], v[i], a);
                if (ZERO) g[i] = 0.0f;
              }
            }
            continue;
          }
          on[u] = ((wb >> (8 * q)) & 0xffu) != 0u;
          tt[u] = ((tb >> (8 * q)) & 0xffu) != 0u;
          ii[u] = sl * ADAM_SLAB + lane * 4;
          if (on[u] && !skip) {
            gg[u] = *reinterpret_cast<const float4*>(g + ii[u]);
            pp[u] = *reinterpret_cast<float4*>(p + ii[u]); mm[u] = *reinterpret_cast<float4*>(m + ii[u]);
            vv[u] = *reinterpret_cast<float4*>(v + ii[u]);
          }
        }
#pragma unroll
        for (int u = 0; u < ADAM_TUN; ++u) {
          if (on[u] && !skip) {
            adam_one(pp[u].x, gg[u].x, mm[u].x, vv[u].x, a); adam_one(pp[u].y, gg[u].y, mm[u].y, vv[u].y, a);
            adam_one(pp[u].z, gg[u].z, mm[u].z, vv[u].z, a); adam_one(pp[u].w, gg[u].w, mm[u].w, vv[u].w, a);
            *reinterpret_cast<float4*>(p + ii[u]) = pp[u];
            *reinterpret_cast<float4*>(m + ii[u]) = mm[u];
            *reinterpret_cast<float4*>(v + ii[u]) = vv[u];
          }
          if (ZERO && tt[u]) *reinterpret_cast<float4*>(g + ii[u]) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
}

template <bool ZERO>
__global__ __launch_bounds__(256) void adam_touched_kernel(float* __restrict__ p, float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          unsigned char* __restrict__ active,
                                                          unsigned char* __restrict__ touched, int64_t n,
                                                          AdamScalars a, const float* __restrict__ guard,
                                                          AdamDevK dev) {
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  if (dev.table) a = dev.table[min(max(dev.step[0], 1), dev.table_len) - 1];
  const bool skip = guard != nullptr && !(guard[0] == guard[0]);
  adam_touched_body<ZERO>(p, g, m, v, active, touched, n, a, skip, wave, nwaves);
}

// Several tensors in ONE launch (miso_adam_step_dev_multi), each stepped by its gradient or by its `touched` flags: the
// levels of a grid differ by a factor of eight each, and as
// launches of their own the two coarse ones of cfg-2 are 8.5 + 13.4 us of ramp and tail for 67 MB (the fine one: 74.5 us
// for 469 MB).  Every wavefront walks the tensors in turn, the grid sized for the largest.
struct AdamSegK {
  float* p; float* g; float* m; float* v;
  unsigned char* active;
  unsigned char* touched;      // or nullptr: the tensor is stepped by reading its gradient
  int64_t n;
  int zero;
};
struct AdamMultiK {
  int count;
  AdamSegK seg[MISO_ADAM_MAX_TENSORS];
};
__global__ __launch_bounds__(256) void adam_active_multi_kernel(AdamMultiK k, AdamScalars a, const float* __restrict__ guard,
                                                               AdamDevK dev) {
  if (dev.table) a = dev.table[min(max(dev.step[0], 1), dev.table_len) - 1];
  const bool skip = guard != nullptr && !(guard[0] == guard[0]);
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int t = 0; t < k.count; ++t) {
    const AdamSegK& sg = k.seg[t];
    if (sg.touched) {
      if (sg.zero) adam_touched_body<true>(sg.p, sg.g, sg.m, sg.v, sg.active, sg.touched, sg.n, a, skip, wave, nwaves);
      else adam_touched_body<false>(sg.p, sg.g, sg.m, sg.v, sg.active, sg.touched, sg.n, a, skip, wave, nwaves);
    } else if (sg.zero) {
      adam_active_body<true>(sg.p, sg.g, sg.m, sg.v, sg.active, sg.n, a, skip, wave, nwaves);
    } else {
      adam_active_body<false>(sg.p, sg.g, sg.m, sg.v, sg.active, sg.n, a, skip, wave, nwaves);
    }
  }
}

static AdamScalars adam_scalars(double lr, double b1, double b2, double eps, int step) {
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  AdamScalars a;
  a.one_minus_b1 = (float)(1.0 - b1);
  a.b2 = (float)b2;
  a.one_minus_b2 = (float)(1.0 - b2);
  a.neg_step_size = (float)(-(lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  return a;
}

hipError_t launch_loss_total_bump(const float* slots, int n, float* total, int32_t* step, float* host_ring,
                                  int ring_len, hipStream_t s) {
  loss_total_bump_kernel<<<1, 256, 0, s>>>(slots, n, total, step, host_ring, ring_len);
  return hipGetLastError();
}

hipError_t launch_adam_bump(int32_t* step, const float* guard, hipStream_t s) {
  adam_bump_kernel<<<1, 64, 0, s>>>(step, guard);
  return hipGetLastError();
}

void adam_scalars_table(double lr, double b1, double b2, double eps, int first_step, int count, float* out) {
  static_assert(sizeof(AdamScalars) == 6 * sizeof(float), "table rows are 6 floats");
  for (int i = 0; i < count; ++i) {
    const AdamScalars a = adam_scalars(lr, b1, b2, eps, first_step + i);
    memcpy(out + 6 * (size_t)i, &a, sizeof(a));
  }
}

hipError_t launch_adam_active(float* p, float* g, float* m, float* v, unsigned char* active, int64_t n, double lr,
                              double b1, double b2, double eps, int step, int zero_grad, const float* guard,
                              hipStream_t s, const float* table, const int32_t* step_dev, int table_len) {
  if (n == 0) return hipSuccess;
  const AdamScalars a = adam_scalars(lr, b1, b2, eps, step);
  const AdamDevK dev{reinterpret_cast<const AdamScalars*>(table), step_dev, table_len};
  const int64_t nslabs = (n + ADAM_SLAB - 1) / ADAM_SLAB;
  int64_t blocks = (nslabs + 4 * ADAM_UN - 1) / (4 * ADAM_UN);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (zero_grad) adam_active_kernel<true><<<(unsigned)blocks, 256, 0, s>>>(p, g, m, v, active, n, a, guard, dev);
  else adam_active_kernel<false><<<(unsigned)blocks, 256, 0, s>>>(p, g, m, v, active, n, a, guard, dev);
  return hipGetLastError();
}

hipError_t launch_adam_active_multi(const miso_adam_tensor_t* t, int count, double lr, double b1, double b2, double eps,
                                    int step, const float* table, int table_len, const int32_t* step_dev,
                                    const float* guard, hipStream_t s) {
  AdamMultiK k;
  memset(&k, 0, sizeof(k));
  int64_t most = 0;
  for (int i = 0; i < count; ++i) {
    if (t[i].numel == 0) continue;
    AdamSegK& sg = k.seg[k.count++];
    sg.p = t[i].param; sg.g = t[i].grad; sg.m = t[i].exp_avg; sg.v = t[i].exp_avg_sq; sg.active = t[i].active;
    sg.touched = t[i].touched;
    sg.n = t[i].numel; sg.zero = t[i].zero_grad;
    // the grid each tensor would get as a launch of its own (launch_adam_active / launch_adam_touched); the largest wins
    int64_t b;
    if (sg.touched) {
      const int64_t nwords = ((sg.n + ADAM_CHUNK - 1) / ADAM_CHUNK + 3) / 4;
      const int64_t per_block = 4 * (int64_t)ADAM_SLOTS * ADAM_GROUP;
      b = (nwords + per_block - 1) / per_block;
    } else {
      const int64_t nslabs = (sg.n + ADAM_SLAB - 1) / ADAM_SLAB;
      b = (nslabs + 4 * ADAM_UN - 1) / (4 * ADAM_UN);
    }
    most = b > most ? b : most;
  }
  if (k.count == 0) return hipSuccess;
  const AdamDevK dev{reinterpret_cast<const AdamScalars*>(table), step_dev, table_len};
  int64_t blocks = most < 1 ? 1 : most;
  if (blocks > 256 * 16) blocks = 256 * 16;
  adam_active_multi_kernel<<<(unsigned)blocks, 256, 0, s>>>(k, adam_scalars(lr, b1, b2, eps, step), guard, dev);
  return hipGetLastError();
}

hipError_t launch_adam_touched(float* p, float* g, float* m, float* v, unsigned char* active, unsigned char* touched,
                               int64_t n, double lr, double b1, double b2, double eps, int step, int zero_grad,
                               const float* guard, hipStream_t s, const float* table, const int32_t* step_dev,
                               int table_len) {
  if (n == 0) return hipSuccess;
  const AdamScalars a = adam_scalars(lr, b1, b2, eps, step);
  const AdamDevK dev{reinterpret_cast<const AdamScalars*>(table), step_dev, table_len};
  const int64_t nwords = ((n + ADAM_CHUNK - 1) / ADAM_CHUNK + 3) / 4;
  const int64_t per_block = 4 * (int64_t)ADAM_SLOTS * ADAM_GROUP;      // slabs four wavefronts look at per round
  int64_t blocks = (nwords + per_block - 1) / per_block;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (zero_grad) adam_touched_kernel<true><<<(unsigned)blocks, 256, 0, s>>>(p, g, m, v, active, touched, n, a, guard, dev);
  else adam_touched_kernel<false><<<(unsigned)blocks, 256, 0, s>>>(p, g, m, v, active, touched, n, a, guard, dev);
  return hipGetLastError();
}

hipError_t launch_adam(float* p, float* g, float* m, float* v, int64_t n, double lr, double b1,
                       double b2, double eps, int step, int zero_grad, hipStream_t s) {
  if (n == 0) return hipSuccess;
  double bc1 = 1.0 - pow(b1, (double)step);
  double bc2 = 1.0 - pow(b2, (double)step);
  AdamScalars a;
  a.one_minus_b1 = (float)(1.0 - b1);
  a.b2 = (float)b2;
  a.one_minus_b2 = (float)(1.0 - b2);
  a.neg_step_size = (float)(-(lr / bc1));
  a.bc2_sqrt = (float)sqrt(bc2);
  a.eps = (float)eps;
  bool aligned = ((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15u) == 0;
  int64_t n4 = aligned ? n / 4 : 0;
  int64_t work = n4 > 0 ? n4 : n;
  unsigned blocks = (unsigned)((work + 255) / 256);
  if (blocks > 256u * 16u) blocks = 256u * 16u;
  if (!aligned) {
    // unaligned storage: scalar path over everything (tail loop covers [0, n))
    blocks = (unsigned)((n + 255) / 256);
  }
  if (zero_grad) adam_kernel<true><<<blocks, 256, 0, s>>>(p, g, m, v, n4, n, a);
  else adam_kernel<false><<<blocks, 256, 0, s>>>(p, g, m, v, n4, n, a);
  return hipGetLastError();
}

}  // namespace miso
