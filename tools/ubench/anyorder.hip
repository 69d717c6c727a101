// Does a launch with hipExtAnyOrderLaunch start before the previous kernel of the SAME stream has finished on gfx950?
// A: one workgroup spinning for `us` microseconds, stamps its start / end; B: stamps its start.  Built by hand:
//   hipcc --offload-arch=gfx950 -O2 -o anyorder anyorder.hip && ./anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>

__global__ void spin_kernel(uint64_t* stamps, int us) {
  const uint64_t t0 = wall_clock64();                 // 100 MHz
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = t0;
  while (wall_clock64() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[1] = wall_clock64();
}
__global__ void stamp_kernel(uint64_t* stamps, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) stamps[slot] = wall_clock64();
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 64);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int mode = 0; mode < 3; ++mode) {
    for (int grid : {1, 256, 4096}) {
      hipMemsetAsync(d, 0, 64, s);
      hipStreamSynchronize(s);
      spin_kernel<<<grid, 256, 0, s>>>(d, 200);
      if (mode == 0) stamp_kernel<<<1, 64, 0, s>>>(d, 2);
      else hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, nullptr, nullptr,
                                 mode == 1 ? (uint32_t)hipExtAnyOrderLaunch : 0u, d, 2);
      stamp_kernel<<<1, 64, 0, s>>>(d, 3);            // ordinary launch behind both
      hipStreamSynchronize(s);
      uint64_t h[4];
      hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
      printf("mode %d (%s) grid %4d: A ran %.1f us; B started %+.1f us after A's END; C started %+.1f us after A's end\n", mode,
             mode == 0 ? "<<<>>>" : mode == 1 ? "ext any-order" : "ext ordered", grid, (h[1] - h[0]) / 100.0,
             ((double)h[2] - (double)h[1]) / 100.0, ((double)h[3] - (double)h[1]) / 100.0);
    }
  }
  return 0;
}
