#!/bin/bash
# dev: phase ablation of grad_pull_mc_kernel at cfg-2 (MISO_DEBUG_PULL bits: grad_pull_mc.hip McK::debug).
# 8 no multiply phase, 16 nothing after the sweep, 64 every other workgroup of each XCD, 128 the first half of the
# workgroups (one per CU: how much of the time is latency a second resident workgroup does not hide).
for d in 0 8 16 64 128; do
  echo -n "debug=$d: "
  MISO_DEBUG_PULL=$d timeout 120 python tools/pull_bench.py 2>&1 | tail -1
done
