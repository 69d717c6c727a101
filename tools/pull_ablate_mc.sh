#!/bin/bash
# dev: phase ablation of grad_pull_mc_kernel at cfg-2 (MISO_DEBUG_PULL bits: grad_pull_mc.hip McK::debug)
for d in 0 1 2 3 4 7 8 12 16 20; do
  echo -n "debug=$d: "
  MISO_DEBUG_PULL=$d timeout 120 python tools/pull_bench.py 2>&1 | tail -1
done
