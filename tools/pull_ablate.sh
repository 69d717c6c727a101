#!/bin/bash
# dev: phases of the pull at cfg-2 by ablation (MISO_DEBUG_PULL bits: 1 no vertex loops / push steps, 2 no staging,
# 16 no per-tile stage at all, 32 no routing, 512 no push)
out=${1:-gpurun_out/ablate.txt}
base=${2:-0}
: > $out
for bits in 48 16 2 1 0; do
  v=$((bits + base))
  echo -n "debug $v: " >> $out
  MISO_DEBUG_PULL=$v python tools/pull_bench.py 2>/dev/null | tail -1 >> $out
done
cat $out
