#!/bin/bash
# A/B of $MISO_TUNE ablation bits on the kernel timings (dev)
for t in ${TUNES:-0 1}; do echo "== MISO_TUNE=$t"; MISO_TUNE=$t python tools/quick_bench.py 2>&1 | grep -E "binned|^\[random\]" ; done
