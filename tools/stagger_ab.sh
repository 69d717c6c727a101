#!/bin/bash
# dev: MISO_TUNE sweeps of the train kernel (bit 6: compute priority by wavefront slot; bits 8..15: late start of the second slot)
cd $GRAFT_REPO_ROOT
for t in 0 64 $((64+2*256)) $((64+6*256)) 0 64; do MISO_TUNE=$t python3 tools/train_ab.py 2>/dev/null | tail -1 | sed "s/^/tune=$t /"; done
