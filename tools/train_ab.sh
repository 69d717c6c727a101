#!/bin/bash
# dev: A/B of differently built libraries (_ab/*.so, not tracked) on the headline workload; each twice, interleaved
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  python3 tools/train_ab.py 2>/dev/null | tail -1
  for f in _ab/*.so; do
    [ -f "$f" ] && MISO_HIP_LIB=$PWD/$f python3 tools/train_ab.py 2>/dev/null | tail -1
  done
done
