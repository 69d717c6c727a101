#!/bin/bash
# usage: VAR=NAME VALS="a b c" tools/ablate_env.sh  -- quick_bench under each value of an env knob (dev)
for v in $VALS; do echo "== $VAR=$v"; env $VAR=$v python tools/quick_bench.py 2>&1 | grep -E "^\[random\] binned" ; done
