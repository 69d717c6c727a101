for d in 0 1 2 4; do echo "== MISO_DEBUG_PULL=$d"; MISO_DEBUG_PULL=$d python tools/quick_bench.py 2>&1 | grep -E "^\[random\] binned" ; done
