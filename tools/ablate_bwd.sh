for d in 0 1 8; do echo "== MISO_DEBUG_BWD=$d"; MISO_DEBUG_BWD=$d python tools/quick_bench.py 2>&1 | grep -E "^\[(random|sorted)\]" ; done
