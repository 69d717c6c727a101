cd $GRAFT_REPO_ROOT
for o in lattice brick lattice brick; do echo "== $o"; MISO_ALIGN_ORDER=$o python3 tools/align8_bench.py 2>/dev/null | grep -E "ms_per_iteration|level|us_per" | head -12; done
