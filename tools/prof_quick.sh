#!/bin/bash
# rocprofv3 kernel trace of tools/quick_bench.py (dev): per-kernel averages
out=gpurun_out/prof_quick
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q -- python tools/quick_bench.py > $out/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_quick/q_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:100]:<100} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
