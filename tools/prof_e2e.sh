#!/bin/bash
# rocprofv3 kernel trace of tools/mapping_e2e_bench.py (dev): per-kernel averages of the ScanNet-shaped iteration
out=gpurun_out/prof_e2e
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q -- python tools/mapping_e2e_bench.py > $out/log.txt 2>&1
tail -12 $out/log.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_e2e/q_kernel_stats.csv")))
for r in rows[:22]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
