#!/bin/bash
# rocprofv3 kernel trace of tools/trainer_bench.py (dev): GRID / N from the environment
out=gpurun_out/prof_trainer
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q -- python tools/trainer_bench.py > $out/log.txt 2>&1
tail -2 $out/log.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_trainer/q_kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.1f}")
PY
