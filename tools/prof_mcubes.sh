#!/bin/bash
# rocprofv3 kernel trace of tools/mcubes_bench.py (dev): per-kernel averages of the marching-cubes launches
out=gpurun_out/prof_mcubes
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q -- python tools/mcubes_bench.py > $out/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_mcubes/q_kernel_stats.csv")))
for r in rows:
    if "mc_" in r['Name']:
        print(f"{r['Name'][:70]:<70} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f}")
PY
tail -1 $out/log.txt
