#!/bin/bash
# dev: rocprofv3 kernel trace of any python tool: tools/prof_any.sh <out-name> <script.py> [args]; prints per-kernel averages
name=$1; shift
out=gpurun_out/$name
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o q -- python3 "$@" > $out/log.txt 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/q_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    print(f"{r['Name'][:110]:<110} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f}")
PY
