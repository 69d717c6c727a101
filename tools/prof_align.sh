out=gpurun_out/prof_align; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o e -- python tools/align_bench.py > $out/log.txt 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_align/e_kernel_stats.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:90]:<90} calls={r['Calls']:>5} avg_us={float(r['AverageNs'])/1e3:9.1f} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f}")
PY
