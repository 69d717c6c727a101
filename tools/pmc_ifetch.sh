#!/bin/bash
# dev: instruction-fetch counters of the step's kernels (is a 44 KB straight-line kernel waiting on its own code?)
out=gpurun_out/pmc_if
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_IFETCH InstrFetchLatency SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/a -o q -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/a.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/b -o q -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/b.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in "ab":
    fs = glob.glob(out + f"/{tag}/**/q_counter_collection.csv", recursive=True)
    if not fs:
        print("no counters for pass", tag, open(out + f"/{tag}.log").read()[-800:]); continue
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"miso::(\w+)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, {c: round(v, 1) for c, v in sorted(avg.items())})
PY
