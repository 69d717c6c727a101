#!/bin/bash
# dev: SQ issue / wait counters per kernel of any python tool (two PMC passes, no trace domains mixed in):
#   tools/pmc_sq_any.sh <out-name> <script.py> [args]
name=$1; shift
out=gpurun_out/$name
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM --kernel-trace --output-format csv -d $out/a -o q -- python3 "$@" > $out/a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/b -o q -- python3 "$@" > $out/b.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob, collections, re
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in "ab":
    fs = glob.glob(out + f"/{tag}/**/q_counter_collection.csv", recursive=True)
    if not fs:
        print("no counters for pass", tag, open(out + f"/{tag}.log").read()[-600:]); continue
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"miso::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    print(k, "launches", len(next(iter(d.values()))), {c: round(v, 1) for c, v in sorted(avg.items())})
PY
