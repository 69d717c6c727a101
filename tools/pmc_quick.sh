#!/bin/bash
# SQ counters for the kernels of tools/quick_bench.py (dev): where do the wave cycles go?
out=gpurun_out/pmc_quick
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $out/a -o q -- python tools/quick_bench.py > $out/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/b -o q -- python tools/quick_bench.py > $out/b.log 2>&1
python - <<'PY'
import csv, collections, glob
for part in ("a","b"):
    f = glob.glob(f"gpurun_out/pmc_quick/{part}/*counter_collection.csv")
    if not f: print("no csv", part); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "miso::" not in n: continue
        key = n.split("(")[0].replace("void ","")[:60]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(k)
        for c, v in d.items():
            print(f"    {c:<24}{sum(v)/len(v):>16.0f}  (n={len(v)})")
PY
