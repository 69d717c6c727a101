#!/bin/bash
# dev: SQ counters of the pull kernels (tools/pull_bench.py), one rocprofv3 --pmc pass
out=gpurun_out/pmc_pull
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/a -o p -- python tools/pull_bench.py > $out/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVES --kernel-trace --output-format csv -d $out/b -o p -- python tools/pull_bench.py > $out/b.log 2>&1
python - <<PY
import csv, collections, glob
for d in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "grad_pull" not in k: continue
            key=("block" if "block" in k else "tile") + (",drain" if ", true>" in k or ",true>" in k else "")
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        print(d,k,{c: round(sum(x)/len(x)) for c,x in v.items()}, "n=",len(next(iter(v.values()))))
PY
tail -3 $out/a.log
