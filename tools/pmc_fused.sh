#!/bin/bash
# dev: SQ counters of the fused forward / backward kernels (bench.py step), one rocprofv3 --pmc pass
out=gpurun_out/pmc_fused
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/a -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/a.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES --kernel-trace --output-format csv -d $out/b -o p -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/b.log 2>&1
python - <<PY
import csv, collections, glob, re
for d in ("a","b"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            m=re.search(r"miso::(\w+)", k)
            if not m: continue
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if k.startswith("sdf_") or k.startswith("grad_pull_block"):
            print(d,k,{c: round(sum(x)/len(x)) for c,x in v.items()})
PY
