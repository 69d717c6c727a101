#!/bin/bash
# FETCH_SIZE calibration (tools/ubench/fetch_calib.hip) under rocprofv3 counters; prints KiB per launch and kernel
out=gpurun_out/fetch_calib
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/a -o q -- tools/ubench/fetch_calib > $out/a.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $out/b -o q -- tools/ubench/fetch_calib > $out/b.log 2>&1
rocprofv3 --pmc TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum TCC_READ_sum --kernel-trace --output-format csv -d $out/c -o q -- tools/ubench/fetch_calib > $out/c.log 2>&1
grep calib_ $out/a.log | head -3
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for tag in "abc":
    for f in glob.glob(f"{out}/{tag}/**/q_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
known = {"calib_stream": 512 << 20, "calib_gather_random": (1 << 22) * 64, "calib_gather_stride": (1 << 22) * 64}
for k, d in acc.items():
    line = {c: round(sorted(v)[len(v) // 2], 1) for c, v in sorted(d.items())}      # median over launches
    kb = known.get(k)
    fs = line.get("FETCH_SIZE")
    extra = f"  FETCH_SIZE KiB x 1024 / known bytes (stream: all; gathers: 64 B per row) = {fs * 1024 / kb:.3f}" if kb and fs else ""
    print(k, line, extra)
PY
rm -rf $out/a $out/b $out/c
