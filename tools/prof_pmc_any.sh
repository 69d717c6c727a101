#!/bin/bash
# dev: HBM read / write bytes per kernel of any python tool (two PMC passes): tools/prof_pmc_any.sh <out-name> <script.py> [args]
name=$1; shift
out=gpurun_out/$name
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/f -o q -- python3 "$@" > $out/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/w -o q -- python3 "$@" > $out/w.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for tag, idx in (("f", 1), ("w", 2)):
    f = glob.glob(out + f"/{tag}/**/q_counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:80]
        acc[k][idx] += float(r["Counter_Value"])
        if tag == "f":
            acc[k][0] += 1
for k, (n, fsz, wsz) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:10]:
    n = max(n, 1)
    # gfx950: FETCH_SIZE counts 64-B units as KiB/2 -> x2 (MI355X_MICROARCH guide), WRITE_SIZE KiB
    print(f"{k:<80} launches={n:>6} read_MB={2 * fsz * 1024 / n / 1e6:9.1f} write_MB={wsz * 1024 / n / 1e6:9.1f}")
PY
