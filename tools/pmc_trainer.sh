#!/bin/bash
# Counters of the trainer step at the ScanNet (cfg-3) and Newer College (cfg-5) shapes: several PMC
# passes of tools/trainer_bench.py each (no trace domains mixed in; python3 directly under rocprofv3), per-kernel averages
# written to gpurun_out/pmc_trainer/<grid>.json.  tools/pmc_summary.py merges them into profiles/<tag>_pmc_summary.json.
out=gpurun_out/pmc_trainer
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
PASSES=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_VALU_MFMA_BUSY_CYCLES"
 "GRBM_GUI_ACTIVE"
 "TCC_ATOMIC_sum TCC_ATOMIC_WITHOUT_RET_REQ_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum"
 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum"
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"
 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_ATOMIC_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TOTAL_ATOMIC_WITHOUT_RET_sum"
 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"
 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM SQ_INSTS_VALU SQ_INSTS_LDS"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA"
)
for grid in scannet ncd; do
  if [ $grid = scannet ]; then n=540000; else n=6144; fi
  i=0
  for p in "${PASSES[@]}"; do
    GRID=$grid N=$n rocprofv3 --pmc $p --kernel-trace --output-format csv -d $out/${grid}_$i -o t -- python3 tools/trainer_bench.py > $out/${grid}_$i.log 2>&1
    i=$((i+1))
  done
  GRID=$grid N=$n rocprofv3 --kernel-trace --stats --output-format csv -d $out/${grid}_trace -o t -- python3 tools/trainer_bench.py > $out/${grid}_trace.log 2>&1
done
python3 - "$out" <<'PY'
import csv, sys, glob, collections, re, json
out = sys.argv[1]
for grid in ("scannet", "ncd"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{out}/{grid}_[0-9]*/**/t_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"miso::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            if m: acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    res = {}
    for k, d in acc.items():
        res[k] = {c: sum(v) / len(v) for c, v in sorted(d.items())}
        res[k]["launches_sampled"] = min(len(v) for v in d.values())
    st = glob.glob(f"{out}/{grid}_trace/**/t_kernel_stats.csv", recursive=True)
    if st:
        for r in csv.DictReader(open(st[0])):
            m = re.search(r"miso::(?:\(anonymous namespace\)::)?(\w+(?:<[^>]*>)?)", r["Name"])
            if m and m.group(1) in res:
                res[m.group(1)]["avg_us"] = float(r["AverageNs"]) / 1e3
                res[m.group(1)]["calls"] = int(r["Calls"])
    json.dump(res, open(f"{out}/{grid}.json", "w"), indent=1)
    print(grid)
    for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0))[:6]:
        print(" ", k[:70], {c: round(v, 1) for c, v in d.items() if c in ("avg_us", "calls", "FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_ATOMIC_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum")})
PY
# keep the summaries and the two --stats tables; the raw per-dispatch tables are ~100 MB
for grid in scannet ncd; do
  cp $(find $out/${grid}_trace -name "t_kernel_stats.csv" | head -1) $out/${grid}_kernel_stats.csv 2>/dev/null
done
rm -rf $out/scannet_[0-9]* $out/ncd_[0-9]* $out/scannet_trace $out/ncd_trace
