"""Dev: print the kernel timeline of the last captured GridTrainer.train_step found in a rocprofv3 kernel trace of
tools/trainer_bench.py (start offset, duration, gap to the previous kernel's end; microseconds)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_trainer/q_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fwd = [i for i, r in enumerate(rows) if "sdf_fwd_kernel" in r["Kernel_Name"]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 20          # which step (the captured run comes first)
sel = rows[fwd[k] - 12:fwd[k + 1] - 12]
t0 = int(sel[0]["Start_Timestamp"])
prev = None
for r in sel:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (st - prev) / 1e3 if prev else 0.0
    name = r["Kernel_Name"][:100]
    print(f"{(st - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f} {gap:7.1f}  {name}")
    prev = en
