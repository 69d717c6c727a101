"""Turn the rocprofv3 outputs of tools/profile_bench.sh into the committed summaries:
profiles/<tag>_kernel_stats.csv (verbatim --stats table) and
profiles/<tag>_pmc_summary.json (per kernel: FETCH_SIZE / WRITE_SIZE averages and HBM bytes
per launch).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters
are in KiB, and on gfx950 FETCH_SIZE reports half of the bytes actually fetched, so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024."""
import collections
import csv
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{src}/trace/bench_kernel_stats.csv", f"profiles/{tag}_kernel_stats.csv")


def short(name):
    m = re.search(r"(?:miso::)?(\w+)(?:<|\()", name.replace("void ", ""))
    base = m.group(1) if m else name[:40]
    # the second ("drain") launch of the gradient pull is its own instantiation <..., true>
    return base + "_drain" if base == "grad_pull_kernel" and re.search(r",\s*true>", name) else base


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for kind in ("fetch", "write", "mfma", "grbm"):
    path = f"{src}/pmc_{kind}/bench_counter_collection.csv"
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        if "miso::" not in r["Kernel_Name"]:
            continue
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    f = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) if d.get("FETCH_SIZE") else 0.0
    w = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) if d.get("WRITE_SIZE") else 0.0
    out[k] = {"FETCH_SIZE_KiB_avg": f, "WRITE_SIZE_KiB_avg": w, "launches_sampled": len(d.get("FETCH_SIZE", [])),
              "hbm_bytes_per_launch": (2 * f + w) * 1024,
              "note": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE halving, KiB units)"}
    # fp32 MFMA busy share.  SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip, in shader cycles
    # (checked: sdf_fwd_kernel issues 4096 chunks x 176 v_mfma_f32_32x32x2_f32 x 64 cycles = 46.14 M, the counter reads
    # 46.14 M); GRBM_GUI_ACTIVE is summed over the 8 XCDs (it reads ~8 x duration x clock).  So the share of the
    # kernel's SIMD-cycles spent issuing MFMAs is busy / (active / 8 * 1024).
    busy, act = d.get("SQ_VALU_MFMA_BUSY_CYCLES"), d.get("GRBM_GUI_ACTIVE")
    if busy and act:
        b, a = sum(busy) / len(busy), sum(act) / len(act)
        out[k].update({"SQ_VALU_MFMA_BUSY_CYCLES_avg": b, "GRBM_GUI_ACTIVE_avg": a,
                       "mfma_busy_frac": b / (a / 8.0 * 1024.0) if a else None})
# cfg-4: the alignment's batched kernels, per alignment level.  The two levels launch the same kernels with different
# grids (28 pairs x the workgroups the level's vertex count needs): launches are grouped by grid size, ascending = level.
align = {}
for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    import glob
    paths = glob.glob(f"{src}/align_pmc_{kind}/**/align_counter_collection.csv", recursive=True)
    if not paths:
        continue
    for r in csv.DictReader(open(paths[0])):
        nm = short(r["Kernel_Name"])
        if nm not in ("pair_stage_kernel", "pair_latent_batch_kernel", "overlap_count_batch_kernel"):
            continue
        align.setdefault(nm, {}).setdefault(int(r["Grid_Size"]), {}).setdefault(cname, []).append(float(r["Counter_Value"]))
for grid in ("scannet", "ncd"):      # cfg-3 / cfg-5 trainer steps (tools/pmc_trainer.sh)
    tp = f"gpurun_out/pmc_trainer/{grid}.json"
    if os.path.exists(tp):
        out[f"trainer_{grid}"] = json.load(open(tp))
if align:
    out["cfg4_align"] = {}
    for nm, by_grid in align.items():
        for lvl, grid in enumerate(sorted(by_grid)):
            d = by_grid[grid]
            f = sum(d.get("FETCH_SIZE", [0.0])) / max(len(d.get("FETCH_SIZE", [])), 1)
            w = sum(d.get("WRITE_SIZE", [0.0])) / max(len(d.get("WRITE_SIZE", [])), 1)
            out["cfg4_align"][f"{nm}_level{lvl}"] = {
                "grid_size": grid, "launches_sampled": len(d.get("FETCH_SIZE", [])), "FETCH_SIZE_KiB_avg": f,
                "WRITE_SIZE_KiB_avg": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                "note": "one launch = all 28 pairs of an iteration; the x2 on FETCH_SIZE holds for 16-byte gathers as for "
                        "coalesced streams (round 6, tools/ubench/fetch_calib.hip: every fill is a 128-byte request of which "
                        "the counter reports 64 bytes)"}
# Stamp the summary with the kernel-source hash of the library the counters were COLLECTED with: every pass's log holds
# the bench line, whose "library" field is miso_version() ("... src=<hash>").  (Stamping the hash of the tree the
# summary is made in would label stale counters as current after a failed or skipped collection run.)
import re
import subprocess
hashes = set()
for name in ("pmc_fetch", "pmc_write", "pmc_mfma", "pmc_grbm"):
    try:
        m = re.findall(r"src=([0-9a-f]{8,})", open(f"gpurun_out/prof_{tag}/{name}.log").read())
    except OSError:
        m = []
    hashes.update(m[-1:])
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
if len(hashes) == 1:
    out["_meta"] = {"source_hash": hashes.pop(), "commit": commit or None,
                    "note": "source_hash = the src= field of miso_version() in the collection runs' bench lines; bench.py "
                            "quotes these figures only while it equals the hash of the library it runs"}
else:
    out["_meta"] = {"error": f"collection logs name {sorted(hashes) or 'no'} kernel-source hash(es)"}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
