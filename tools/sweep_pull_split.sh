# dev: slice size / drain width of the heavy-tile split (grad_pull.hip) on the ScanNet-shaped iteration
for db in 256 1024; do for w0 in 4096 2048 1024 512; do
echo "== drain_blocks=$db work0=$w0"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/sw; MISO_E2E_ONLY_PADDED=1 MISO_PULL_DRAIN_BLOCKS=$db MISO_PULL_WORK0=$w0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sw -o q -- python tools/mapping_e2e_bench.py 2>&1 | grep "step.run\|padded="
python - <<'PY'
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/sw/q_kernel_trace.csv")) if 'grad_pull' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
print("pull main/drain us:", [round(x) for x in d[-8:]])
PY
done; done
