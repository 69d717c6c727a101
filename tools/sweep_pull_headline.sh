# dev: the heavy-tile split on the headline (uniform) batch and on the ScanNet-shaped iteration
p='import json,sys; d=json.loads(sys.stdin.read()); print("headline ms", round(d["ms_per_step"],5), "pull", round(d["kernels_us"]["grad_pull_kernel"],1))'
echo "== split"; python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "$p"
echo "== no split"; MISO_PULL_NO_SPLIT=1 python bench.py --steps 200 --warmup 20 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "$p"
MISO_E2E_ONLY_PADDED=1 python tools/mapping_e2e_bench.py 2>&1 | grep "step.run\|padded="
