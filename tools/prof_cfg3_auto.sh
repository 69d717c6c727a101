cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
MISO_STEP_TILES=auto GRID=scannet N=540000 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cfg3_auto -o t -- python tools/trainer_bench.py > gpurun_out/cfg3_auto.log 2>&1
head -8 gpurun_out/cfg3_auto/*/t_kernel_stats.csv 2>/dev/null | cut -c1-60,100-200 || find gpurun_out/cfg3_auto -name "*stats*"
