#!/bin/bash
# One profiling pass of the round on the GPU box: the contract bench (kernel trace + PMC passes), the 8-submap
# alignment (cfg-4) and the trainer step at the ScanNet / Newer College shapes (cfg-3 / cfg-5), kernel traces only.
# Summaries are produced afterwards in the build container: python tools/pmc_summary.py <tag>; cp of the *_stats.csv.
tag=${1:-r02}
bash tools/profile_bench.sh $tag > gpurun_out/prof_${tag}_bench.log 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out/align -o align -- python tools/align8_bench.py > $out/align.log 2>&1
# HBM counters of the alignment's pair stage (cfg-4), own passes
ITERS=4 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/align_pmc_fetch -o align -- python tools/align8_bench.py > $out/align_pmc_fetch.log 2>&1
ITERS=4 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/align_pmc_write -o align -- python tools/align8_bench.py > $out/align_pmc_write.log 2>&1
GRID=scannet N=540000 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trainer_scannet -o t -- python tools/trainer_bench.py > $out/trainer_scannet.log 2>&1
GRID=ncd N=6144 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trainer_ncd -o t -- python tools/trainer_bench.py > $out/trainer_ncd.log 2>&1
python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
find $out -name "*_kernel_stats.csv" | head; tail -c 400 $out/bench.json
