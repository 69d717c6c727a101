#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM counters in their own passes
# (rocprofv3 cannot mix --pmc with the API trace domains on this pool).  Output: gpurun_out/prof_$1/
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/pmc_write.log 2>&1
# fp32 MFMA busy cycles and GPU-active cycles (SQ and GRBM blocks have their own slots; own passes to stay safe)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_mfma -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_grbm -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/pmc_grbm.log 2>&1
find $out -name "*.csv" | head -20
