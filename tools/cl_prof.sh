#!/bin/bash
# collect-path profile: per-kernel times of one headline search (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cl_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cl_prof -o cl -- python3 tools/kbench.py --n 10000000 --opt prefilter=2 --reps 3 > gpurun_out/r2_cl_bench.log 2>&1
tail -2 gpurun_out/r2_cl_bench.log
f=$(find gpurun_out/cl_prof -name "*kernel_stats.csv" | head -1)
python3 tools/kstats_print.py $f | tee gpurun_out/r2_cl_kstats.txt
