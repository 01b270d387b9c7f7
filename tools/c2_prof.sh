#!/bin/bash
# per-kernel times of a C2 search (N = 1M, d = 128, nq = 10k)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c2_prof -o c2 -- python3 tools/kbench.py --n 1000000 --reps 5 > gpurun_out/r2_c2_prof.log 2>&1
tail -1 gpurun_out/r2_c2_prof.log | cut -c1-200
f=$(find gpurun_out/c2_prof -name "*kernel_stats.csv" | head -1)
python3 tools/kstats_print.py $f | tee gpurun_out/r2_c2_kstats.txt
