#!/bin/bash
# per-kernel times of one wide-d search (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=${1:-768}; M=${2:-IP}; N=${3:-2000000}
mkdir -p gpurun_out/wide_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wide_prof -o w -- python3 tools/kbench.py --n $N --d $D --metric $M --opt prefilter=2 --reps 3 > gpurun_out/wide_prof.log 2>&1
tail -1 gpurun_out/wide_prof.log | cut -c1-200
f=$(find gpurun_out/wide_prof -name "*kernel_stats.csv" | head -1)
python3 tools/kstats_print.py $f | head -24 | tee gpurun_out/wide_kstats_d$D.txt
