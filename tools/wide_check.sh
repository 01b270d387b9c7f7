#!/bin/bash
# wide coarse filter (128 < d <= 512): parity tests, then kernel timings against the f32 kernel
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_collect_wide_gpu.py -q 2>&1 | tail -15 | tee gpurun_out/wide_tests.txt
for cfg in "256 IP 2000000 10000" "384 IP 2000000 10000" "512 IP 2000000 10000" "256 L2 10000000 10000" "384 L2 500000 2000" "256 L2 1000000 64" "512 L2 200000 10000"; do
  set -- $cfg
  for pf in 2 0 -1; do
    echo "== d=$1 $2 n=$3 nq=$4 prefilter=$pf" | tee -a gpurun_out/wide_bench2.txt
    timeout 300 python3 tools/kbench.py --n $3 --nq $4 --d $1 --metric $2 --k 10 --opt prefilter=$pf --reps 3 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/wide_bench2.txt
  done
done
