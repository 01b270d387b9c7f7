#!/bin/bash
# wide coarse filter (128 < d <= 768): parity tests, then timings against the f32 kernel
cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_collect_wide_gpu.py -q 2>&1 | tail -15 | tee gpurun_out/wide_tests.txt
for cfg in "768 IP 2000000 10000" "640 L2 2000000 10000" "768 L2 1000000 64"; do
  set -- $cfg
  for pf in 2 0; do
    echo "== d=$1 $2 n=$3 nq=$4 prefilter=$pf" | tee -a gpurun_out/wide_bench3.txt
    timeout 300 python3 tools/kbench.py --n $3 --nq $4 --d $1 --metric $2 --k 10 --opt prefilter=$pf --reps 3 2>&1 | tail -1 | cut -c1-330 | tee -a gpurun_out/wide_bench3.txt
  done
done
echo "== C4 (12.5M x 768, IP, normalised)" | tee -a gpurun_out/wide_bench3.txt
timeout 900 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | tee -a gpurun_out/wide_bench3.txt
