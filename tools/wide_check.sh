#!/bin/bash
# wide coarse filter (128 < d <= 1024): kernel timings against the f32 kernel, N = 2M, nq = 10k
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/wide_table.txt
for cfg in "256 L2" "256 IP" "384 L2" "384 IP" "512 L2" "512 IP" "768 L2" "768 IP" "1024 L2" "1024 IP"; do
  set -- $cfg
  echo "== d=$1 $2 n=2000000 nq=10000" | tee -a gpurun_out/wide_table.txt
  timeout 300 python3 tools/kbench.py --n 2000000 --nq 10000 --d $1 --metric $2 --k 10 --reps 3 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/wide_table.txt
done
echo "== d=256 L2 n=10000000" | tee -a gpurun_out/wide_table.txt
timeout 300 python3 tools/kbench.py --n 10000000 --nq 10000 --d 256 --metric L2 --k 10 --reps 3 2>&1 | tail -1 | cut -c1-300 | tee -a gpurun_out/wide_table.txt
