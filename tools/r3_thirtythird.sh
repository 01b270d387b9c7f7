#!/bin/bash
# HNSW register lists, level 2 (ef <= 256, k <= 256): tests, then the harness shapes at N = 1 M with the lists in LDS / in registers
out=gpurun_out/r3; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_hnsw_gpu.py tests/test_fuzz_gpu.py -q -m gpu -k "hnsw" > $out/t33_tests.txt 2>&1; echo "hnsw tests exit $?"; tail -3 $out/t33_tests.txt
for o in "hnsw_reg_lists=0" "hnsw_reg_lists=1" "hnsw_reg_lists=1 --opt hnsw_visited_lds=4096"; do
  echo "== $o"; timeout 900 python3 tools/harness_bench.py --n 1000000 --ks 11,50,100,200,500 --reps 5 --opt $o 2>&1 | grep -v amdgpu.ids | grep "k=\|run_\|build"
done
