#!/bin/bash
# query_norms_kernel variants: time in the C3 run (steady search + index build), then the tests that depend on the norms' bits
O=$1
MINCALLS=10 TAG=norms ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" bash tools/r4_steps/kstats.sh $O | grep -E "query_norms|one steady"
grep -A20 "one steady" $O/kstats_norms.txt | grep query_norms
timeout 1500 python3 -m pytest tests/test_flat_gpu.py tests/test_ivf_gpu.py tests/test_coarse_matrix_gpu.py -m gpu -x -q 2>&1 | tail -3
