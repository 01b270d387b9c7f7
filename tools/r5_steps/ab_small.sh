#!/bin/bash
# same-box A/B of two builds on the N/8 shard, C2 and C3 (prev = libmi355faiss_prev.so), then the tests that exercise the change
O=$1
for v in prev new prev new; do
  if [ $v = prev ]; then export MVS_LIB_PATH=$PWD/duckdb-faiss-ext_amd/libmi355faiss_prev.so; else unset MVS_LIB_PATH; fi
  echo "lib=$v"
  SHAPES="1250000 1000000" bash tools/r4_steps/shapes.sh $O 2>&1 | tail -2
  C3OPTS="none" STEPS=20 bash tools/r5_steps/c3ab.sh $O
done
unset MVS_LIB_PATH
timeout 1500 python3 -m pytest tests/test_collect_gpu.py tests/test_coarse_matrix_gpu.py tests/test_ivf_probe_prune_gpu.py tests/test_flat_shadow_gpu.py -m gpu -x -q 2>&1 | tail -3
