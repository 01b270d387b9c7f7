#!/bin/bash
# round 5: the restaged coarse distance matrix (ivf_coarse_mfma = 2) -- parity, then the C3 step and its kernels against mode 1
O=$1
timeout 1500 python3 -m pytest tests/test_coarse_matrix_gpu.py tests/test_options_threads_gpu.py -m gpu -x -q 2>&1 | tail -4
C3OPTS="none ivf_coarse_mfma=3 ivf_coarse_mfma=1 none ivf_coarse_mfma=3 ivf_coarse_mfma=1" STEPS=20 bash tools/r5_steps/c3ab.sh $O
for m in 2 3 1; do
  MINCALLS=10 TAG=coarse$m OPT="ivf_coarse_mfma=$m" ARGS="--index IVF4096,Flat --data clustered" bash tools/r4_steps/kstats.sh $O | grep -E "coarse|sum of"
done
