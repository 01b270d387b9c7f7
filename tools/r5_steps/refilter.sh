#!/bin/bash
O=$1
C3OPTS="ivf_cl_refilter=0 none ivf_cl_refilter=0 none" STEPS=20 bash tools/r5_steps/c3ab.sh $O
C3OPTS="ivf_cl_refilter=0 none" METRIC=IP STEPS=10 bash tools/r5_steps/c3ab.sh $O
timeout 1500 python3 -m pytest tests/test_ivf_probe_prune_gpu.py tests/test_flat_shadow_gpu.py -m gpu -x -q 2>&1 | tail -5
bash tools/r5_steps/ivf_quick.sh $O
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3_refilter STEPS=10 WARMUP=2 DOM="ivf_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O | tail -22
KINDS="clustered" timeout 1500 python3 tools/collect_sensitivity.py 2>&1 | tail -1
