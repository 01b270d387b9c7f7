#!/bin/bash
# probe pruning (round 5): the new parity tests, the IVF quick pass, C3 A/B with / without pruning on one box, per-kernel listing
O=$1
timeout 1500 python3 -m pytest tests/test_ivf_probe_prune_gpu.py tests/test_flat_shadow_gpu.py -m gpu -x -q 2>&1 | tail -30
bash tools/r5_steps/ivf_quick.sh $O
C3OPTS="none ivf_probe_prune=0 none" bash tools/r5_steps/c3ab.sh $O
C3OPTS="none" METRIC=IP bash tools/r5_steps/c3ab.sh $O
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3_prune STEPS=10 WARMUP=2 DOM="ivf_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O | tail -40
KINDS="clustered uniform" timeout 1500 python3 tools/collect_sensitivity.py 2>&1 | tail -5 | tee $O/prune_sens.txt
