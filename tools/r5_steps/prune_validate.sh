#!/bin/bash
O=$1
C3ARGS="--nprobe 4" C3OPTS="ivf_probe_prune=0 none" STEPS=10 bash tools/r5_steps/c3ab.sh $O
C3OPTS="ivf_probe_prune=0 none ivf_probe_prune=0 none" STEPS=20 bash tools/r5_steps/c3ab.sh $O
timeout 1500 python3 -m pytest tests/test_ivf_probe_prune_gpu.py tests/test_flat_shadow_gpu.py -m gpu -x -q 2>&1 | tail -5
bash tools/r5_steps/ivf_quick.sh $O
bash tools/r5_steps/flat_tests.sh $O 2>&1 | tail -5
timeout 1500 python3 tools/collect_sensitivity.py 2>&1 | tail -10 | tee $O/prune_sens.txt
SHAPES="10000000 1250000" bash tools/r4_steps/shapes.sh $O 2>&1 | tail -4
