#!/bin/bash
# C3 quick A/B + the IVF parity subset
O=$1
C3OPTS="${C3OPTS:-none none}" STEPS=20 bash tools/r5_steps/c3ab.sh $O
C3OPTS="none" METRIC=IP STEPS=10 bash tools/r5_steps/c3ab.sh $O
timeout 1500 python3 -m pytest tests/test_ivf_probe_prune_gpu.py tests/test_flat_shadow_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/r5_steps/ivf_quick.sh $O
KINDS="clustered" timeout 1500 python3 tools/collect_sensitivity.py 2>&1 | tail -1
