#!/bin/bash
O=$1
timeout 2400 python3 -m pytest tests/test_collect_gpu.py tests/test_flat_gpu.py tests/test_flat_shadow_gpu.py tests/test_prefilter_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -8
SHAPES="1250000 1000000 10000000" OPTS="cl_fbucket=0 none" bash tools/r4_steps/shapes.sh $O 2>&1 | tail -7
