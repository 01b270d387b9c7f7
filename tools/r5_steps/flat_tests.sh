#!/bin/bash
timeout 2400 python3 -m pytest tests/test_flat_shadow_gpu.py tests/test_collect_gpu.py tests/test_flat_gpu.py tests/test_prefilter_gpu.py -m gpu -x -q 2>&1 | tail -8
