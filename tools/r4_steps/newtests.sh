#!/bin/bash
timeout 1800 python3 -m pytest tests/test_collect_gpu.py tests/test_ivf_gpu.py -m gpu -x -q -k "round4 or coarse_quantiser or coarse or kmeans" 2>&1 | tail -6
