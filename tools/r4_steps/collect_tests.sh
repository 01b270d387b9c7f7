#!/bin/bash
timeout 1500 python3 -m pytest tests/test_collect_gpu.py tests/test_flat_gpu.py tests/test_prefilter_gpu.py tests/test_sharded_inprocess_gpu.py -m gpu -x -q 2>&1 | tail -6
