#!/bin/bash
timeout 1200 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_ivf_gpu.py -m gpu -x -q -k "large_k_on_eight or tie_pass_lds or row_shards_equal or large_k_select" 2>&1 | tail -8
