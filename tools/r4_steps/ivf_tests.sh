#!/bin/bash
timeout 1800 python3 -m pytest tests/test_ivf_gpu.py tests/test_sharded_inprocess_gpu.py -m gpu -x -q 2>&1 | tail -6
