#!/bin/bash
timeout 1800 python3 -m pytest tests/test_flat_gpu.py tests/test_sharded_inprocess_gpu.py -m gpu -x -q -k "reservoir or large_k or boundary_ties" 2>&1 | tail -12
