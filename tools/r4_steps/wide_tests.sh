#!/bin/bash
timeout 1800 python3 -m pytest tests/test_collect_wide_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -6
