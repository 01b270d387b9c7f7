#!/bin/bash
# quick IVF parity pass (a subset of tests/test_ivf_gpu.py that covers the coarse filter's paths) -- minutes, not ten
timeout 1500 python3 -m pytest tests/test_ivf_gpu.py -m gpu -x -q -k "coarse_filter or ties or sort_sized or round4 or class_limits or selector or grouping or edge_cases or row_sharded" 2>&1 | tail -8
