#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 2400 python -m pytest tests/test_ivf_gpu.py tests/test_boundary_driver_gpu.py tests/test_index_io_gpu.py -x -q -m gpu > gpurun_out/r6_ivf_suite.txt 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/r6_ivf_suite.txt | tail -8 | cut -c1-300
