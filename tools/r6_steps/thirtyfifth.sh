#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_ivf_gpu.py tests/test_sharded_inprocess_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "tie or ties or ivf or IVF or c3" > gpurun_out/r6_ties_suite.txt 2>&1
grep -E "passed|failed|Error|assert " gpurun_out/r6_ties_suite.txt | tail -8 | cut -c1-300
