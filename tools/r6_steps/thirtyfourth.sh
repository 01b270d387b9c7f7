#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_collect_gpu.py tests/test_collect_wide_gpu.py tests/test_ivf_gpu.py tests/test_sharded_inprocess_gpu.py -x -q -m gpu -k "beyond or big_list or big_lists or k_up_to_128 or large_k" 2>&1 | tail -5 | cut -c1-300
