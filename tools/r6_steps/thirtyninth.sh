#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "beyond_128 or big_list" 2>&1 | tail -6 | cut -c1-300
D=128 N=200000 NQ=4096 KS="128 200 1000" METRICS="L2" timeout 600 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200
