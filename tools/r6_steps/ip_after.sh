#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_flat_gpu.py tests/test_collect_gpu.py tests/test_collect_wide_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
D=128 N=10000000 NQ=2048 KS="10 100 128 1000 2048" METRICS="IP" timeout 900 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/r6_big_k_ip2.txt
