#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu -k "beyond_128 or k_up_to_128" 2>&1 | tail -12 | cut -c1-400
D=768 N=2000000 NQ=2048 KS="128 129 200 1000" METRICS="IP L2" timeout 900 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/r6_big_k_wide.txt
D=1536 N=1000000 NQ=43 KS="200 2000" METRICS="IP" timeout 900 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a gpurun_out/r6_big_k_wide.txt
