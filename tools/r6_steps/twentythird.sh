#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 900 python tools/ivf_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/r6_ivf_k.txt
NQ=43 timeout 900 python tools/ivf_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a gpurun_out/r6_ivf_k.txt
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
