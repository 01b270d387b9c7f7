#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_ivf_gpu.py -x -q -m gpu -k "beyond_32 or large_k or select_path or tie_pass_lds" 2>&1 | tail -25 | cut -c1-300
timeout 900 python tools/ivf_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee gpurun_out/r6_ivf_k_after.txt
NQ=43 timeout 900 python tools/ivf_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200 | tee -a gpurun_out/r6_ivf_k_after.txt
