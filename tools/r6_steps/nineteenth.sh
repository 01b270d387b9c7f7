#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
python tools/batch_state_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6_batch_state.txt
