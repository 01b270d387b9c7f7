#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 900 python -m pytest tests/test_hnsw_gpu.py -x -q -m gpu 2>&1 | tail -4 | cut -c1-300
PARTS=bench bash tools/r6_steps/final_measure.sh 2>&1 | cut -c1-4200
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | cut -c1-300
