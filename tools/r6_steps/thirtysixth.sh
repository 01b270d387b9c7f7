#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest tests/test_ivf_gpu.py -x -q -m gpu --durations=12 > gpurun_out/r6_ivf_dur.txt 2>&1
grep -E "passed|failed|s call|s setup" gpurun_out/r6_ivf_dur.txt | tail -16 | cut -c1-200
