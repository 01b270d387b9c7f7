#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
python bench.py --nq 2048 --k 1000 --no-cpu-baseline --parity-device 64 --steps 3 --warmup 1 --no-configs --no-host-pointer --no-ingest 2>&1 | tail -3 | cut -c1-1500
