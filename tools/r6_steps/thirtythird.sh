#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 900 python -m pytest tests/test_sharded_inprocess_gpu.py -x -q -m gpu -k "big_lists" 2>&1 | tail -5 | cut -c1-300
python bench.py --index IVF4096,Flat --data clustered --nq 2048 --k 100 --no-cpu-baseline --parity-device 64 --steps 5 --warmup 2 --no-configs --no-host-pointer --no-ingest 2>&1 | tail -1 | cut -c1-1800
