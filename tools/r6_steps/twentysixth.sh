#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
ROWS=10000000 ARGS="--nq 2048 --k 1000" TAG=k1000 STEPS=5 WARMUP=2 DOM="collect_kernel" bash tools/kstats.sh $O > /dev/null 2>&1
head -40 $O/kstats_k1000.txt | cut -c1-160
