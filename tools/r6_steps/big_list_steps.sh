#!/bin/bash
# step tables of the two big-list bench configs (H_k1000, C3_k100) under rocprofv3 --kernel-trace --stats
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
ROWS=10000000 ARGS="--nq 2048 --k 1000" TAG=k1000 STEPS=5 WARMUP=2 DOM="collect_kernel" bash tools/kstats.sh $O > /dev/null 2>&1
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered --nq 2048 --k 100" TAG=c3k100 STEPS=5 WARMUP=2 DOM="collect_kernel" bash tools/kstats.sh $O > /dev/null 2>&1
for t in k1000 c3k100; do echo "== $t"; grep -o '"ms_per_step": [0-9.]*' $O/kstats_$t.json | head -1; head -26 $O/kstats_$t.txt | cut -c1-150; done
