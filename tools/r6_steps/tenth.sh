#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 900 python -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu 2>&1 | tail -8 | cut -c1-300
timeout 600 python tools/wide_k_bench.py > $O/r6_wide_k.txt 2>&1; cat $O/r6_wide_k.txt | cut -c1-200
D=1536 N=1000000 NQ=1024 KS="10 100" METRICS=IP timeout 600 python tools/wide_k_bench.py > $O/r6_wide_k_1536.txt 2>&1; cat $O/r6_wide_k_1536.txt | cut -c1-200
D=256 N=4000000 NQ=2048 KS="10 100" METRICS=L2 timeout 600 python tools/wide_k_bench.py > $O/r6_wide_k_256.txt 2>&1; cat $O/r6_wide_k_256.txt | cut -c1-200
