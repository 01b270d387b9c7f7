#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 900 python -m pytest tests/test_hnsw_gpu.py -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
WGS="1 2 4 8" timeout 600 python tools/hnsw_build_probe.py 60000 768 40 > $O/r6_hnsw_wg.txt 2>&1; cat $O/r6_hnsw_wg.txt | cut -c1-200
WGS="4 8" WAVES="2048" timeout 600 python tools/hnsw_build_probe.py 60000 768 40 > $O/r6_hnsw_wg2.txt 2>&1; cat $O/r6_hnsw_wg2.txt | cut -c1-200
