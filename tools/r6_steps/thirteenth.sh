#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 900 python -m pytest tests/test_hnsw_gpu.py -x -q -m gpu 2>&1 | tail -6 | cut -c1-300
WGS="1 4 8" timeout 600 python tools/hnsw_build_probe.py 60000 768 40 > $O/r6_hnsw_wg.txt 2>&1; cat $O/r6_hnsw_wg.txt | cut -c1-200
MVS_INGEST_PROFILE=1 WGS="4" timeout 600 python tools/hnsw_build_probe.py 40960 768 40 2>&1 | grep -v amdgpu | tail -4 | cut -c1-200
KIND=clustered WGS="1 4" timeout 600 python tools/hnsw_build_probe.py 100000 768 40 2>&1 | grep -v amdgpu | cut -c1-200
