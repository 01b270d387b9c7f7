#!/bin/bash
# same-box A/B of the two-thread staging copy: MVS_STAGE_THREADS=1 (the calling thread alone) vs the default
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
: > $O/r6_ingest_mt.log
for rep in 1 2 3; do
  for st in 1 2 3 4; do
    echo "MVS_STAGE_THREADS=$st" >> $O/r6_ingest_mt.log
    MVS_STAGE_THREADS=$st MVS_INGEST_PROFILE=1 duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 IDMap,Flat 2>&1 | grep -E "ingestprofile|ingestrate" >> $O/r6_ingest_mt.log
  done
done
cat $O/r6_ingest_mt.log | cut -c1-200
timeout 900 python -m pytest tests/test_flat_gpu.py tests/test_boundary_driver_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
