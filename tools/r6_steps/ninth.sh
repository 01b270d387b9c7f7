#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
: > $O/r6_ingest_ab.log
for rep in 1 2 3; do
  for lz in 0 1; do
    echo "MVS_LAZY_ADDS=$lz" >> $O/r6_ingest_ab.log
    MVS_LAZY_ADDS=$lz MVS_INGEST_PROFILE=1 duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 IDMap,Flat 2>&1 | grep -E "ingestprofile|ingestrate" >> $O/r6_ingest_ab.log
  done
done
cat $O/r6_ingest_ab.log | cut -c1-220
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | cut -c1-400
