#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
nproc; uptime
for env in "" "OMP_NUM_THREADS=16" "OMP_NUM_THREADS=32" "OMP_NUM_THREADS=32 OMP_WAIT_POLICY=passive" "OMP_NUM_THREADS=64 OMP_PROC_BIND=close"; do
  echo "== $env"
  env $env timeout 300 python tools/dbg_ivf_ties_time.py 2>&1 | grep -E "oracle train|add Index|300 16" | cut -c1-150
done
