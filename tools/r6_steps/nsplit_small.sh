#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
: > gpurun_out/r6_nsplit_small.txt
for rows in 1250000 1000000; do
  for ns in 0 16 24 32 40 48 56 256 384 512; do
    ms=$(python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --opt cl_nsplit=$ns 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
    echo "rows=$rows cl_nsplit=$ns -> $ms" | tee -a gpurun_out/r6_nsplit_small.txt
  done
done
