#!/bin/bash
# N/8 shard and C2: sweeps of the pre-pass rows and of the row splits (fixed costs of a small step)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
: > $O/r6_small_step_sweep.txt
run() { # rows, opts...
  local rows=$1; shift
  local args=""
  for o in "$@"; do args="$args --opt $o"; done
  local ms=$(python bench.py --rows $rows --steps 20 --warmup 3 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest $args 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1)
  echo "rows=$rows $* -> $ms" | tee -a $O/r6_small_step_sweep.txt
}
for rows in 1250000 1000000; do
  run $rows
  for sr in 4096 8192 16384; do run $rows cl_seed_reg_rows=$sr; done
  for ns in 64 96 160 192 256; do run $rows cl_nsplit=$ns; done
  run $rows
done
timeout 900 python -m pytest tests/test_hnsw_gpu.py tests/test_index_io_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-300
