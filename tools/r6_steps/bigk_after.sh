#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
: > gpurun_out/r6_bigk_after.txt
for nq in 32 256 2048 10000; do
  for k in 200 1000 2048; do
    r=$(python bench.py --nq $nq --k $k --no-cpu-baseline --steps 3 --warmup 2 --no-configs --no-host-pointer --no-ingest 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('candidates_admitted_per_query'))")
    echo "nq=$nq k=$k (defaults) -> ms, candidates/query: $r" | tee -a gpurun_out/r6_bigk_after.txt
  done
done
timeout 900 python -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "beyond_128 or big_list" 2>&1 | tail -3 | cut -c1-200
