#!/bin/bash
# big lists, pass A: a quarter of the rows vs all of them, rows per split that decide (options cl_bigk_whole, cl_bigk_per)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
: > gpurun_out/r6_bigk_passa.txt
for nq in 32 256 2048 10000; do
  for opt in "cl_bigk_whole=0 cl_bigk_per=8" "cl_bigk_whole=1 cl_bigk_per=8" "cl_bigk_whole=1 cl_bigk_per=4" "cl_bigk_whole=0 cl_bigk_per=4"; do
    args=""; for o in $opt; do args="$args --opt $o"; done
    for k in 200 1000; do
      r=$(python bench.py --nq $nq --k $k --no-cpu-baseline --steps 3 --warmup 2 --no-configs --no-host-pointer --no-ingest $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline'].get('candidates_admitted_per_query'))")
      echo "nq=$nq k=$k $opt -> ms, candidates/query: $r" | tee -a gpurun_out/r6_bigk_passa.txt
    done
  done
done
