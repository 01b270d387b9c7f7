#!/bin/bash
# round 6 closing pass on the GPU box: default bench line, per-step kernel lists, counter passes.  PARTS="bench kstats pmc" (default all)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
P=${PARTS:-bench kstats pmc}
if [[ $P == *bench* ]]; then
  timeout 1200 python bench.py > $O/r6_default_bench_no_flags.json 2> $O/r6_default_bench.err; echo "bench rc=$?"
  cp $O/bench_detail.json $O/r6_default_bench_detail.json
  head -c 5000 $O/r6_default_bench_no_flags.json; echo
fi
if [[ $P == *kstats* ]]; then
  ROWS=10000000 ARGS="" TAG=h STEPS=10 WARMUP=2 bash tools/kstats.sh $O > /dev/null 2>&1
  ROWS=1250000 ARGS="" TAG=n8 bash tools/kstats.sh $O > /dev/null 2>&1
  ROWS=10000000 ARGS="--metric IP" TAG=hip STEPS=5 WARMUP=2 bash tools/kstats.sh $O > /dev/null 2>&1
  ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3 bash tools/kstats.sh $O > /dev/null 2>&1
  for t in h n8 hip c3; do echo "== $t"; tail -24 $O/kstats_$t.txt | cut -c1-150; grep -o '"ms_per_step": [0-9.]*' $O/kstats_$t.json; done
fi
if [[ $P == *pmc* ]]; then
  TAG=h ARGS="" KERNELS="collect_kernel" bash tools/pmc_all.sh $O > /dev/null 2>&1
  TAG=n8 ARGS="--rows 1250000" KERNELS="collect_kernel" bash tools/pmc_all.sh $O > /dev/null 2>&1
  TAG=c3 ARGS="--index IVF4096,Flat --data clustered" KERNELS="ivf_bf16 coarse_bf16 bucket" bash tools/pmc_all.sh $O > /dev/null 2>&1
  TAG=c4 ARGS="--rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0" KERNELS="big_kernel" bash tools/pmc_all.sh $O > /dev/null 2>&1
  TAG=c5 ARGS="--index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0" KERNELS="hnsw_search" bash tools/pmc_all.sh $O > /dev/null 2>&1
  for t in h n8 c3 c4 c5; do echo "== $t"; cat $O/${t}_pmc.txt | cut -c1-200; done
fi
