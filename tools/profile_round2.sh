#!/bin/bash
# Round-2 measurement pass (run on the GPU box): every number DESIGN.md section 5 quotes comes from these files.
# usage: tools/profile_round2.sh [part ...]   parts: headline chunk c2 c3 c3ip c4 c5 ingest multi pmc   (default: all)
out=gpurun_out/r2; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
parts=${@:-headline chunk c2 c3 c3ip c4 c5 ingest multi pmc}
kstats() { # <tag> <bench args...>: rocprofv3 kernel-trace stats of the same command
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err
  f=$(find $out/trace_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -14 "$f" > $out/${tag}_kernel_stats.csv
  rm -rf $out/trace_$tag
}
pmc() { # <tag> <counter> <bench args...>
  tag=$1; c=$2; shift; shift
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/${tag}_pmc_$c.err
  f=$(find $out/pmc_${tag}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c >> $out/${tag}_pmc_hbm.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:4]:
    print(f"{sys.argv[2]} kernel={k!r} dispatches={len(v)} sum={sum(v):.6g} mean_per_dispatch={sum(v)/len(v):.6g}")
PY
  rm -rf $out/pmc_${tag}_$c
}
for p in $parts; do case $p in
headline)
  python3 bench.py > $out/headline_bench.json 2> $out/headline_bench.err; cut -c1-300 $out/headline_bench.json
  python3 bench.py --opt prefilter=0 --no-cpu-baseline > $out/headline_f32kernel_bench.json 2>/dev/null
  python3 bench.py --opt prefilter=1 --no-cpu-baseline > $out/headline_bf16x3_bench.json 2>/dev/null
  kstats headline ;;
chunk)   # the DuckDB granularity: <= 2048 queries per search call (src/faiss_extension.cpp:903-925)
  python3 bench.py --chunk 2048 --no-cpu-baseline > $out/headline_chunk2048_bench.json 2>/dev/null; cut -c1-200 $out/headline_chunk2048_bench.json ;;
c2) python3 bench.py --rows 1000000 > $out/c2_bench.json 2>/dev/null; cut -c1-200 $out/c2_bench.json ;;
c3) python3 bench.py --index IVF4096,Flat --data clustered > $out/c3_ivf_bench.json 2> $out/c3.err; cut -c1-300 $out/c3_ivf_bench.json
    kstats c3_ivf --index IVF4096,Flat --data clustered ;;
c3ip) python3 bench.py --index IVF4096,Flat --data clustered --metric IP --no-cpu-baseline > $out/c3_ivf_ip_bench.json 2>/dev/null; cut -c1-200 $out/c3_ivf_ip_bench.json ;;
c4) python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --cpu-seconds 10 > $out/c4_shard_bench.json 2> $out/c4.err; cut -c1-300 $out/c4_shard_bench.json ;;
c5) python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 > $out/c5_hnsw_bench.json 2> $out/c5.err; cut -c1-300 $out/c5_hnsw_bench.json ;;
ingest)  # AddFunction's call pattern: <= 2048-row add() calls from 8 threads under the index lock (:475-547)
  ( time duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 ) > $out/ingest_flat_10m.txt 2>&1; cat $out/ingest_flat_10m.txt ;;
multi)   # the N > 1 flow on this 1-GPU box: two ranks share the device over gloo (correctness of the flow, not a number)
  for m in L2 IP; do
    MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --rows 2000000 --metric $m > $out/multi_2rank_shared_gpu_$m.json 2> $out/multi_$m.err; cut -c1-250 $out/multi_2rank_shared_gpu_$m.json; grep -o '"merged[a-z_]*": [a-z]*' $out/multi_2rank_shared_gpu_$m.json
  done
  MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --rows 2000000 --index IVF256,Flat --data clustered --no-cpu-baseline > $out/multi_2rank_shared_gpu_ivf.json 2> $out/multi_ivf.err; cut -c1-250 $out/multi_2rank_shared_gpu_ivf.json ;;
pmc)
  rm -f $out/headline_pmc_hbm.txt; pmc headline FETCH_SIZE; pmc headline WRITE_SIZE; cat $out/headline_pmc_hbm.txt
  tools/pmc_sq.sh r2/sq_headline > /dev/null; cp gpurun_out/r2/sq_headline/pmc_sq.txt $out/headline_pmc_sq.txt; cat $out/headline_pmc_sq.txt ;;
esac; done
