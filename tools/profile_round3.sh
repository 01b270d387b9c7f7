#!/bin/bash
# Round-3 measurement pass (run on the GPU box): every number DESIGN.md section 5 quotes for round 3 comes from these files.
# usage: tools/profile_round3.sh [part ...]   parts: headline chunk c2 c3 c3ip c4 c5 shard sens ingest pmc harness  (default: all but harness)
out=gpurun_out/r3p; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
parts=${@:-headline chunk c2 c3 c3ip c4 c5 shard sens ingest pmc}
kstats() { # <tag> <bench args...>: rocprofv3 kernel-trace stats of the same command
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tag -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-configs --no-host-pointer "$@" > $out/${tag}_trace_bench.json 2> $out/${tag}_trace.err
  f=$(find $out/trace_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -16 "$f" > $out/${tag}_kernel_stats.csv && python3 tools/kstats_print.py "$f" 2>/dev/null | head -22 > $out/${tag}_step_kernels.txt
  rm -rf $out/trace_$tag
}
pmc() { # <tag> <counter> <bench args...>
  tag=$1; c=$2; shift; shift
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_${tag}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-host-pointer "$@" > /dev/null 2> $out/${tag}_pmc_$c.err
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
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/headline_bench.json 2> $out/headline_bench.err; cut -c1-300 $out/headline_bench.json
  kstats headline ;;
chunk) python3 bench.py --chunk 2048 --no-cpu-baseline > $out/headline_chunk2048_bench.json 2>/dev/null; cut -c1-200 $out/headline_chunk2048_bench.json ;;
c2) python3 bench.py --rows 1000000 > $out/c2_bench.json 2>/dev/null; cut -c1-200 $out/c2_bench.json; kstats c2 --rows 1000000 ;;
c3) python3 bench.py --index IVF4096,Flat --data clustered > $out/c3_ivf_bench.json 2> $out/c3.err; cut -c1-300 $out/c3_ivf_bench.json
    kstats c3_ivf --index IVF4096,Flat --data clustered ;;
c3ip) python3 bench.py --index IVF4096,Flat --data clustered --metric IP --no-cpu-baseline > $out/c3_ivf_ip_bench.json 2>/dev/null; cut -c1-200 $out/c3_ivf_ip_bench.json ;;
c4) python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --cpu-seconds 10 > $out/c4_shard_bench.json 2> $out/c4.err; cut -c1-300 $out/c4_shard_bench.json
    tools/pmc_sq.sh r3p/sq_c4 --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-configs --no-host-pointer > /dev/null; cp gpurun_out/r3p/sq_c4/pmc_sq.txt $out/c4_pmc_sq.txt; cat $out/c4_pmc_sq.txt ;;
c5) python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 > $out/c5_hnsw_bench.json 2> $out/c5.err; cut -c1-300 $out/c5_hnsw_bench.json
    python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --efconstruction 200 --no-cpu-baseline > $out/c5_hnsw_efc200_bench.json 2>/dev/null; cut -c1-200 $out/c5_hnsw_efc200_bench.json ;;
shard) python3 tools/shard_overhead.py > $out/shard_overhead_virtual.txt 2>&1; grep -v amdgpu.ids $out/shard_overhead_virtual.txt ;;
sens) KINDS="uniform clustered normalised offset integer dup10 sift_like" timeout 900 python3 tools/collect_sensitivity.py > $out/collect_sensitivity.txt 2>&1; grep -v amdgpu.ids $out/collect_sensitivity.txt ;;
ingest) ( time duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 ) > $out/ingest_flat_10m.txt 2>&1; cat $out/ingest_flat_10m.txt ;;
pmc)
  rm -f $out/headline_pmc_hbm.txt; pmc headline FETCH_SIZE; pmc headline WRITE_SIZE; cat $out/headline_pmc_hbm.txt
  tools/pmc_sq.sh r3p/sq_headline --no-configs --no-host-pointer > /dev/null; cp gpurun_out/r3p/sq_headline/pmc_sq.txt $out/headline_pmc_sq.txt; cat $out/headline_pmc_sq.txt ;;
harness) timeout 1500 python3 tools/harness_bench.py --n 8841823 --reps 3 > $out/harness_shapes_full.txt 2>&1; grep -v amdgpu.ids $out/harness_shapes_full.txt ;;
esac; done
