#!/bin/bash
# Runs on the GPU box: headline bench, rocprofv3 kernel-trace stats of the SAME command, and two PMC passes
# (FETCH_SIZE / WRITE_SIZE collected separately, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/<tag>/.
# usage: tools/profile_round.sh <tag> [bench.py args...]   (default: the headline configuration)
tag=${1:-r1}
shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 bench.py --steps 5 --warmup 1 "$@" > $out/bench.json 2> $out/bench.err
cat $out/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 1 "$@" > $out/trace_bench.json 2> $out/trace.err
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); cp "$f" $out/kernel_stats.csv; head -8 $out/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $out/pmc_$c.json 2> $out/pmc_$c.err
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c > $out/pmc_$c.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{sys.argv[2]} kernel={k!r} dispatches={len(v)} sum={sum(v):.6g} mean_per_dispatch={sum(v)/len(v):.6g}")
PY
  cat $out/pmc_$c.txt | head -4
done
rm -rf $out/trace $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
