#!/bin/bash
# small batches (what one DuckDB point query is): per-search kernel breakdown at nq = 1 and nq = 64, N = 10 M
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for nq in 1 64; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t34_$nq -- python3 bench.py --nq $nq --no-cpu-baseline --no-configs --no-host-pointer --steps 37 --warmup 2 > $out/t34_$nq.json 2>/dev/null
f=$(find $out/t34_$nq -name "*kernel_stats.csv" | head -1); echo "== nq=$nq"; python3 tools/kstats_search.py "$f" 39 | cut -c1-150; cut -c1-220 $out/t34_$nq.json; rm -rf $out/t34_$nq
done
