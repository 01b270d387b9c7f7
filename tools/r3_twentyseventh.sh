#!/bin/bash
# per-search kernel breakdown of C3 (L2) and C2 on the final tree
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t27_c3 -- python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --steps 17 --warmup 2 > $out/t27_c3.json 2>/dev/null
f=$(find $out/t27_c3 -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | tee $out/t27_c3_search_kernels.txt | cut -c1-150; rm -rf $out/t27_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t27_c2 -- python3 bench.py --rows 1000000 --no-cpu-baseline --no-configs --no-host-pointer --steps 17 --warmup 2 > $out/t27_c2.json 2>/dev/null
f=$(find $out/t27_c2 -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | tee $out/t27_c2_search_kernels.txt | cut -c1-150; rm -rf $out/t27_c2
cut -c1-200 $out/t27_c2.json
