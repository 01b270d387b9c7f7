#!/bin/bash
# select kernel on v_readlane / DPP: kernel times at N = 1.25 M and C3, a short test subset
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t36_a -- python3 bench.py --rows 1250000 --no-cpu-baseline --no-configs --no-host-pointer --steps 17 --warmup 2 > $out/t36_a.json 2>/dev/null
f=$(find $out/t36_a -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | head -5 | cut -c1-150; rm -rf $out/t36_a
cut -c1-200 $out/t36_a.json
timeout 600 python3 -m pytest tests/test_collect_gpu.py -q -m gpu -x -k "equals_exact or k_up_to_32 or duplicates or idmap" > $out/t36_tests.txt 2>&1; echo "tests exit $?"; tail -2 $out/t36_tests.txt
