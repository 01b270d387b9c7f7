#!/bin/bash
# exact re-scoring kernels without ds_bpermute in front of the row loads: kernel times + tests
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t35_a -- python3 bench.py --rows 1250000 --no-cpu-baseline --no-configs --no-host-pointer --steps 17 --warmup 2 > $out/t35_a.json 2>/dev/null
f=$(find $out/t35_a -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | head -4 | cut -c1-150; rm -rf $out/t35_a
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t35_b -- python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --steps 17 --warmup 2 > $out/t35_b.json 2>/dev/null
f=$(find $out/t35_b -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | head -6 | cut -c1-150; rm -rf $out/t35_b
cut -c1-160 $out/t35_a.json; cut -c1-200 $out/t35_b.json
timeout 900 python3 -m pytest tests/test_collect_gpu.py tests/test_ivf_gpu.py -q -m gpu -x > $out/t35_tests.txt 2>&1; echo "tests exit $?"; tail -2 $out/t35_tests.txt
