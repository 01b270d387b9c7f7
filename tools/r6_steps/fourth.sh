#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 300 python tools/dbg_ip_dups.py > $O/r6_dbg_ip.log 2>&1; echo "dbg rc=$?" >> $O/r6_dbg_ip.log
cut -c1-400 $O/r6_dbg_ip.log
timeout 300 python tools/coarse_bench.py > $O/r6_coarse_bench.log 2>&1; echo "rc=$?" >> $O/r6_coarse_bench.log
cat $O/r6_coarse_bench.log
timeout 1200 python -m pytest -x -q -m gpu tests/test_coarse_matrix_gpu.py tests/test_ivf_gpu.py tests/test_collect_gpu.py tests/test_prefilter_gpu.py > $O/r6_fourth_tests.log 2>&1; echo "rc=$?" >> $O/r6_fourth_tests.log
tail -12 $O/r6_fourth_tests.log | cut -c1-300
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3 bash tools/kstats.sh $O > /dev/null 2>&1
tail -22 $O/kstats_c3.txt | cut -c1-160
grep -o '"ms_per_step": [0-9.]*\|"build_seconds": [0-9.]*' $O/kstats_c3.json
