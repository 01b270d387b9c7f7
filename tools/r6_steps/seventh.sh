#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 600 python -m pytest -x -q -m gpu "tests/test_collect_gpu.py::test_outlier_rows_stay_out_of_the_store_and_in_every_candidate_set" > $O/r6_outlier_tests.log 2>&1; echo "rc=$?" >> $O/r6_outlier_tests.log
tail -15 $O/r6_outlier_tests.log | cut -c1-300
N=10000000 D=128 METRIC=L2 KINDS="uniform outlier" timeout 600 python tools/collect_sensitivity.py 2>&1 | grep -v amdgpu | tee $O/r6_sens_outlier.txt | cut -c1-200
timeout 300 python tools/coarse_bench.py 2>&1 | grep -v amdgpu | tee $O/r6_coarse_bench.log
timeout 900 python tools/hnsw_graph_recall.py 100000 768 2000 > $O/r6_hnsw_graph_recall.txt 2>&1; cat $O/r6_hnsw_graph_recall.txt | grep -v amdgpu.ids | cut -c1-300
bash tools/r6_steps/hnsw_prof.sh > /dev/null 2>&1; cat $O/r6_hnsw_build.txt | grep -v amdgpu | cut -c1-300
timeout 2400 python -m pytest -x -q -m gpu tests > $O/r6_full_gpu_tests.log 2>&1; echo "rc=$?" >> $O/r6_full_gpu_tests.log
tail -15 $O/r6_full_gpu_tests.log | cut -c1-300
