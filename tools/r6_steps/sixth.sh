#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
timeout 300 python tools/coarse_bench.py > $O/r6_coarse_bench.log 2>&1; echo "rc=$?" >> $O/r6_coarse_bench.log
cat $O/r6_coarse_bench.log
timeout 600 python -m pytest -x -q -m gpu tests/test_coarse_matrix_gpu.py tests/test_ivf_probe_prune_gpu.py "tests/test_ivf_gpu.py::test_training_rejects_non_finite_values_wherever_they_sit" > $O/r6_sixth_tests.log 2>&1; echo "rc=$?" >> $O/r6_sixth_tests.log
tail -5 $O/r6_sixth_tests.log | cut -c1-300
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3 bash tools/kstats.sh $O > /dev/null 2>&1
tail -22 $O/kstats_c3.txt | cut -c1-160
grep -o '"ms_per_step": [0-9.]*\|"build_seconds": [0-9.]*' $O/kstats_c3.json
# data sensitivity of the wide stores (C4's shape) and the d = 128 lifecycle rows
N=12500000 D=768 METRIC=IP KINDS="normalised normalised_s03 normalised_s01 outlier" timeout 900 python tools/collect_sensitivity.py > $O/r6_sens_d768_ip.txt 2>&1
N=12500000 D=768 METRIC=L2 KINDS="normalised normalised_s03 normalised_s01 outlier" timeout 900 python tools/collect_sensitivity.py > $O/r6_sens_d768_l2.txt 2>&1
N=10000000 D=128 METRIC=L2 KINDS="uniform clustered interleaved outlier" timeout 600 python tools/collect_sensitivity.py > $O/r6_sens_d128_l2.txt 2>&1
N=10000000 D=128 METRIC=IP KINDS="uniform clustered normalised outlier" timeout 600 python tools/collect_sensitivity.py > $O/r6_sens_d128_ip.txt 2>&1
cat $O/r6_sens_d768_ip.txt $O/r6_sens_d768_l2.txt $O/r6_sens_d128_l2.txt $O/r6_sens_d128_ip.txt | grep -v amdgpu.ids | cut -c1-200
# HNSW evidence
timeout 900 python tools/hnsw_graph_recall.py 100000 768 2000 > $O/r6_hnsw_graph_recall.txt 2>&1; cat $O/r6_hnsw_graph_recall.txt | grep -v amdgpu.ids | cut -c1-300
