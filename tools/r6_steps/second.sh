#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 300 python tools/dbg_shadow_extend.py > gpurun_out/r6_dbg_shadow.log 2>&1; echo "dbg rc=$?" >> gpurun_out/r6_dbg_shadow.log
timeout 600 python -m pytest -x -q -m gpu tests/test_coarse_matrix_gpu.py > gpurun_out/r6_coarse_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6_coarse_tests.log
tail -5 gpurun_out/r6_coarse_tests.log
timeout 300 python tools/coarse_bench.py > gpurun_out/r6_coarse_bench.log 2>&1; echo "rc=$?" >> gpurun_out/r6_coarse_bench.log
cat gpurun_out/r6_coarse_bench.log
timeout 900 python -m pytest -x -q -m gpu tests/test_ivf_gpu.py tests/test_ivf_probe_prune_gpu.py "tests/test_configs_gpu.py::test_c3_ivf4096_10m_nprobe32" > gpurun_out/r6_ivf_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6_ivf_tests.log
tail -5 gpurun_out/r6_ivf_tests.log
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3 bash tools/kstats.sh gpurun_out > /dev/null 2>&1
ROWS=10000000 ARGS="--metric IP" TAG=hip STEPS=5 WARMUP=2 bash tools/kstats.sh gpurun_out > /dev/null 2>&1
tail -25 gpurun_out/kstats_c3.txt; tail -25 gpurun_out/kstats_hip.txt
cat gpurun_out/r6_dbg_shadow.log | cut -c1-400
