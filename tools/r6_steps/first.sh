#!/bin/bash
# round 6, first GPU call: the changed paths' tests (IP bucketed finish, shadow lifecycle, C ABI), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 900 python -m pytest -x -q -m gpu tests/test_flat_shadow_gpu.py tests/test_collect_gpu.py tests/test_prefilter_gpu.py \
  tests/test_flat_gpu.py tests/test_sharded_inprocess_gpu.py tests/test_options_threads_gpu.py > gpurun_out/r6_first_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r6_first_tests.log
tail -15 gpurun_out/r6_first_tests.log
timeout 900 python bench.py > gpurun_out/r6_first_bench.json 2> gpurun_out/r6_first_bench.err
echo "bench rc=$?"
cp gpurun_out/bench_detail.json gpurun_out/r6_first_bench_detail.json 2>/dev/null
cat gpurun_out/r6_first_bench.json | head -c 6000
