#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/first_call
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
MARKS_DIR=$O/first_call rocprofv3 --kernel-trace --output-format csv -d $O/first_call -- python3 $GRAFT_REPO_ROOT/tools/first_call_probe.py 2>&1 | grep -v amdgpu | tail -5
cd $GRAFT_REPO_ROOT
python3 tools/first_call_probe.py --report $O/first_call | tee $O/r6_first_call.txt
rm -rf $O/first_call
python bench.py --no-cpu-baseline --no-configs --no-host-pointer --no-ingest 2>/dev/null | python -c "import sys,json; d=json.load(open('gpurun_out/bench_detail.json')); print(json.dumps(d['config']['state_sensitivity']))"
D=128 N=10000000 NQ=2048 KS="128 129 200 1000" METRICS=L2 timeout 900 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200
D=128 N=10000000 NQ=32 KS="128 200 1000" METRICS=L2 timeout 900 python tools/wide_k_bench.py 2>&1 | grep -v amdgpu | cut -c1-200
