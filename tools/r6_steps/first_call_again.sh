#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/first_call
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
MARKS_DIR=$O/first_call rocprofv3 --kernel-trace --output-format csv -d $O/first_call -- python3 $GRAFT_REPO_ROOT/tools/first_call_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/first_call_probe.py --report $O/first_call | head -10 | tee $O/r6_first_call_final.txt
rm -rf $O/first_call
timeout 1500 python -m pytest tests/test_collect_gpu.py tests/test_collect_wide_gpu.py tests/test_flat_shadow_gpu.py -x -q -m gpu 2>&1 | tail -3 | cut -c1-200
