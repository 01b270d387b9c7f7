#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/r6_gpu_suite.txt 2>&1
grep -E "passed|failed|error" gpurun_out/r6_gpu_suite.txt | tail -5 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-400
