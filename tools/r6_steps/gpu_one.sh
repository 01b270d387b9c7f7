#!/bin/bash
# one test selection on the GPU box: SEL="-k expr" FILES="tests/a.py tests/b.py"
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 1500 python -m pytest $FILES -x -q -m gpu $SEL 2>&1 | grep -E "passed|failed|^E |Error in" | tail -8 | cut -c1-300
