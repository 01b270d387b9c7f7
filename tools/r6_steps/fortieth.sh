#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 600 python -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "beyond_128 and 4096" 2>&1 | grep -E "^E |FaissException|Error in" | head -8 | cut -c1-400
