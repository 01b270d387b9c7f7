#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
timeout 300 python tools/dbg_ivf_ties_time.py 2>&1 | grep -v amdgpu | tail -14
