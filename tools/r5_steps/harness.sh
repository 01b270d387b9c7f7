#!/bin/bash
(timeout 1500 python3 tools/harness_bench.py --n 1000000 --ks 11,100,200,500,1000,2000 2>&1 | grep -v amdgpu; echo "--- hnsw_reg_lists=0 (LDS lists, round 4) ---"; timeout 1500 python3 tools/harness_bench.py --n 1000000 --ks 500,1000,2000 --opt hnsw_reg_lists=0 2>&1 | grep -v amdgpu) | tee $1/harness.txt
