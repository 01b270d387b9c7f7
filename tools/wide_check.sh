#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_collect_wide_gpu.py tests/test_collect_gpu.py -q -x 2>&1 | tail -4 | tee gpurun_out/wide_tests.txt
bash tools/wide_prof.sh 768 IP 2000000
rm -f gpurun_out/wide_bench6.txt
echo "== C4 (12.5M x 768, IP, normalised)" | tee -a gpurun_out/wide_bench6.txt
timeout 900 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --steps 5 --warmup 1 --cpu-seconds 10 2>&1 | tail -1 | tee -a gpurun_out/wide_bench6.txt
