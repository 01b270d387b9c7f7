#!/bin/bash
# round 3, the last closing pass: the driver's three commands on the committed tree
out=gpurun_out/r3p; mkdir -p $out
( time timeout 2400 python3 -m pytest tests/ -q -m gpu > $out/full_suite.txt 2>&1 ) 2> $out/full_suite_time.txt; echo "pytest -m gpu exit code $?"; grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $out/full_suite.txt | tail -2; tail -3 $out/full_suite_time.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
( time python3 bench.py > $out/default_bench.json 2> $out/default_bench.err ) 2> $out/default_bench_time.txt; echo "bench exit $?"; cut -c1-300 $out/default_bench.json; tail -3 $out/default_bench_time.txt
