#!/bin/bash
# round 3, closing pass after the HNSW changes: C5 twice (a new graph every build), the driver's three commands, the C5 profile files
out=gpurun_out/r3p; mkdir -p $out
for rep in 1 2; do
  python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'), r.get('grid'))"
done
( time timeout 2400 python3 -m pytest tests/ -q -m gpu > $out/full_suite.txt 2>&1 ) 2> $out/full_suite_time.txt; echo "pytest -m gpu exit code $?"; grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $out/full_suite.txt | tail -2; tail -3 $out/full_suite_time.txt
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
( time python3 bench.py > $out/default_bench.json 2> $out/default_bench.err ) 2> $out/default_bench_time.txt; echo "bench exit $?"; cut -c1-300 $out/default_bench.json; tail -3 $out/default_bench_time.txt
tools/profile_round3.sh c5
