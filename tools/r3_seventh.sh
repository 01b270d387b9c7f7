#!/bin/bash
# round 3, seventh GPU pass: big kernel modes after the flush-barrier fix (first: small checks), C5 bf16 A/B, ablations of the small-N scan
out=gpurun_out/r3; mkdir -p $out
for mode in 0 3; do timeout 300 python3 tools/big_mode_check.py $mode 768 2>&1 | grep -v amdgpu.ids | tail -4; done
timeout 300 python3 tools/big_mode_check.py 3 1024 2>&1 | grep -v amdgpu.ids | tail -4
for mode in 1 2; do timeout 300 python3 tools/big_mode_check.py $mode 768 2>&1 | grep -v amdgpu.ids | tail -2; done
timeout 900 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/seventh_wide_tests.txt 2>&1; tail -4 $out/seventh_wide_tests.txt
timeout 300 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --parity-device 256 > $out/seventh_c4.json 2>$out/seventh_c4.err
python3 -c "
import json; j=json.load(open('$out/seventh_c4.json')); r=j['roofline']
print('C4 big v2', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['grid'], r['candidates_rescored_per_query'], j.get('parity_device'))" || tail -3 $out/seventh_c4.err
for bf in 1 0; do
  MVS_HNSW_STATS=1 timeout 400 python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 --opt hnsw_bf16=$bf > $out/seventh_c5_bf$bf.json 2> $out/seventh_c5_bf$bf.err
  python3 -c "
import json; j=json.load(open('$out/seventh_c5_bf$bf.json')); r=j['roofline']
print('C5 hnsw_bf16=$bf', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"; grep "\[hnsw\]" $out/seventh_c5_bf$bf.err | tail -1
done
for abl in 0 1 3 7; do for rows in 1250000 10000000; do
  python3 bench.py --rows $rows --no-cpu-baseline --steps 6 --warmup 2 --opt cl_abl=$abl 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=$rows cl_abl=$abl (1: no rare path, 3: no fold either, 7: + only the first tile staged)', j['ms_per_step'], r['avg_launch_ms'])"
done; done
timeout 900 python3 -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "overflow or duplicates" > $out/seventh_collect_tests.txt 2>&1; tail -4 $out/seventh_collect_tests.txt
KINDS="all_dup" N=2000000 timeout 300 python3 tools/collect_sensitivity.py > $out/seventh_sensitivity.txt 2>&1; grep -v amdgpu.ids $out/seventh_sensitivity.txt
