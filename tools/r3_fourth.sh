#!/bin/bash
# round 3, fourth GPU pass: big kernel after the hazard fix, register pre-pass, heavy-query overflow, io cross tests, sensitivity table
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/fourth_wide_tests.txt 2>&1; tail -6 $out/fourth_wide_tests.txt
for big in 1; do
  timeout 300 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --parity-device 256 --opt cl_wide_big=$big > $out/fourth_c4_big$big.json 2>$out/fourth_c4_big$big.err
  python3 -c "
import json; j=json.load(open('$out/fourth_c4_big$big.json')); r=j['roofline']
print('C4 cl_wide_big=$big', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['grid'], r['candidates_rescored_per_query'], j.get('parity_device'))" || tail -3 $out/fourth_c4_big$big.err
  timeout 300 python3 bench.py --rows 2000000 --d 1024 --no-cpu-baseline --parity-device 256 --opt cl_wide_big=$big 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=1024 N=2M cl_wide_big=$big', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], j.get('parity_device'))"
done
timeout 900 python3 -m pytest tests/test_collect_gpu.py tests/test_prefilter_gpu.py -x -q -m gpu > $out/fourth_collect_tests.txt 2>&1; tail -8 $out/fourth_collect_tests.txt
for sr in 1 0; do for rows in 10000000 1250000 1000000; do
  python3 bench.py --rows $rows --no-cpu-baseline --steps 10 --warmup 2 --parity-device 512 --opt cl_seed_regs=$sr 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=$rows cl_seed_regs=$sr', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done
python3 bench.py --chunk 2048 --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('chunk2048', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'])"
timeout 600 python3 -m pytest tests/test_ivf_gpu.py -q -m gpu -k "exact_ties" > $out/fourth_ivf_tests.txt 2>&1; tail -4 $out/fourth_ivf_tests.txt
timeout 600 python3 -X faulthandler -m pytest tests/test_sharded_inprocess_gpu.py -x -q -m gpu > $out/fourth_shard_tests.txt 2>&1; tail -25 $out/fourth_shard_tests.txt
timeout 600 python3 -m pytest tests/test_index_io_gpu.py -q -m gpu > $out/fourth_io_tests.txt 2>&1; tail -25 $out/fourth_io_tests.txt
timeout 900 python3 tools/collect_sensitivity.py > $out/fourth_sensitivity.txt 2>&1; grep -v amdgpu.ids $out/fourth_sensitivity.txt
