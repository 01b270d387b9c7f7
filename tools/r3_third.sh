#!/bin/bash
# round 3, third GPU pass: one-wave-per-SIMD wide kernel (flat_collect_big.hip), sharded search with the device merge, remaining IVF tests
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/third_wide_tests.txt 2>&1; tail -12 $out/third_wide_tests.txt
for big in 1 0; do
  timeout 300 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --parity-device 256 --opt cl_wide_big=$big > $out/third_c4_big$big.json 2>$out/third_c4_big$big.err
  python3 -c "
import json; j=json.load(open('$out/third_c4_big$big.json')); r=j['roofline']
print('C4 cl_wide_big=$big', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['grid'], j.get('parity_device'))" || tail -3 $out/third_c4_big$big.err
done
for big in 1 0; do
  timeout 300 python3 bench.py --rows 2000000 --d 1024 --no-cpu-baseline --parity-device 256 --opt cl_wide_big=$big 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=1024 N=2M cl_wide_big=$big', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], j.get('parity_device'))"
done
timeout 900 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py -x -q -m gpu > $out/third_shard_tests.txt 2>&1; tail -12 $out/third_shard_tests.txt
timeout 600 python3 tools/shard_overhead.py > $out/third_shard_overhead.txt 2>&1; cat $out/third_shard_overhead.txt | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_ivf_gpu.py tests/test_fuzz_gpu.py -q -m gpu > $out/third_ivf_tests.txt 2>&1; tail -12 $out/third_ivf_tests.txt
