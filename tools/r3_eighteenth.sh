#!/bin/bash
# round 3, eighteenth GPU pass: the 1536-dim store on the one-wave-per-SIMD kernel (two 768-dim parts per row) -- tests first (a fault
# in the new instance must not take the other steps with it: each step is its own process), then the rate against the f32 kernel
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_collect_gpu.py -x -q -m gpu > $out/eighteenth_tests_a.txt 2>&1; tail -3 $out/eighteenth_tests_a.txt
timeout 1200 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/eighteenth_tests_b.txt 2>&1; tail -6 $out/eighteenth_tests_b.txt
for pf in -1 0; do
  timeout 600 python3 bench.py --d 1536 --rows 2000000 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --steps 3 --warmup 1 --parity-device 256 --opt prefilter=$pf 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=1536 N=2M IP prefilter=$pf', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done
timeout 600 python3 bench.py --d 1536 --rows 2000000 --metric L2 --data clustered --sigma 1.0 --no-cpu-baseline --steps 3 --warmup 1 --parity-device 256 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=1536 N=2M L2', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
timeout 600 python3 bench.py --d 768 --rows 2000000 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --steps 3 --warmup 1 --parity-device 256 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=768 N=2M IP (unchanged instance)', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], j['parity_device']['labels_equal'])"
