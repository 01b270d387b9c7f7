#!/bin/bash
# round 3, eighth GPU pass: wave-private candidate queue (flat d <= 128, IVF): parity + timings; then the full GPU suite
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_collect_gpu.py tests/test_prefilter_gpu.py tests/test_ivf_gpu.py -x -q -m gpu > $out/eighth_tests.txt 2>&1; tail -6 $out/eighth_tests.txt
for rows in 10000000 1250000 1000000; do
  python3 bench.py --rows $rows --no-cpu-baseline --steps 10 --warmup 2 --parity-device 512 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done
python3 bench.py --chunk 2048 --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('chunk2048', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'])"
python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --parity-device 1024 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['parity_device']['labels_equal'], j.get('recall_at_10'))"
python3 bench.py --index IVF4096,Flat --data clustered --metric IP --no-cpu-baseline --parity-device 1024 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 IP', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['parity_device']['labels_equal'], j.get('recall_at_10'))"
timeout 1500 python3 -m pytest tests -x -q -m gpu > $out/eighth_full_suite.txt 2>&1; tail -8 $out/eighth_full_suite.txt
