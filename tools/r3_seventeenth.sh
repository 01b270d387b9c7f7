#!/bin/bash
# round 3, seventeenth GPU pass: 16 < d <= 64 and IVF 16 < k <= 31 on the coarse filters -- tests, then the IVF rate at k = 20
out=gpurun_out/r3; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_collect_gpu.py tests/test_ivf_gpu.py tests/test_fuzz_gpu.py tests/test_flat_gpu.py -x -q -m gpu > $out/seventeenth_tests.txt 2>&1; tail -6 $out/seventeenth_tests.txt
for k in 10 20 31; do for o in 1 0; do
  python3 bench.py --index IVF4096,Flat --data clustered --k $k --no-cpu-baseline --steps 5 --warmup 2 --parity-device 512 --opt ivf_collect_k32=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 k=$k ivf_collect_k32=$o', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], j.get('recall_at_10'))"
done; done
