#!/bin/bash
# round 3, sixteenth GPU pass: coarse filter for 16 < d <= 64 (tests + rate against the round-2 routes)
out=gpurun_out/r3; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_collect_gpu.py tests/test_flat_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $out/sixteenth_tests.txt 2>&1; tail -6 $out/sixteenth_tests.txt
for d in 64 32 24; do for pf in -1 1 0; do
  python3 bench.py --d $d --no-cpu-baseline --no-configs --no-host-pointer --steps 3 --warmup 1 --parity-device 256 --opt prefilter=$pf 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=$d prefilter=$pf', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done
