#!/bin/bash
# planner: >= 7680 rows per split (C2), IVF stream growth test, a spread of N for the planner change
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_ivf_gpu.py -q -m gpu -k "overflow or coarse_filter" > $out/t26_tests.txt 2>&1; echo "ivf tests exit $?"; tail -3 $out/t26_tests.txt
for rows in 1000000 1250000 2000000 3000000 750000 10000000; do
  python3 bench.py --rows $rows --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --parity-device 256 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('grid'), j['parity_device']['labels_equal'])"
done
