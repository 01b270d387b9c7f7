#!/bin/bash
# C2 (N = 1 M): the planner's 120 row splits leave the fifth round of workgroups 69 % full; 128 splits of 7 812 rows fill it
for rep in 1 2; do for ns in 0 128 256; do
  python3 bench.py --rows 1000000 --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --parity-device 512 --opt cl_nsplit=$ns 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C2 cl_nsplit=$ns', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'])"
done; done
for ns in 0 64 128; do
  python3 bench.py --rows 500000 --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --opt cl_nsplit=$ns 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=500k cl_nsplit=$ns', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'))"
done
