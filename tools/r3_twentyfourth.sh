#!/bin/bash
# IVF: no pre-pass at all? (one grouping + packing + a 0.1 ms scan less, cold bounds in the main pass)
for rep in 1 2; do for pp in 0 -1; do for m in L2 IP; do
  python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 10 --warmup 2 --parity-device 1024 --opt ivf_cl_prepass=$pp 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 $m ivf_cl_prepass=$pp', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], j.get('recall_at_10'))"
done; done; done
