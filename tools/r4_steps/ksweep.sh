#!/bin/bash
# headline shape at several k: QPS, ms per step, kernel, candidates per query, parity vs the exact f32 kernel
O=$1
for k in ${KS:-10 32 33 64 100 128}; do
  python3 bench.py --k $k --no-cpu-baseline --no-configs --no-host-pointer --steps 5 --warmup 2 --parity-device 256 ${ARGS:-} 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('k=$k qps=%.0f step_ms=%.3f kernel=%s scan_ms=%.3f cand/q=%s parity=%s/%s' % (j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal']))" | tee -a $O/ksweep.txt
done
