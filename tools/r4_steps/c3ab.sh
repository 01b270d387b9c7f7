#!/bin/bash
O=$1
for o in ${C3OPTS:-none ivf_cl_xcd=0}; do
  extra=""; [ "$o" != "none" ] && extra="--opt ${o//,/ --opt }"
  python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --no-configs --no-host-pointer --steps 20 --warmup 3 --parity-device 512 --metric ${METRIC:-L2} $extra 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('C3 ${METRIC:-L2} opt=$o qps=%.0f step_ms=%.3f scan_ms=%.4f frac=%.4f frac_8d=%.4f cand_per_q=%s parity=%s/%s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_list_major_8d'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal']))" | tee -a $O/c3ab.txt
done
