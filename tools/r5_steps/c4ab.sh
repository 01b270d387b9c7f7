#!/bin/bash
# C4's one-GPU shard (FlatIP d=768 N=12.5M) by option on one box: C4OPTS="none cl_big_mode=1" bash tools/r5_steps/c4ab.sh <outdir>
O=$1
for o in ${C4OPTS:-none}; do
  extra=""; [ "$o" != "none" ] && extra="--opt ${o//,/ --opt }"
  python3 bench.py --rows ${C4ROWS:-12500000} --d ${C4D:-768} --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps ${STEPS:-5} --warmup 2 --parity-device 256 $extra 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('C4 d=${C4D:-768} N=${C4ROWS:-12500000} opt=$o qps=%.0f step_ms=%.3f scan_ms=%.3f frac=%.4f frac_step=%.4f kernel=%s cand_per_q=%s parity=%s/%s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step'], r['kernel'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal']))" | tee -a $O/c4ab.txt
done
