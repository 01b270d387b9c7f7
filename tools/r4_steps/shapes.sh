#!/bin/bash
# step time of ONE GPU at the (queries x rows) shapes a multi-GPU headline run hands it, and at C2; A/B by option.
# usage: SHAPES="1250000 1000000" OPTS="cl_tab=1 cl_tab=0" bash tools/r4_steps/shapes.sh <outdir>
O=$1
SHAPES=${SHAPES:-"1250000 1000000 2500000 5000000 10000000"}
OPTS=${OPTS:-"none"}
for n in $SHAPES; do for o in $OPTS; do
  extra=""; [ "$o" != "none" ] && extra="--opt ${o//,/ --opt }"
  python3 bench.py --rows $n --no-cpu-baseline --no-configs --no-host-pointer --steps ${STEPS:-10} --warmup 2 --parity-device 512 $extra 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('N=$n opt=$o qps=%.0f step_ms=%.3f scan_ms=%.4f frac=%.4f cand/q=%.1f grid=%d parity=%s/%s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], r['grid'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal']))" | tee -a $O/shapes.txt
done; done
