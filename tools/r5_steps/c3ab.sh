#!/bin/bash
# C3 step A/B by option on ONE box: C3OPTS="none ivf_cl_bucket=0" METRIC=L2 bash tools/r5_steps/c3ab.sh <outdir>
O=$1
for o in ${C3OPTS:-none}; do
  extra=""; [ "$o" != "none" ] && extra="--opt ${o//,/ --opt }"
  python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --no-configs --no-host-pointer --steps ${STEPS:-20} --warmup 3 --parity-device 512 --metric ${METRIC:-L2} ${C3ARGS:-} $extra 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; s=j['config'].get('state_sensitivity') or {}
print('C3 ${METRIC:-L2} ${C3ARGS:-} opt=$o qps=%.0f step_ms=%.3f scan_ms=%.4f frac=%.4f frac_step=%.4f cand_per_q=%s/%s pairs=%s/%s bursts=%s parity=%s/%s first_ms=%s other_ms=%s after_ms=%s' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step'], r.get('candidates_rescored_per_query'), r.get('candidates_admitted_per_query'), r.get('probe_pairs_scanned'), r.get('probe_pairs'), r.get('scan_forced_drains'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], s.get('first_call_ms'), s.get('other_batch_ms'), s.get('step_after_other_batch_ms')))" | tee -a $O/c3ab.txt
done
