#!/bin/bash
# round 3, fifteenth GPU pass: 32 row classes below k = 17? (option cl_nc32_from), per-kernel breakdown of a C3 step after the re-scoring change
out=gpurun_out/r3; mkdir -p $out
for k in 12 14 16; do for f in 17 12; do
  python3 bench.py --k $k --no-cpu-baseline --no-configs --no-host-pointer --steps 5 --warmup 2 --parity-device 256 --opt cl_nc32_from=$f 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('k=$k cl_nc32_from=$f', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in L2 IP; do
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t15_c3$m -- python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find $out/t15_c3$m -name "*kernel_stats.csv" | head -1); python3 tools/kstats_print.py "$f" 2>/dev/null | head -24 > $out/fifteenth_c3${m}_step_kernels.txt; cat $out/fifteenth_c3${m}_step_kernels.txt | cut -c1-150; rm -rf $out/t15_c3$m
done
