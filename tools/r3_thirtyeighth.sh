#!/bin/bash
# coarse quantiser: 128 x 128 workgroup tile with 8 x 8 chains per thread -- tests (bit-exact probes), kernel time, C3
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_ivf_gpu.py -q -m gpu -x -k "coarse_quantiser or coarse_selection or ivf_search_matches" > $out/t38_tests.txt 2>&1; echo "tests exit $?"; tail -2 $out/t38_tests.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t38_b -- python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --steps 17 --warmup 2 > $out/t38_b.json 2>/dev/null
f=$(find $out/t38_b -name "*kernel_stats.csv" | head -1); python3 tools/kstats_search.py "$f" 19 | head -8 | cut -c1-150; rm -rf $out/t38_b
for m in L2 IP; do
  python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 10 --warmup 2 --parity-device 1024 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 $m', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], j.get('recall_at_10'))"
done
