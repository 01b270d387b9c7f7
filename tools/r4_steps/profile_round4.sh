#!/bin/bash
# Round-4 measurement pass (run on the GPU box): the files DESIGN.md section 5 / 6 quote. usage: PARTS="shapes chunk c3ip shard sens harness fuzz" bash tools/r4_pass.sh profile_round4
O=$1; P=$O/prof; mkdir -p $P
for part in ${PARTS:-shapes chunk c3ip shard sens kst fuzz harness}; do case $part in
shapes) # what ONE GPU does with the (queries x rows) shapes the decompositions of an N-GPU headline run hand it
  rm -f $P/shard_shapes.txt
  for shape in "10000 10000000" "10000 5000000" "10000 2500000" "10000 1250000" "5000 10000000" "5000 5000000" "5000 2500000" "2500 5000000" "1250 10000000" "10000 1000000"; do
    set -- $shape
    python3 bench.py --nq $1 --rows $2 --no-cpu-baseline --no-configs --no-host-pointer --steps 20 --warmup 3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('nq=$1 N=$2 qps=%.0f step_ms=%.3f scan_ms=%.4f frac=%.4f frac_step=%.4f cand/q=%.1f grid=%d' % (j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['frac_step'], r['candidates_rescored_per_query'], r['grid']))" | tee -a $P/shard_shapes.txt
  done ;;
chunk) python3 bench.py --chunk 2048 --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 2 > $P/headline_chunk2048_bench.json 2>/dev/null; cut -c1-220 $P/headline_chunk2048_bench.json ;;
c3ip) python3 bench.py --index IVF4096,Flat --data clustered --metric IP --no-cpu-baseline --steps 10 --warmup 2 > $P/c3_ivf_ip_bench.json 2>/dev/null; cut -c1-220 $P/c3_ivf_ip_bench.json ;;
shard) python3 tools/shard_overhead.py 2>&1 | grep -v amdgpu.ids | tee $P/shard_overhead_virtual.txt ;;
sens) KINDS="uniform clustered normalised offset integer dup10 sift_like" timeout 900 python3 tools/collect_sensitivity.py 2>&1 | grep -v amdgpu.ids | tee $P/collect_sensitivity.txt
      N=2000000 KINDS=all_dup timeout 300 python3 tools/collect_sensitivity.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $P/collect_sensitivity.txt ;;
kst) for t in "h 10000000" "c2 1000000" "n8 1250000"; do set -- $t; TAG=$1 ROWS=$2 MINCALLS=12 bash tools/r4_steps/kstats.sh $O > /dev/null; cp $O/kstats_$1.txt $P/step_kernels_$1.txt; head -12 $P/step_kernels_$1.txt; done ;;
fuzz) MVS_FUZZ_SCALE=5 timeout 1800 python3 -m pytest tests/test_fuzz_gpu.py -m gpu -q 2>&1 | tail -3 | tee $P/fuzz_soak.txt ;;
harness) timeout 1700 python3 tools/harness_bench.py --n ${HARNESS_N:-8841823} --reps 3 2>&1 | grep -v amdgpu.ids | tee $P/harness_shapes.txt ;;
esac; done
