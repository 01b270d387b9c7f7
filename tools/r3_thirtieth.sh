#!/bin/bash
# HNSW: candidate / result lists in registers (ef <= 128, k <= 64) -- tests first, then A/B by option on one box
out=gpurun_out/r3; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_hnsw_gpu.py tests/test_fuzz_gpu.py -q -m gpu -k "hnsw" > $out/t30_tests.txt 2>&1; echo "hnsw tests exit $?"; tail -3 $out/t30_tests.txt
for rep in 1 2; do for o in 0 1; do
  python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 --opt hnsw_reg_lists=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 hnsw_reg_lists=$o', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"
done; done
python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --efconstruction 200 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 efC=200', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'))"
