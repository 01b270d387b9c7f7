#!/bin/bash
# 16 < k <= 32 on the wide stores (32 row classes in flat_bf16_wide_kernel / flat_bf16_big_kernel): tests, then rates
out=gpurun_out/r3; mkdir -p $out
timeout 1500 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/t28_tests.txt 2>&1; echo "wide tests exit $?"; tail -4 $out/t28_tests.txt
for d in 256 768 1536; do for o in 1 0; do
  timeout 900 python3 bench.py --d $d --rows 2000000 --k 20 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --steps 3 --warmup 1 --parity-device 256 --opt cl_k32=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=$d N=2M IP k=20 cl_k32=$o', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done
