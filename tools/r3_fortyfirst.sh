#!/bin/bash
# IVF: the pre-pass on the main pass's items (one grouping + packing per search) -- A/B by option, then the IVF tests
out=gpurun_out/r3; mkdir -p $out
for rep in 1 2; do for o in 0 1; do for m in L2 IP; do
  python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 10 --warmup 2 --parity-device 1024 --opt ivf_cl_prepass_shared=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 $m shared=$o', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], j.get('recall_at_10'))"
done; done; done
timeout 1200 python3 -m pytest tests/test_ivf_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "ivf or c3" > $out/t41_tests.txt 2>&1; echo "tests exit $?"; tail -2 $out/t41_tests.txt
