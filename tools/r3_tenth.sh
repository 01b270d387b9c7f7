#!/bin/bash
# round 3, tenth GPU pass: block barrier that does not wait for the slot atomics (A/B: cl_ksplit_opt=16 = plain barrier), IVF IP on list means
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_collect_gpu.py tests/test_prefilter_gpu.py tests/test_flat_gpu.py -x -q -m gpu > $out/tenth_tests.txt 2>&1; tail -4 $out/tenth_tests.txt
for rep in 1 2; do for opt in 0 16; do for rows in 10000000 1250000; do
  python3 bench.py --rows $rows --no-cpu-baseline --steps 10 --warmup 3 --parity-device 512 --opt cl_ksplit_opt=$opt 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('opt=$opt (16: plain barrier) N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done; done
python3 bench.py --chunk 2048 --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('chunk2048', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'])"
timeout 600 python3 tools/dbg_ivf_ip.py 2>&1 | grep -v amdgpu.ids
timeout 900 python3 -m pytest tests/test_ivf_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "ivf" > $out/tenth_ivf_tests.txt 2>&1; tail -4 $out/tenth_ivf_tests.txt
