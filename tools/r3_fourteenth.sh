#!/bin/bash
# round 3, fourteenth GPU pass: 16 < k <= 32 on the coarse filter (tests + rate), step times of the (queries, rows) shapes an 8-GPU
# run can hand one GPU, bound-refresh cadence at short scans
out=gpurun_out/r3; mkdir -p $out
timeout 1200 python3 -m pytest tests/test_collect_gpu.py -x -q -m gpu > $out/fourteenth_tests.txt 2>&1; tail -4 $out/fourteenth_tests.txt
for k in 10 16 20 32; do for o in 1 0; do
  python3 bench.py --k $k --no-cpu-baseline --no-configs --no-host-pointer --steps 5 --warmup 2 --parity-device 256 --opt cl_k32=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('k=$k cl_k32=$o', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r.get('candidates_rescored_per_query'), j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done
for shape in "10000 1250000" "5000 2500000" "2500 5000000" "1250 10000000" "10000 2500000" "5000 5000000" "10000 5000000" "5000 10000000"; do set -- $shape
  python3 bench.py --nq $1 --rows $2 --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('shape nq=$1 N=$2', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'))"
done
for o in 0 4; do for rows in 1250000 2500000; do
  python3 bench.py --rows $rows --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --opt cl_ksplit_opt=$o 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('cadence opt=$o N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r.get('candidates_rescored_per_query'))"
done; done
