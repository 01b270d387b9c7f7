#!/bin/bash
# round 3, second GPU pass: IVF exact-tie parity, the new bench line (embedded configs + host pointer), C2 split sweep
out=gpurun_out/r3; mkdir -p $out
python3 -m pytest tests/test_ivf_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $out/second_tests.txt 2>&1; tail -15 $out/second_tests.txt
python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c3" > $out/second_c3test.txt 2>&1; tail -5 $out/second_c3test.txt
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/second_bench.json 2> $out/second_bench.err; cut -c1-200 $out/second_bench.json; tail -4 $out/second_bench.err
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r3/second_bench.json"))
print(json.dumps(j.get("host_pointer"), indent=1))
print(json.dumps(j.get("configs"), indent=1)[:6000])
PY
for ns in 24 48 72 96 120 160; do
  python3 bench.py --rows 1000000 --no-cpu-baseline --steps 10 --warmup 2 --opt cl_nsplit=$ns 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('C2 cl_nsplit=$ns', j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['grid'])"
done
for ns in 24 32 48 96 152; do
  python3 bench.py --rows 1250000 --no-cpu-baseline --steps 10 --warmup 2 --opt cl_nsplit=$ns 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); r = j['roofline']
print('N/8 cl_nsplit=$ns', j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['grid'])"
done
