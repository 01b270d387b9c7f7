#!/bin/bash
# round 3, ninth GPU pass: same-box A/B of the candidate-queue change (two builds), IVF inner-product regression hunt
out=gpurun_out/r3; mkdir -p $out
for rep in 1 2; do for lib in libmi355faiss.so libmi355faiss_oldq.so; do for rows in 10000000 1250000; do
  MVS_LIB_PATH=$PWD/duckdb-faiss-ext_amd/$lib python3 bench.py --rows $rows --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'])"
done; done; done
for lib in libmi355faiss.so libmi355faiss_oldq.so; do
MVS_LIB_PATH=$PWD/duckdb-faiss-ext_amd/$lib python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib C3', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'])"
done
timeout 600 python3 tools/dbg_ivf_ip.py 2>&1 | grep -v amdgpu.ids
METRIC=L2 timeout 600 python3 tools/dbg_ivf_ip.py 2>&1 | grep -v amdgpu.ids
