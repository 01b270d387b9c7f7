#!/bin/bash
# d <= 128 scan: look at the candidate queue every 8 staged blocks instead of every 2 -- same-box A/B of two builds
L=$PWD/duckdb-faiss-ext_amd
for rep in 1 2; do for lib in libmi355faiss_prev.so libmi355faiss.so; do for rows in 10000000 1250000; do
  MVS_LIB_PATH=$L/$lib python3 bench.py --rows $rows --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --parity-device 512 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done; done
