#!/bin/bash
# round 3, eleventh GPU pass: IVF pre-pass over the main pass's items (A/B), HNSW 32 bf16 rows in flight (two builds), harness at full size
out=gpurun_out/r3; mkdir -p $out
for pp in 128 256 64 0; do for m in L2 IP; do
  python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 10 --warmup 2 --parity-device 1024 --opt ivf_cl_prepass=$pp 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C3 $m ivf_cl_prepass=$pp', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'], j.get('recall_at_10'))"
done; done
for rep in 1 2; do for lib in libmi355faiss.so libmi355faiss_prev.so; do
  MVS_LIB_PATH=$PWD/duckdb-faiss-ext_amd/$lib python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 $lib', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"
done; done
timeout 900 python3 -m pytest tests/test_ivf_gpu.py tests/test_hnsw_gpu.py -x -q -m gpu > $out/eleventh_tests.txt 2>&1; tail -4 $out/eleventh_tests.txt
timeout 1500 python3 tools/harness_bench.py --n 8841823 --reps 3 > $out/harness_shapes_full.txt 2>&1; grep -v amdgpu.ids $out/harness_shapes_full.txt
