#!/bin/bash
# HNSW: the wave reduction's last two steps on the vector ALU (v_permlane16/32_swap) instead of ds_bpermute -- same-box A/B + tests
out=gpurun_out/r3; mkdir -p $out
L=$PWD/duckdb-faiss-ext_amd
for rep in 1 2; do for lib in libmi355faiss_prev.so libmi355faiss.so; do
  MVS_LIB_PATH=$L/$lib python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 $lib', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"
done; done
timeout 1200 python3 -m pytest tests/test_hnsw_gpu.py tests/test_fuzz_gpu.py -q -m gpu -k "hnsw" > $out/t29_tests.txt 2>&1; echo "hnsw tests exit $?"; tail -2 $out/t29_tests.txt
