#!/bin/bash
# round 3, thirteenth GPU pass: exact re-scoring with the whole query in registers -- same-box A/B of two builds, then the tests
out=gpurun_out/r3; mkdir -p $out
L=$PWD/duckdb-faiss-ext_amd
for rep in 1 2; do for lib in libmi355faiss_expect.so libmi355faiss.so; do for rows in 10000000 1250000; do
  MVS_LIB_PATH=$L/$lib python3 bench.py --rows $rows --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 3 --parity-device 512 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib N=$rows', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
done; done; done
for lib in libmi355faiss_expect.so libmi355faiss.so; do
  MVS_LIB_PATH=$L/$lib python3 bench.py --chunk 2048 --no-cpu-baseline --no-configs --no-host-pointer --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib chunk2048', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'])"
  for m in L2 IP; do
  MVS_LIB_PATH=$L/$lib python3 bench.py --index IVF4096,Flat --data clustered --metric $m --no-cpu-baseline --steps 10 --warmup 2 --parity-device 1024 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('$lib C3 $m', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], j['parity_device']['labels_equal'], j['parity_device']['distances_bit_equal'])"
  done
done
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t13_c2 -- python3 bench.py --rows 1250000 --no-cpu-baseline --no-configs --no-host-pointer --steps 5 --warmup 2 > /dev/null 2>&1
f=$(find $out/t13_c2 -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-150; cp "$f" $out/thirteenth_n1250k_kernel_stats.csv; rm -rf $out/t13_c2
timeout 1500 python3 -m pytest tests/test_collect_gpu.py tests/test_ivf_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py tests/test_flat_gpu.py tests/test_hnsw_gpu.py -x -q -m gpu > $out/thirteenth_tests.txt 2>&1; tail -4 $out/thirteenth_tests.txt
