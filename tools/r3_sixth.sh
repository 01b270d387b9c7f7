#!/bin/bash
# round 3, sixth GPU pass: HNSW bf16 first look, overflow policy, io cross tests, N/8 step profile, sensitivity; LAST: big kernel modes
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_hnsw_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "hnsw" > $out/sixth_hnsw_tests.txt 2>&1; tail -6 $out/sixth_hnsw_tests.txt
timeout 600 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c5" > $out/sixth_c5_test.txt 2>&1; tail -4 $out/sixth_c5_test.txt
for bf in 1 0; do
  MVS_HNSW_STATS=1 timeout 400 python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 --opt hnsw_bf16=$bf > $out/sixth_c5_bf$bf.json 2> $out/sixth_c5_bf$bf.err
  python3 -c "
import json; j=json.load(open('$out/sixth_c5_bf$bf.json')); r=j['roofline']
print('C5 hnsw_bf16=$bf', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"; grep "\[hnsw\]" $out/sixth_c5_bf$bf.err | tail -1
done
timeout 900 python3 -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "overflow or duplicates" > $out/sixth_collect_tests.txt 2>&1; tail -8 $out/sixth_collect_tests.txt
timeout 600 python3 -m pytest tests/test_index_io_gpu.py -q -m gpu -k "cross" > $out/sixth_io_tests.txt 2>&1; tail -5 $out/sixth_io_tests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_n8 -- python3 bench.py --rows 1250000 --steps 10 --warmup 2 --no-cpu-baseline > $out/sixth_n8_trace_bench.json 2> $out/sixth_n8_trace.err
f=$(find $out/trace_n8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kstats_print.py "$f" 2>/dev/null | head -24 > $out/sixth_n8_step_kernels.txt; cat $out/sixth_n8_step_kernels.txt; rm -rf $out/trace_n8
for sr in 65536 131072; do for rows in 10000000 1250000; do
  python3 bench.py --rows $rows --no-cpu-baseline --steps 10 --warmup 2 --opt cl_seed_rows=$sr 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('N=$rows cl_seed_rows=$sr', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], r['candidates_rescored_per_query'])"
done; done
KINDS="clustered all_dup" timeout 600 python3 tools/collect_sensitivity.py > $out/sixth_sensitivity.txt 2>&1; grep -v amdgpu.ids $out/sixth_sensitivity.txt
for mode in 0 2 1 3; do timeout 300 python3 tools/big_mode_check.py $mode 768 2>&1 | grep -v amdgpu.ids | tail -4; done
