#!/bin/bash
# round 3, fifth GPU pass: big kernel v2 (mid-tile barrier, spread LDS-DMA), HNSW bf16 first look, overflow policy, N/8 step profile
out=gpurun_out/r3; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_collect_wide_gpu.py -x -q -m gpu > $out/fifth_wide_tests.txt 2>&1; tail -6 $out/fifth_wide_tests.txt
timeout 300 python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --parity-device 256 > $out/fifth_c4.json 2>$out/fifth_c4.err
python3 -c "
import json; j=json.load(open('$out/fifth_c4.json')); r=j['roofline']
print('C4 big v2', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], r['grid'], r['candidates_rescored_per_query'], j.get('parity_device'))" || tail -3 $out/fifth_c4.err
timeout 300 python3 bench.py --rows 2000000 --d 1024 --no-cpu-baseline --parity-device 256 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=1024 N=2M big v2', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], j.get('parity_device'))"
timeout 300 python3 bench.py --rows 2000000 --d 768 --no-cpu-baseline --parity-device 256 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('d=768 L2 N=2M big v2', j['value'], j['ms_per_step'], r['kernel'], r['avg_launch_ms'], r['frac'], j.get('parity_device'))"
timeout 900 python3 -m pytest tests/test_hnsw_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "hnsw" > $out/fifth_hnsw_tests.txt 2>&1; tail -6 $out/fifth_hnsw_tests.txt
timeout 600 python3 -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c5" > $out/fifth_c5_test.txt 2>&1; tail -4 $out/fifth_c5_test.txt
for bf in 1 0; do
  MVS_HNSW_STATS=1 timeout 400 python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --cpu-seconds 2 --opt hnsw_bf16=$bf > $out/fifth_c5_bf$bf.json 2> $out/fifth_c5_bf$bf.err
  python3 -c "
import json; j=json.load(open('$out/fifth_c5_bf$bf.json')); r=j['roofline']
print('C5 hnsw_bf16=$bf', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"; grep "\[hnsw\]" $out/fifth_c5_bf$bf.err | tail -1
done
timeout 900 python3 -m pytest tests/test_collect_gpu.py -x -q -m gpu -k "overflow or duplicates" > $out/fifth_collect_tests.txt 2>&1; tail -8 $out/fifth_collect_tests.txt
timeout 600 python3 -m pytest tests/test_index_io_gpu.py -q -m gpu -k "cross" > $out/fifth_io_tests.txt 2>&1; tail -5 $out/fifth_io_tests.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_n8 -- python3 bench.py --rows 1250000 --steps 10 --warmup 2 --no-cpu-baseline > $out/fifth_n8_trace_bench.json 2> $out/fifth_n8_trace.err
f=$(find $out/trace_n8 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python3 tools/kstats_print.py "$f" 2>/dev/null | head -24 > $out/fifth_n8_step_kernels.txt; cat $out/fifth_n8_step_kernels.txt; rm -rf $out/trace_n8
KINDS="clustered all_dup" timeout 600 python3 tools/collect_sensitivity.py > $out/fifth_sensitivity.txt 2>&1; grep -v amdgpu.ids $out/fifth_sensitivity.txt
