#!/bin/bash
# C5 line with the row bytes the walk actually moves next to the algorithmic ones; smoke; the HNSW + C-ABI tests
out=gpurun_out/r3p; mkdir -p $out gpurun_out/r3
python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 > $out/c5_hnsw_bench.json 2> $out/c5.err; python3 -c "
import json; j=json.loads(open('$out/c5_hnsw_bench.json').read().strip().splitlines()[-1]); r=j['roofline']
print('C5', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['achieved'], r['frac'], r.get('row_bytes_moved_GBps'), r.get('f32_rows_fetched_frac'), j.get('recall_at_10'), j.get('labels_and_distances_bit_exact_vs_oracle'))"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 600 python3 -m pytest tests/test_hnsw_gpu.py tests/test_boundary_driver_gpu.py -q -m gpu > gpurun_out/r3/t42_tests.txt 2>&1; echo "tests exit $?"; tail -2 gpurun_out/r3/t42_tests.txt
