#!/bin/bash
# round 3, first measurement after the bound fix: parity of the coarse-filter paths + what the doubled E costs
out=gpurun_out/r3; mkdir -p $out
python3 -m pytest tests/test_collect_gpu.py tests/test_collect_wide_gpu.py tests/test_prefilter_gpu.py tests/test_ivf_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu > $out/first_tests.txt 2>&1; tail -3 $out/first_tests.txt
python3 bench.py --no-cpu-baseline > $out/first_headline.json 2>$out/first_headline.err; cut -c1-400 $out/first_headline.json
python3 bench.py --rows 1000000 --no-cpu-baseline > $out/first_c2.json 2>/dev/null; cut -c1-300 $out/first_c2.json
python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline > $out/first_c3.json 2>/dev/null; cut -c1-300 $out/first_c3.json
python3 bench.py --rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline > $out/first_c4.json 2>/dev/null; cut -c1-300 $out/first_c4.json
grep -o '"candidates_rescored_per_query": [0-9.]*' $out/first_*.json
