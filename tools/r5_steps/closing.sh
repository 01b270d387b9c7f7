#!/bin/bash
# end of round 5 on the final tree: kernel listings, the default bench line
O=$1
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3_final STEPS=10 WARMUP=2 DOM="ivf_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O > /dev/null
ROWS=1250000 TAG=n8_final STEPS=10 WARMUP=2 DOM="flat_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O > /dev/null
ROWS=10000000 TAG=h_final STEPS=10 WARMUP=2 DOM="flat_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O > /dev/null
timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
