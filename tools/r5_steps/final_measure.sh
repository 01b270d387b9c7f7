#!/bin/bash
# round-5 closing measurements on one box: counters for the kernels that changed (C3 scan + exact stage, C4 big kernel), kernel listings,
# data sensitivity, the default bench line
O=$1
TAG=c3 ARGS="--index IVF4096,Flat --data clustered" KERNELS="ivf_bf16 bucket_exact bucket_scatter" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/c3_pmc.txt
TAG=c4 ARGS="--rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0" KERNELS="big_kernel" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/c4_pmc.txt
ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered" TAG=c3_final STEPS=10 WARMUP=2 DOM="ivf_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O > /dev/null
ROWS=1250000 TAG=n8_final STEPS=10 WARMUP=2 DOM="flat_bf16_collect_kernel" bash tools/r5_steps/kstats.sh $O > /dev/null
SHAPES="10000000 1250000 1000000" bash tools/r4_steps/shapes.sh $O 2>&1 | tail -3
bash tools/r5_steps/sens.sh $O
timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 1500 $O/bench_default.json
