#!/bin/bash
# counter passes of the other bench workloads (C3: TAG=c3 ARGS=... pmc_all on its own)
O=$1
TAG=h ARGS="" KERNELS="flat_bf16_collect_kernel" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/h_pmc.txt
TAG=n8 ARGS="--rows 1250000" KERNELS="flat_bf16_collect_kernel" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/n8_pmc.txt
TAG=c4 ARGS="--rows 12500000 --d 768 --metric IP --normalize --data clustered --sigma 1.0" KERNELS="wide_kernel big_kernel" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/c4_pmc.txt
TAG=c5 ARGS="--index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0" KERNELS="hnsw_search" bash tools/r5_steps/pmc_all.sh $O > /dev/null; cat $O/c5_pmc.txt
