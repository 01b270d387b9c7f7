#!/bin/bash
# 2 gloo ranks sharing the GPU: IVF row shards with the coarse filter on / off (flow check)
cd $GRAFT_REPO_ROOT
for o in -1 0; do
MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$((o+2)) bench.py --gpus 2 --steps 2 --warmup 1 --rows 2000000 --index IVF256,Flat --data clustered --no-cpu-baseline --opt ivf_collect=$o 2>/dev/null | grep -o '{"metric.*' | cut -c1-200
done
python3 bench.py --steps 2 --warmup 1 --rows 1000000 --index IVF256,Flat --data clustered --no-cpu-baseline 2>/dev/null | cut -c1-200
