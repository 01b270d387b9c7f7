#!/bin/bash
# the N > 1 flow of bench.py on a 1-GPU box: two (four) ranks on device 0 over gloo -- a functional check of the distributed path, no measurement
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
export MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --rows 2000000 2>&1 | grep '^{' | head -c 900; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 --rows 2000000 --index IVF1024,Flat --data clustered 2>&1 | grep '^{' | head -c 900; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 2 --warmup 1 --rows 2000000 --nq 512 --k 300 2>&1 | grep '^{' | head -c 900; echo
