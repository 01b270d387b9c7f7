#!/bin/bash
# the abort at exit: which order / what cures it
out=gpurun_out/r3; mkdir -p $out
timeout 600 python3 -m pytest tests/test_merge_device_gpu.py tests/test_sharded_inprocess_gpu.py -q -m gpu -k "merge or rccl" > $out/t22_a.txt 2>&1; echo "torch first, then rccl (the suite's order): exit $?"; tail -2 $out/t22_a.txt
timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py -q -m gpu -k "merge or rccl" > $out/t22_b.txt 2>&1; echo "rccl first, then torch: exit $?"; tail -2 $out/t22_b.txt
MVS_RCCL_KEEP_COMMS=1 timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py -q -m gpu -k "merge or rccl" > $out/t22_c.txt 2>&1; echo "rccl first, then torch, communicators not destroyed: exit $?"; tail -2 $out/t22_c.txt
timeout 600 python3 -c "
import sys; sys.path.insert(0,'duckdb-faiss-ext_amd/pyhost'); sys.path.insert(0,'.')
import torch
import numpy as np, mi355_faiss as mf
ix = mf.index_factory(40, 'Flat', mf.METRIC_L2); ix.add(np.random.rand(12000,40).astype('float32')); ix.shard_to_gpus([0]); ix.set_option('shard_exchange', 1)
print(ix.search(np.random.rand(5,40).astype('float32'), 3)[1][:2]); del ix
t = torch.ones(4, device='cuda'); print(float(t.sum()))
" > $out/t22_d.txt 2>&1; echo "script: import torch, rccl search, torch tensor: exit $?"; tail -3 $out/t22_d.txt
