#!/bin/bash
# the abort at exit with RCCL loaded before torch: RTLD_LOCAL
out=gpurun_out/r3; mkdir -p $out
timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py -q -m gpu -k "merge or rccl" > $out/t23_b.txt 2>&1; echo "rccl first, then torch: exit $?"; tail -2 $out/t23_b.txt
timeout 600 python3 -m pytest tests/test_merge_device_gpu.py tests/test_sharded_inprocess_gpu.py -q -m gpu > $out/t23_a.txt 2>&1; echo "torch first, then all sharded tests: exit $?"; tail -2 $out/t23_a.txt
