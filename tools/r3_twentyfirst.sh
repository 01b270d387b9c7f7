#!/bin/bash
# which tests leave the process unable to exit cleanly?
out=gpurun_out/r3; mkdir -p $out
timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py -q -m gpu -k "not rccl" > $out/t21_a.txt 2>&1; echo "sharded without rccl: exit $?"; tail -2 $out/t21_a.txt
timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py -q -m gpu -k "rccl" > $out/t21_b.txt 2>&1; echo "rccl only: exit $?"; tail -2 $out/t21_b.txt
MALLOC_CHECK_=3 timeout 600 python3 -m pytest tests/test_sharded_inprocess_gpu.py -q -m gpu -k "rccl" > $out/t21_c.txt 2>&1; echo "rccl only, MALLOC_CHECK_=3: exit $?"; tail -3 $out/t21_c.txt
timeout 600 python3 -m pytest tests/test_merge_device_gpu.py -q -m gpu > $out/t21_d.txt 2>&1; echo "merge_device: exit $?"; tail -2 $out/t21_d.txt
