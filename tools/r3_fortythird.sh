#!/bin/bash
mkdir -p gpurun_out/r3
timeout 900 python3 -m pytest tests/test_hnsw_gpu.py tests/test_boundary_driver_gpu.py tests/test_index_io_gpu.py -q -m gpu > gpurun_out/r3/t43_tests.txt 2>&1; echo "tests exit $?"; tail -2 gpurun_out/r3/t43_tests.txt
