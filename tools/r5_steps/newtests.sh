#!/bin/bash
timeout 1200 python3 -m pytest tests/test_options_threads_gpu.py tests/test_multi_device_gpu.py tests/test_collect_gpu.py tests/test_cabi_cpu.py -m "gpu or not gpu" -x -q 2>&1 | tail -8
