#!/bin/bash
# round 3, twentieth GPU pass: the process must exit cleanly after the sharded / RCCL tests (Index.__del__ at interpreter shutdown)
out=gpurun_out/r3; mkdir -p $out
timeout 900 python3 -m pytest tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py tests/test_prefilter_gpu.py -q -m gpu > $out/twentieth_tests.txt 2>&1; echo "pytest exit code $?"; tail -3 $out/twentieth_tests.txt
timeout 600 python3 tools/shard_overhead.py > $out/twentieth_shard.txt 2>&1; echo "shard_overhead exit code $?"; tail -2 $out/twentieth_shard.txt
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/twentieth_smoke.txt 2>&1; echo "smoke exit code $?"; tail -1 $out/twentieth_smoke.txt
