#!/bin/bash
timeout 1500 python3 -m pytest tests/test_flat_shadow_gpu.py -m gpu -x -q > $1/shadow_tests.log 2>&1; head -60 $1/shadow_tests.log; tail -5 $1/shadow_tests.log
if [ -n "${KINDS:-}" ]; then KINDS="$KINDS" timeout 1500 python3 tools/collect_sensitivity.py 2>&1 | tail -9 | tee $1/shadow_sens.txt; fi
