#!/bin/bash
# One parametrised GPU-box pass (round 5; steps under tools/r5_steps, then tools/r4_steps): run as
#   gpurun --timeout T -- 'bash tools/r5_pass.sh <step> [<step> ...]'
# every step writes under gpurun_out/r5/ ; summaries worth judging are copied into profiles/ by hand afterwards.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5; mkdir -p $O
for step in "$@"; do
  echo "=== $step ==="
  case "$step" in
    selflaunch2)   # VERDICT r3 #1b: python3 bench.py --gpus 2 with no launcher, two gloo ranks sharing the device
      MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --rows 2000000 > $O/selflaunch2.json 2> $O/selflaunch2.err; echo "rc=$?"; tail -c 1500 $O/selflaunch2.json; tail -5 $O/selflaunch2.err ;;
    selflaunch4)
      MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo timeout 1200 python3 bench.py --gpus 4 --steps 3 --warmup 1 --rows 2000000 > $O/selflaunch4.json 2> $O/selflaunch4.err; echo "rc=$?"; tail -c 2500 $O/selflaunch4.json; tail -5 $O/selflaunch4.err ;;
    bench)         # the driver's command
      timeout 1500 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "rc=$?"; tail -c 6000 $O/bench_default.json; tail -3 $O/bench_default.err ;;
    bench_quick)
      timeout 900 python3 bench.py --no-configs --no-host-pointer > $O/bench_quick.json 2> $O/bench_quick.err; echo "rc=$?"; tail -c 3000 $O/bench_quick.json; tail -3 $O/bench_quick.err ;;
    tests)         timeout 3000 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15 ;;
    tests_ivf)     timeout 2400 python3 -m pytest tests/test_ivf_gpu.py tests/test_fuzz_gpu.py tests/test_sharded_inprocess_gpu.py -m gpu -x -q 2>&1 | tail -15 ;;
    tests_fast)    timeout 2400 python3 -m pytest tests -m gpu -x -q --deselect tests/test_configs_gpu.py 2>&1 | tail -15 ;;
    tests_configs) timeout 2400 python3 -m pytest tests/test_configs_gpu.py -m gpu -x -q -s 2>&1 | tail -25 ;;
    smoke)         timeout 900 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ;;
    *)             if [ -f "tools/r5_steps/$step.sh" ]; then bash "tools/r5_steps/$step.sh" "$O"; elif [ -f "tools/r4_steps/$step.sh" ]; then bash "tools/r4_steps/$step.sh" "$O"; else echo "unknown step $step"; fi ;;
  esac
done
