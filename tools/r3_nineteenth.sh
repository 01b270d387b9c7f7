#!/bin/bash
# round 3, nineteenth GPU pass: bench.py's N > 1 flow with query groups (4 gloo ranks sharing the one GPU: 2 x 2, then 1 x 4 and 4 x 1),
# the new selector / k = 24 / d = 48 cases of the coarse-filter tests
out=gpurun_out/r3; mkdir -p $out
cd $GRAFT_REPO_ROOT
for g in 0 1 4; do
MVS_BENCH_SHARED_GPU=1 MVS_BENCH_BACKEND=gloo timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 2961$g bench.py --gpus 4 --steps 2 --warmup 1 --rows 2000000 --query-groups $g 2> $out/nineteenth_g$g.err | grep -o '{"metric.*' | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); print('query-groups=$g', j['n_gpus'], j['value'], j['ms_per_step'], j['config'].get('row_shards'), j['config'].get('replicas'), {k: v for k, v in j.items() if 'merged' in k or 'oracle' in k})"
tail -2 $out/nineteenth_g$g.err | grep -v amdgpu.ids | cut -c1-300
done
timeout 1500 python3 -m pytest tests/test_collect_gpu.py tests/test_collect_wide_gpu.py tests/test_prefilter_gpu.py tests/test_sharded_inprocess_gpu.py tests/test_merge_device_gpu.py -q -m gpu > $out/nineteenth_tests.txt 2>&1; tail -4 $out/nineteenth_tests.txt
