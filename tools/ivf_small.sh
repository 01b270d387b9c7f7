#!/bin/bash
# IVF C3 index, small batches: coarse filter forced (ivf_collect=1) vs scanner (0)
cd $GRAFT_REPO_ROOT
for nq in 1 16 64; do for o in 1 0; do python3 bench.py --index IVF4096,Flat --data clustered --nq $nq --no-cpu-baseline --steps 20 --warmup 3 --opt ivf_collect=$o 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read()); print('nq=$nq ivf_collect=$o', round(j['value']), 'QPS', j['ms_per_step'], 'ms/step', j['roofline']['kernel'])"; done; done
