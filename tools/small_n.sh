#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in 16384 32768 65536 131072; do for nq in 64 2048 10000; do for p in 0 2; do timeout 300 python tools/kbench.py --n $n --nq $nq --opt prefilter=$p --reps 5 2>&1 | grep ms/launch | sed -e 's/opts=.*n=/n=/' | cut -c1-130; done; done; done
