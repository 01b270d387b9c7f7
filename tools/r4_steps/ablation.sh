#!/bin/bash
# The headline scan's ablation instances (profiling library only: results are WRONG): what the MFMA loop costs without rare path /
# fold / staging -- the measurement behind DESIGN.md 3.0's "bare loop = 0.76 of the nominal peak" (VERDICT r3 weak #9).
O=$1
export MVS_LIB_PATH=$GRAFT_REPO_ROOT/duckdb-faiss-ext_amd/libmi355faiss_prof.so
for n in 10000000 1250000; do for abl in 0 1 3 7; do
  python3 bench.py --rows $n --no-cpu-baseline --no-configs --no-host-pointer --steps 6 --warmup 2 --opt cl_abl=$abl --opt cl_defer_count=0 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('N=$n cl_abl=$abl (0: full kernel, 1: no rare path, 3: no fold either, 7: + only the first tile staged) scan_ms=%.3f frac_of_2.5PF=%.4f' % (r['avg_launch_ms'], r['frac']))" | tee -a $O/ablation.txt
done; done
