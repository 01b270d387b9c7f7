#!/bin/bash
# C3's scan with its rare path switched off (profiling library only: results are WRONG)
O=$1
export MVS_LIB_PATH=$GRAFT_REPO_ROOT/duckdb-faiss-ext_amd/libmi355faiss_prof.so
for abl in 0 1; do
  python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 2 --opt ivf_cl_abl=$abl 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('C3 ivf_cl_abl=$abl (1: no rare path, results wrong) step_ms=%.3f scan_ms=%.4f' % (j['ms_per_step'], r['avg_launch_ms']))" | tee -a $O/ivf_abl.txt
done
