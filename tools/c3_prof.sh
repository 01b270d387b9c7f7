#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_prof -o c3 -- python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --steps 5 --warmup 1 > gpurun_out/r2_c3_prof.json 2> gpurun_out/r2_c3_prof.err
f=$(find gpurun_out/c3_prof -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY' | tee gpurun_out/r2_c3_kstats.txt
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:28]:
    print(r["Name"][:90].ljust(90), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%8.1f ms total" % (float(r["TotalDurationNs"])/1e6))
PY
