#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_ivfk
KS="100 100 100" rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ivfk -- python3 $GRAFT_REPO_ROOT/tools/ivf_k_bench.py > /dev/null 2>&1
f=$(find $O/trace_ivfk -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if int(r["Calls"]) >= 15]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:22]:
    print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/trace_ivfk
