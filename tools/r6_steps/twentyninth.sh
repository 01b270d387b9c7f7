#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_bigk
D=128 N=10000000 NQ=2048 KS="1000" METRICS="IP" rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bigk -- python3 $GRAFT_REPO_ROOT/tools/wide_k_bench.py > /dev/null 2>&1
f=$(find $O/trace_bigk -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-70s calls %5s avg_us %10.1f total_ms %9.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/trace_bigk
