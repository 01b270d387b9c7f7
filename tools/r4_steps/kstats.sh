#!/bin/bash
# per-kernel times of one bench shape under rocprofv3 --kernel-trace --stats: ROWS=1250000 OPT="cl_defer_count=0" TAG=x
O=$1
extra=""; [ -n "${OPT:-}" ] && extra="--opt ${OPT//,/ --opt }"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/trace_${TAG}
if [ -n "${PYCMD:-}" ]; then  # any python command line instead of bench.py: PYCMD="tools/collect_sensitivity.py"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${TAG} -- python3 $PYCMD > $O/kstats_${TAG}.json 2> $O/kstats_${TAG}.err
else
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${TAG} -- python3 bench.py --rows ${ROWS:-1250000} --no-cpu-baseline --no-configs --no-host-pointer --steps 10 --warmup 2 $extra ${ARGS:-} > $O/kstats_${TAG}.json 2> $O/kstats_${TAG}.err
fi
f=$(find $O/trace_${TAG} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee $O/kstats_${TAG}.txt
import csv, sys
import os
rows = list(csv.DictReader(open(sys.argv[1])))
lo = int(os.environ.get("MINCALLS", "10"))
tot = 0.0
for r in rows:
    if int(r["Calls"]) < lo:
        continue
    tot += float(r["TotalDurationNs"]) / 1e6
    print("%-64s calls %5s avg_us %9.1f total_ms %9.3f per_step_us %8.1f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e3 / 12))
print("sum of kernels with >= %d calls: %.3f ms = %.3f ms per step (10 timed + 2 warm-up steps)" % (lo, tot, tot / 12))
PY
rm -rf $O/trace_${TAG}
