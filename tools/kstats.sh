#!/bin/bash
# per-kernel times of one bench shape under rocprofv3 --kernel-trace --stats; per-step figures from the step count of the run
#   ROWS=1250000 OPT="a=1,b=2" ARGS="--index IVF4096,Flat --data clustered" TAG=x STEPS=10 WARMUP=2 bash tools/kstats.sh <outdir>
O=$1
extra=""; [ -n "${OPT:-}" ] && extra="--opt ${OPT//,/ --opt }"
S=${STEPS:-10}; W=${WARMUP:-2}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/trace_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_${TAG} -- python3 bench.py --rows ${ROWS:-1250000} --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps $S --warmup $W $extra ${ARGS:-} > $O/kstats_${TAG}.json 2> $O/kstats_${TAG}.err
f=$(find $O/trace_${TAG} -name "*kernel_stats.csv" | head -1)
t=$(find $O/trace_${TAG} -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$t" $S $W <<'PY' | tee $O/kstats_${TAG}.txt
import csv, sys, collections
S, W = int(sys.argv[3]), int(sys.argv[4])
# the searches of the run: W warm-up + S timed + 3 of the state-sensitivity leg (bench.py) + parity searches if asked for
rows = list(csv.DictReader(open(sys.argv[1])))
# per-search launch list from the trace: the LAST S + W + 3 repetitions are too entangled to cut by name, so count launches
# between consecutive occurrences of the dominant kernel instead
tr = list(csv.DictReader(open(sys.argv[2])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in tr]
import os
pat = os.environ.get("DOM", "collect_kernel hnsw_search_kernel flat_mfma_resident").split()
cands = [r for r in rows if any(p_ in r["Name"] for p_ in pat[:2])] or [r for r in rows if any(p_ in r["Name"] for p_ in pat)]
dom = max(cands or rows, key=lambda r: float(r["TotalDurationNs"]))["Name"]
idx = [i for i, nm in enumerate(names) if nm == dom]
nsearch = S + W
lo = S
tot = 0.0
for r in rows:
    c = int(r["Calls"])
    if c < lo:
        continue
    tot += float(r["TotalDurationNs"]) / 1e6
    print("%-64s calls %5s avg_us %9.1f total_ms %9.3f" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("sum of kernels with >= %d calls: %.3f ms" % (lo, tot))
# one steady search, cut out of the trace: from the first kernel after the previous search's last dominant launch ... here simply
# the launches between the dominant kernel's (last - 1)-th and last occurrence groups
if len(idx) >= 12:
    # dominant launches per search (IVF: pre-pass + main = 2; Flat: 1)
    per = 2 if "ivf_bf16" in dom else 1
    e = idx[-1]                     # last dominant launch of the last TIMED search (bench.py skips its extra searches under the profiler)
    b = idx[-1 - per]               # ... of the search before it
    seg = tr[b + 1 : e + 1]
    # rotate: a search starts with its first kernel after the previous search's tail; print in launch order
    agg = collections.OrderedDict()
    for r in seg:
        k = r["Kernel_Name"][:64]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if k not in agg:
            agg[k] = [0, 0.0]
        agg[k][0] += 1
        agg[k][1] += d
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    print("--- one steady search (launch order, tail of the previous one first): %d launches, %.1f us of kernels in a %.1f us window" % (len(seg), sum(v[1] for v in agg.values()), (t1 - t0) / 1e3))
    for k, v in agg.items():
        print("   %-64s x%d %8.1f us" % (k, v[0], v[1]))
PY
rm -rf $O/trace_${TAG}
