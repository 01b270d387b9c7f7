#!/bin/bash
# HNSW32 build on uniform rows: kernel trace + one SQ counter pass (say what binds before changing it)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf $O/trace_hb; rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_hb -- python3 tools/hnsw_build_probe.py 60000 768 > $O/r6_hnsw_build.txt 2> $O/r6_hnsw_build.err
f=$(find $O/trace_hb -name "*kernel_stats.csv" | head -1)
python3 - "$f" >> $O/r6_hnsw_build.txt <<'PY'
import csv, sys
for r in sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print("%-70s calls %5s avg_us %10.1f total_ms %10.2f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/trace_hb $O/pmc_hb
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_hb -- python3 tools/hnsw_build_probe.py 60000 768 > /dev/null 2>&1
f=$(find $O/pmc_hb -name "*counter_collection.csv" | head -1)
python3 - "$f" >> $O/r6_hnsw_build.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "hnsw_build" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k, {c: "%.4g" % v for c, v in sorted(d.items())})
PY
rm -rf $O/pmc_hb
cat $O/r6_hnsw_build.txt | grep -v amdgpu.ids | cut -c1-300
