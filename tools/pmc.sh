#!/bin/bash
# usage: tools/pmc.sh <outdir> <counters...> -- <python args>   (run on the GPU box; one PMC pass)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d $out -- python3 "$@" > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "mfma" in k or "direct" in k or "ivf" in k or "hnsw" in k or "bf16" in k:
        print(k, {c: f"{v:.4g}" for c, v in d.items()})
PY
