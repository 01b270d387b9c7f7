#!/bin/bash
# usage: tools/pmc_sq.sh <tag> [bench.py args]   one SQ counter pass of the bench (run on the GPU box)
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $out/pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-host-pointer "$@" > $out/bench.json 2> $out/err.txt
f=$(find $out/pmc -name "*counter_collection.csv" | head -1)
python3 - "$f" > $out/pmc_sq.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "bf16" in k or "flat_mfma" in k:
        print(k)
        for c, v in sorted(d.items()):
            print(f"   {c}: dispatches={len(v)} mean={sum(v)/len(v):.6g}")
PY
cat $out/pmc_sq.txt; rm -rf $out/pmc
