#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=$PWD/gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_ATOMIC_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf $O/pmc_fc
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_fc -- python3 $GRAFT_REPO_ROOT/tools/first_call_probe.py > /dev/null 2>&1
  f=$(find $O/pmc_fc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "rows_to_bf16" in k or "colsum" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items():
        print(c, k, len(v), "sum=%.6g" % sum(v))
PY
done
rm -rf $O/pmc_fc
