#!/bin/bash
# where do the coarse distance kernel's cycles go?  SQ counter passes over the C3 step (separate rocprofv3 --pmc runs); LIBP=prof XOPT=...
O=$1
R=$PWD
[ "${LIBP:-}" = prof ] && export MVS_LIB_PATH=$R/duckdb-faiss-ext_amd/libmi355faiss_prof.so
cd /tmp && export TMPDIR=/tmp && cd $R
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_LDS SQ_WAVES"; do
  rm -rf $O/pm_c3
  rocprofv3 --pmc $c --output-format csv -d $O/pm_c3 -- python3 bench.py --index IVF4096,Flat --data clustered --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps 2 --warmup 1 ${XOPT:-} > /dev/null 2> $O/pm_c3.err
  f=$(find $O/pm_c3 -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $c"; tail -2 $O/pm_c3.err; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    if "coarse_dist" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        print("%s last=%.6g n=%d  [%s]" % (c, v[-1], len(v), k[:50]))
PY
  rm -rf $O/pm_c3
done 2>&1 | tee $O/coarse_pmc.txt
