#!/bin/bash
# where do the big kernel's wave cycles go?  SQ counter passes over the C4 shard (separate rocprofv3 --pmc runs)
O=$1
R=$PWD
cd /tmp && export TMPDIR=/tmp && cd $R
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM_RD" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  rm -rf $O/pm_c4
  rocprofv3 --pmc $c --output-format csv -d $O/pm_c4 -- python3 bench.py --rows ${C4ROWS:-4000000} --d 768 --metric IP --normalize --data clustered --sigma 1.0 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps 2 --warmup 1 > /dev/null 2> $O/pm_c4.err
  f=$(find $O/pm_c4 -name "*counter_collection.csv" | head -1)
  [ -z "$f" ] && { echo "no counters for: $c"; tail -2 $O/pm_c4.err; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]
    if "big_kernel" in k and "true" in k.split("<")[1].split(",")[3]:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        print("%s last=%.6g n=%d  [%s]" % (c, v[-1], len(v), k[20:62]))
PY
  rm -rf $O/pm_c4
done 2>&1 | tee $O/c4pmc.txt
