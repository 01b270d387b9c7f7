#!/bin/bash
# why is the scan slower with pruned probe lists?  per-launch durations + instruction / fetch counters, nprobe = 4, with / without
O=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in 0 1; do
  rm -rf $O/tr_p$p
  rocprofv3 --kernel-trace --output-format csv -d $O/tr_p$p -- python3 bench.py --index IVF4096,Flat --data clustered --nprobe 4 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps 3 --warmup 1 --opt ivf_probe_prune=$p > /dev/null 2> $O/tr_p$p.err
  t=$(find $O/tr_p$p -name "*kernel_trace.csv" | head -1)
  python3 - "$t" $p <<'PY'
import csv, sys
tr = list(csv.DictReader(open(sys.argv[1])))
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in tr:
    if "ivf_bf16_collect" in r["Kernel_Name"] or "pack2" in r["Kernel_Name"] or "exact_bucket" in r["Kernel_Name"]:
        out.append("%s %.1f" % (r["Kernel_Name"][10:34], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print("prune=%s" % sys.argv[2], " | ".join(out[-12:]))
PY
  rm -rf $O/tr_p$p
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES" "FETCH_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum"; do
    rm -rf $O/pm_p$p
    rocprofv3 --pmc $c --output-format csv -d $O/pm_p$p -- python3 bench.py --index IVF4096,Flat --data clustered --nprobe 4 --no-cpu-baseline --no-configs --no-host-pointer --no-ingest --steps 2 --warmup 1 --opt ivf_probe_prune=$p > /dev/null 2> $O/pm_p$p.err
    f=$(find $O/pm_p$p -name "*counter_collection.csv" | head -1)
    [ -z "$f" ] && { echo "no counters for: $c"; tail -3 $O/pm_p$p.err; continue; }
    python3 - "$f" $p <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:60]
    if "ivf_bf16_collect" in k:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        print("prune=%s %s last=%.6g prev=%.6g n=%d" % (sys.argv[2], c, v[-1], v[-2] if len(v) > 1 else -1, len(v)))
PY
    rm -rf $O/pm_p$p
  done
done 2>&1 | tee $O/prune3.txt
