#!/bin/bash
# Counter evidence of one bench workload: three SEPARATE rocprofv3 --pmc passes of the same command, python directly
# after "--": FETCH_SIZE, WRITE_SIZE, and the SQ set (matrix-pipe busy cycles, LDS waits, GUI-active clocks).
#   TAG=h ARGS="" KERNELS="collect_kernel" bash tools/pmc_all.sh <outdir>
O=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $O/${TAG}_pmc.txt
for c in FETCH_SIZE WRITE_SIZE SQ; do
  rm -rf $O/pmc_${TAG}_$c
  if [ $c = SQ ]; then ctrs="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; else ctrs=$c; fi
  rocprofv3 --pmc $ctrs --output-format csv -d $O/pmc_${TAG}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-host-pointer ${ARGS:-} > $O/${TAG}_pmc_$c.json 2> $O/${TAG}_pmc_$c.err
  f=$(find $O/pmc_${TAG}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" "${KERNELS:-collect_kernel hnsw_search ivf_bf16 exact_bucket wide_kernel big_kernel}" >> $O/${TAG}_pmc.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:90]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    grid[k] = r.get("Grid_Size", "")
want = sys.argv[2].split()
for k, d in sorted(agg.items()):
    if any(w in k for w in want):
        for c, v in sorted(d.items()):
            print(f"{c} kernel={k!r} dispatches={len(v)} grid_size={grid[k]} mean={sum(v)/len(v):.6g} last={v[-1]:.6g}")
PY
  rm -rf $O/pmc_${TAG}_$c
done
cat $O/${TAG}_pmc.txt
