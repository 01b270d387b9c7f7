#!/bin/bash
# HBM-side bytes of a bench workload's kernels: two SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), python directly after "--".
# usage: TAG=c3 ARGS="--index IVF4096,Flat --data clustered" bash tools/r4_steps/pmc_hbm.sh <outdir>
O=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f $O/${TAG}_pmc_hbm.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_${TAG}_$c
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_${TAG}_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-configs --no-host-pointer ${ARGS:-} > $O/${TAG}_pmc_$c.json 2> $O/${TAG}_pmc_$c.err
  f=$(find $O/pmc_${TAG}_$c -name "*counter_collection.csv" | head -1)
  python3 - "$f" $c "${KERNELS:-collect_kernel hnsw_search ivf_bf16 exact}" >> $O/${TAG}_pmc_hbm.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
grid = {}
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        agg[r["Kernel_Name"][:80]].append(float(r["Counter_Value"]))
        grid[r["Kernel_Name"][:80]] = r.get("Grid_Size", "")
want = sys.argv[3].split()
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    if any(w in k for w in want):
        print(f"{sys.argv[2]} kernel={k!r} dispatches={len(v)} grid_size={grid[k]} sum_KiB={sum(v):.6g} mean_KiB_per_dispatch={sum(v)/len(v):.6g} last_KiB={v[-1]:.6g}")
PY
  rm -rf $O/pmc_${TAG}_$c
done
cat $O/${TAG}_pmc_hbm.txt
