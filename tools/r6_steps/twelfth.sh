#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
MVS_INGEST_PROFILE=1 WGS="1" timeout 600 python tools/hnsw_build_probe.py 40960 768 40 > $O/r6_hnsw_spins.txt 2>&1; grep -v amdgpu $O/r6_hnsw_spins.txt | tail -12 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
WGS="4" rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/hnsw_trace -o t -- python3 $GRAFT_REPO_ROOT/tools/hnsw_build_probe.py 40960 768 40 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/hnsw_trace/**/*kernel_trace.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "hnsw_build_kernel" in r["Kernel_Name"]]
print(len(rows), "build launches; last 16: grid(waves) duration_ms")
for r in rows[-16:]:
    print(int(r["Grid_Size_X"] if "Grid_Size_X" in r else r["Grid_Size"]) , round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2))
PY
rm -rf gpurun_out/hnsw_trace
