#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export PYTHONPATH=$PWD:$PWD/duckdb-faiss-ext_amd/pyhost
O=gpurun_out
MVS_INGEST_PROFILE=1 duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 IVF4096,Flat 2>&1 | grep -E "ivfprofile|ingestrate|ingest\s" | tee $O/r6_ingest_ivf.log
for t in 8 1; do MVS_INGEST_PROFILE=1 duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 $t IDMap,Flat 2>&1 | grep -E "ingestprofile|ingestrate|ingest\s" ; done | tee $O/r6_ingest.log
duckdb-faiss-ext_amd/host/boundary_driver ingest 10000000 128 8 Flat 2>&1 | grep -E "ingestrate|ingest\s" | tee -a $O/r6_ingest.log
N=12500000 D=768 METRIC=IP KINDS="outlier" timeout 600 python tools/collect_sensitivity.py 2>&1 | grep -v amdgpu | tee $O/r6_sens_d768_ip_outlier.txt | cut -c1-200
N=10000000 D=128 METRIC=IP KINDS="outlier" timeout 600 python tools/collect_sensitivity.py 2>&1 | grep -v amdgpu | tee $O/r6_sens_d128_ip_outlier.txt | cut -c1-200
bash tools/r6_steps/final_measure.sh
