#!/bin/bash
# The IVF coarse distance matrix kernel with parts switched off (profiling library only: results are WRONG): 1 = no matrix written,
# 2 = no MFMA loop, 3 = staging only.
O=$1
export MVS_LIB_PATH=$GRAFT_REPO_ROOT/duckdb-faiss-ext_amd/libmi355faiss_prof.so
for abl in 0 1 2 3; do
  TAG=coarse_abl$abl MINCALLS=12 ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered --opt coarse_abl=$abl" bash tools/r4_steps/kstats.sh $O | grep "coarse_dist\|coarse_select" | sed "s/^/coarse_abl=$abl  /" | tee -a $O/coarse_abl.txt
done
