#!/bin/bash
# round 5's coarse distance kernel with parts switched off (profiling library only: results are WRONG): 1 = no matrix written,
# 2 = no MFMAs, 4 = all stores into one tile (instructions without HBM traffic); ABLS="0 1 2 3 4"
O=$1
export MVS_LIB_PATH=$GRAFT_REPO_ROOT/duckdb-faiss-ext_amd/libmi355faiss_prof.so
for abl in ${ABLS:-0 1 2 3 9 25 33 41 105}; do
  TAG=coarse_abl$abl MINCALLS=12 ROWS=10000000 ARGS="--index IVF4096,Flat --data clustered --opt coarse_abl=$abl ${XOPT:-}" bash tools/r4_steps/kstats.sh $O | grep "coarse_dist" | sed "s/^/coarse_abl=$abl  /" | tee -a $O/coarse_abl2.txt
done
