#!/bin/bash
# same-box A/B of two builds: libmi355faiss_prev.so (a copy of the previous build) against libmi355faiss.so, alternating
O=$1
for v in prev new prev new; do
  if [ $v = prev ]; then export MVS_LIB_PATH=$PWD/duckdb-faiss-ext_amd/libmi355faiss_prev.so; else unset MVS_LIB_PATH; fi
  echo "lib=$v"
  SHAPES="${SHAPES:-10000000 1250000 1000000}" bash tools/r4_steps/shapes.sh $O 2>&1 | tail -3
done
