#!/bin/bash
O=$1
for o in "cl_bound_mode=1" "cl_bound_mode=0"; do
  echo "## OPTS=$o" | tee -a $O/sens.txt
  OPTS="$o" KINDS="${KINDS:-uniform clustered normalised offset integer dup10 sift_like}" N=${N:-10000000} timeout 900 python3 tools/collect_sensitivity.py 2>&1 | tee -a $O/sens.txt
done
