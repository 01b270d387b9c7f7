#!/bin/bash
# data sensitivity of the Flat L2 headline shape on the final tree: all kinds at N = 10 M, all-duplicates at 2 M (as in round 4)
O=$1
KINDS="uniform clustered normalised offset integer dup10 sift_like" timeout 2400 python3 tools/collect_sensitivity.py 2>&1 | grep -v amdgpu.ids | tee $O/sens.txt
KINDS="all_dup" N=2000000 timeout 900 python3 tools/collect_sensitivity.py 2>&1 | tail -1 | tee -a $O/sens.txt
