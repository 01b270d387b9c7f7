#!/bin/bash
# HNSW after the LDS chain is gone: rows in flight x waves per CU once more
for g in 16 8; do for w in 0 12 16; do
  python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --no-cpu-baseline --opt hnsw_search_g=$g --opt hnsw_search_waves=$w 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 g=$g waves=$w', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'))"
done; done
