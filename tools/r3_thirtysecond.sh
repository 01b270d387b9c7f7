#!/bin/bash
# HNSW: fewer rows in flight per wave, more waves per CU (visited hash 4096 / 2048 slots), repeated (the concurrent build gives a new graph every run)
for rep in 1 2; do for g in 8 4 2; do for h in 1 2048; do
  python3 bench.py --index IDMap,HNSW32 --rows 1000000 --d 768 --normalize --data clustered --sigma 1.0 --no-cpu-baseline --opt hnsw_search_g=$g --opt hnsw_visited_lds=$h 2>/dev/null | python3 -c "
import sys, json; j=json.loads(sys.stdin.read()); r=j['roofline']
print('C5 g=$g visited_lds=$h', j['value'], j['ms_per_step'], r['avg_launch_ms'], r['frac'], j.get('recall_at_10'), r.get('grid'))"
done; done; done
