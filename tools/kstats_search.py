"""per-search kernel breakdown from a rocprofv3 kernel_stats.csv: only kernels launched a multiple of `calls` times (= once or more per
search call of the bench), average per search.  usage: kstats_search.py <kernel_stats.csv> <search calls in the run>"""
import csv, sys
n = int(sys.argv[2])
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    c = int(r["Calls"])
    if c >= n and c % n == 0:
        rows.append((float(r["TotalDurationNs"]) / n / 1e6, c // n, r["Name"][:100]))
tot = sum(x[0] for x in rows)
for ms, per, name in sorted(rows, reverse=True):
    print("%8.4f ms  x%-3d %s" % (ms, per, name))
print("%8.4f ms  sum of the kernels of one search" % tot)
