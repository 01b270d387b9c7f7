import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), "%9.3f ms avg" % (float(r["AverageNs"])/1e6), "%8.1f ms total" % (float(r["TotalDurationNs"])/1e6))
