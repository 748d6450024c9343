"""Summarise rocprofv3 counter_collection.csv for kernels matching a substring."""
import collections
import csv
import sys

f, pat = sys.argv[1], sys.argv[2]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if pat in r["Kernel_Name"]:
        d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()):
    print("%-28s n=%d last=%.4g" % (k, len(v), v[-1]))
