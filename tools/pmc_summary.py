"""summarise rocprofv3 --pmc counter_collection.csv files: mean per dispatch for kernels matching a substring"""
import collections
import csv
import sys

pat = sys.argv[1]
for f in sys.argv[2:]:
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("{:34s} n={:3d} mean={:18.1f}".format(c, len(v), sum(v) / len(v)))
