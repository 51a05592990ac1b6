"""Mean of every collected counter per kernel name from rocprofv3 --pmc counter_collection CSVs (one or more passes).
usage: pmc_kernels.py <name filter> pass1.csv [pass2.csv ...]"""
import collections
import csv
import re
import sys

csv.field_size_limit(1 << 30)
flt = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").strip()
        if flt in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(agg.items()):
    print(name)
    for c, v in sorted(cs.items()):
        print("    %-32s n=%4d mean=%14.1f" % (c, len(v), sum(v) / len(v)))
