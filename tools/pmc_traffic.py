"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, counter_collection.csv) into per-kernel HBM
traffic per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950:
bytes = 2 * FETCH_SIZE_KB * 1024 (FETCH_SIZE tallies 128-B requests at 64 B) + WRITE_SIZE_KB * 1024.
usage: pmc_traffic.py fetch.csv write.csv out.json"""
import collections
import csv
import json
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    return name.replace("void ", "").strip()


def mean_per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


fetch = mean_per_kernel(sys.argv[1], "FETCH_SIZE")
write = mean_per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) & set(write)):
    if not k.startswith("dim::"):
        continue
    f, n = fetch[k]
    w, _ = write[k]
    out[k] = {"launches": n, "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024}
json.dump({"correction": "2*FETCH_SIZE + WRITE_SIZE (KB -> bytes), MI355X_MICROARCH.md HBM section", "kernels": out}, open(sys.argv[3], "w"),
          indent=1)
for k, v in out.items():
    print("%-70s n=%4d  %10.2f MB/launch" % (k[:70], v["launches"], v["hbm_bytes_per_launch"] / 1e6))
