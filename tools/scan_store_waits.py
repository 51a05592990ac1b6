"""Scan a hipcc -S listing for stores that sit behind `s_waitcnt vmcnt(0)`: on gfx950 vmcnt counts stores too, so a store loop whose
every element waits for vmcnt(0) (hipcc opens conditional blocks that way while any load is pending) goes out one round trip at a
time.  usage: scan_store_waits.py file.s"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
for k in re.split(r"\n(?=_Z[^\n]*:\s*; @)", s):
    m = re.match(r"(_Z\S+):", k)
    if not m:
        continue
    try:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = m.group(1)
    lines = k.split("s_endpgm")[0].split("\n")
    stores = [i for i, l in enumerate(lines) if re.search(r"\b(global_store|buffer_store|flat_store)", l)]
    if len(stores) < 8:
        continue
    n0 = sum(1 for i in stores if any("s_waitcnt vmcnt(0)" in l for l in lines[max(0, i - 14):i]))
    print("%-100s stores=%4d behind_vmcnt0=%4d" % (name[:100], len(stores), n0))
