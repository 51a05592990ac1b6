"""Print the kernels of the last training iteration of a rocprofv3 --kernel-trace CSV, in launch order.
usage: trace_iter.py <kernel_trace.csv> [name filter]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "sgd_momentum" in r["Kernel_Name"]]
# an update is a run of consecutive sgd launches; an iteration = from the end of the previous run to the end of the last one
runs = []
for i in idx:
    if runs and rows[i - 1]["Kernel_Name"] == rows[i]["Kernel_Name"] and i - runs[-1][1] <= 3:
        runs[-1][1] = i
    else:
        runs.append([i, i])
start, end = runs[-2][1] + 1, runs[-1][1] + 1
t0 = int(rows[start]["Start_Timestamp"])
tot = 0.0
agg = {}
for r in rows[start:end]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
    if flt in n:
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f} grid={r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg={r['Workgroup_Size_X']} lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']}+{r['Accum_VGPR_Count']} {n[:80]}")
print("kernel time us", round(tot, 1), "span us", (int(rows[end - 1]["End_Timestamp"]) - t0) / 1e3)
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{d:9.1f} {c:4d} {n[:100]}")
