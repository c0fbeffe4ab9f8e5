"""Summarise one Euler step from a rocprofv3 kernel trace CSV: per-launch duration and the idle gap before it."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# find the last occurrences of final_kernel -> one full step is between two consecutive final_kernel launches
idx = [i for i, n in enumerate(names) if "final_kernel" in n]
a, b = idx[-3] + 1, idx[-2] + 1
prev_end = int(rows[a - 1]["End_Timestamp"])
tot_k = tot_gap = 0
agg = collections.OrderedDict()
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - prev_end
    prev_end = e
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("dex::", "").split("(")[0][:44]
    grid = f'{r.get("Grid_Size_X","?")}x{r.get("Grid_Size_Y","?")}x{r.get("Grid_Size_Z","?")}'
    print(f"{nm:46s} grid={grid:16s} dur={(e-s)/1e3:8.2f}us gap={gap/1e3:7.2f}us")
    tot_k += e - s; tot_gap += gap
    k = nm
    agg.setdefault(k, [0, 0]); agg[k][0] += 1; agg[k][1] += e - s
print(f"step: {b-a} launches, kernel time {tot_k/1e3:.1f} us, gaps {tot_gap/1e3:.1f} us, total {(tot_k+tot_gap)/1e3:.1f} us")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:46s} x{c:3d} {t/1e3:8.1f} us")
