"""Development-only: average duration per (kernel, grid size) of a rocprofv3 kernel trace. usage: trace_by_grid.py csv substr"""
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        grid = r.get("Grid_Size") or "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        a = acc[(r["Kernel_Name"][:70], grid, r.get("LDS_Block_Size", ""))]
        a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("%-72s grid %8s lds %6s n %4d avg %8.1f us" % (k[0], k[1], k[2], v[0], v[1] / v[0]))
