"""Development-only: FETCH_SIZE of the level-0 gather per work list, from a rocprofv3 --pmc FETCH_SIZE pass over
`tools/gather_order_bench.py <cin> pmc` (three plain launches per order: rows, morton, cells, random; 1 sphere, then 8).
usage: pmc_gather_orders.py <counter_collection.csv>"""
import csv, sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE" and "kpconv_gather_vec" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
names = ("rows", "morton", "cells", "random")
print("FETCH_SIZE of kpconv_gather_vec per work list (KiB as counted; x2 = bytes on gfx950 for 16-B-per-lane reads, MI355X_MICROARCH.md)")
for i in range(0, len(rows) - 2, 3):
    grp = rows[i:i + 3]
    v = sum(float(r["Counter_Value"]) for r in grp) / 3
    print("grid %9d  order %-7s  FETCH_SIZE %10.0f KiB  -> %7.1f MB fetched" % (int(grp[0]["Grid_Size"]), names[(i // 3) % 4], v, 2 * v * 1024 / 1e6))
