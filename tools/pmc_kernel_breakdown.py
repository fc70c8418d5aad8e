"""Development-only: per (kernel, grid) means of every counter in a rocprofv3 --pmc counter_collection.csv.
usage: pmc_kernel_breakdown.py counter_collection.csv [kernel substring]"""
import collections, csv, sys
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if sub not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:64], r["Grid_Size"])
    d = acc.setdefault(key, {})
    e = d.setdefault(r["Counter_Name"], {})
    e[r["Dispatch_Id"]] = e.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
for (k, g), d in acc.items():
    print(k, "grid", g)
    for c, e in d.items():
        print("    %-28s %14.0f  (mean of %d launches)" % (c, sum(e.values()) / len(e), len(e)))
