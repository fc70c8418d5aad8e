"""Development-only: per-kernel means of the counters of one rocprofv3 --pmc pass.
usage: pmc_kernel.py counter_collection.csv <kernel-name substring>   -> one row per (kernel, grid): counter means per launch"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"].split("(")[0][-48:], "grid": r["Grid_Size"], "c": {}})
    d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
groups = collections.OrderedDict()
for d in disp.values():
    groups.setdefault((d["k"], d["grid"]), []).append(d["c"])
for (k, grid), cs in groups.items():
    names = sorted(cs[0])
    mean = {n: sum(c.get(n, 0.0) for c in cs) / len(cs) for n in names}
    print("%s grid %s (%d launches)" % (k, grid, len(cs)))
    for n in names:
        print("    %-28s %14.0f" % (n, mean[n]))
