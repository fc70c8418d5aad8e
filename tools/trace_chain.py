"""Development-only: the kernels of ONE steady-state step on the hardware queue that carries the network branch, in
launch order: offset, duration, gap to the previous kernel, grid, name -- to see where the serial chain spends its time.
usage: trace_chain.py run_kernel_trace.csv [all]     ("all": every queue, with the queue id)"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sgd_clip_kernel" in r["Kernel_Name"]]
ends = [int(rows[i]["End_Timestamp"]) for i in marks]
steps = [(ends[i], ends[i + 1]) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 1e6]
t0, t1 = steps[-2]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
byq = collections.defaultdict(list)
for r in sel:
    byq[r["Queue_Id"]].append(r)
netq = max(byq, key=lambda q: sum("gemm_f32_mfma" in r["Kernel_Name"] for r in byq[q]))


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"at::native::", "", n)
    return n[:64]


print("step %.2f ms; network queue %s: %d launches" % ((t1 - t0) / 1e6, netq, len(byq[netq])))
prev = None
tot = collections.Counter()
for r in (sel if len(sys.argv) > 2 else byq[netq]):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    grid = r.get("Grid_Size") or "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = r.get("Workgroup_Size") or "x".join(r.get(k, "?") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    print("%s+%7.1f us  %6.1f us  gap %5.1f  grid %9s wg %5s  %s" % (
        ("q%s " % r["Queue_Id"]) if len(sys.argv) > 2 else "", (s - t0) / 1e3, (e - s) / 1e3, gap, grid, wg, short(r["Kernel_Name"])))
    if len(sys.argv) <= 2 or r["Queue_Id"] == netq:
        prev = e
    tot[short(r["Kernel_Name"])[:40]] += (e - s) / 1e3
