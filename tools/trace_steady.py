"""Development-only: per-kernel time of the steady-state steps of a rocprofv3 kernel trace (csv).

usage: trace_steady.py run_kernel_trace.csv [n_last_steps] -- a step boundary is every launch of the fused clip + SGD
kernel (one per step whatever the number of spheres; traces without one: the unprojection kernel of build_batch)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [int(r["End_Timestamp"]) for r in rows if "sgd_clip_kernel" in r["Kernel_Name"]]
if len(marks) < 3:
    marks = [int(r["Start_Timestamp"]) for r in rows if "unproject_kernel" in r["Kernel_Name"]]
nlast = min(nlast, len(marks) - 1)
t0, t1 = marks[-nlast - 1], marks[-1]
sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in sel:
    n = r["Kernel_Name"]
    agg[n][0] += 1
    agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("steady window: %d steps, %.2f ms wall per step, %.2f ms kernel time per step, %d launches per step" % (
    nlast, (t1 - t0) / 1e6 / nlast, tot / 1e3 / nlast, len(sel) / nlast))
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print("%-100s %6.1f/step %8.1f us/step %7.1f us avg" % (n[:100], c / nlast, us / nlast, us / c))
