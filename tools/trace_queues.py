"""Development-only: per hardware queue busy time / launches of the last steady steps of a rocprofv3 kernel trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("columns:", list(rows[0].keys()))
marks = [int(r["End_Timestamp"]) for r in rows if "sgd_clip_kernel" in r["Kernel_Name"]]
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1) if marks[i + 1] - marks[i] > 2e6][-3:]
for t0, t1 in steps:
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    per = collections.defaultdict(lambda: [0, 0.0, None, None, collections.Counter()])
    for r in sel:
        k = (r.get("Queue_Id"), r.get("Stream_Id"))
        p = per[k]
        p[0] += 1
        p[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        p[2] = int(r["Start_Timestamp"]) if p[2] is None else p[2]
        p[3] = int(r["End_Timestamp"])
        p[4][r["Kernel_Name"][:40]] += 1
    print("step %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(sel)))
    for k, p in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  queue/stream %s: %d launches, busy %.0f us, span %.0f us (from +%.0f us); top %s" % (
            k, p[0], p[1], (p[3] - p[2]) / 1e3, (p[2] - t0) / 1e3, p[4].most_common(3)))
