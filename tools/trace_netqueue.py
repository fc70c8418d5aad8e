"""Development-only: the hardware queue that carries the network branch of a graph replay (the one with the most
gemm_f32_mfma launches), its busy time, its idle gaps and what ran on the OTHER queues during the largest gaps.
usage: trace_netqueue.py run_kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "sgd_clip_kernel" in r["Kernel_Name"]]
ends = [int(rows[i]["End_Timestamp"]) for i in marks]
steps = [(ends[i], ends[i + 1]) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 2e6][-3:]
for (t0, t1) in steps:
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    byq = collections.defaultdict(list)
    for r in sel:
        byq[r["Queue_Id"]].append(r)
    netq = max(byq, key=lambda q: sum("gemm_f32_mfma" in r["Kernel_Name"] for r in byq[q]))
    print("step %.2f ms, %d launches, queues: %s" % ((t1 - t0) / 1e6, len(sel), {q: len(v) for q, v in byq.items()}))
    for q, ks in byq.items():
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3
        print("  queue %s%s: %d launches, busy %.0f us, from +%.0f to +%.0f us" % (
            q, " (network)" if q == netq else "", len(ks), busy, (int(ks[0]["Start_Timestamp"]) - t0) / 1e3,
            (int(ks[-1]["End_Timestamp"]) - t0) / 1e3))
    ks = byq[netq]
    gaps = [((int(ks[i + 1]["Start_Timestamp"]) - int(ks[i]["End_Timestamp"])) / 1e3, i) for i in range(len(ks) - 1)]
    pos = [g for g, _ in gaps if g > 0]
    print("  network queue gaps: %d positive, sum %.0f us; > 2 us: %d sum %.0f us; > 10 us: %d sum %.0f us" % (
        len(pos), sum(pos), len([g for g in pos if g > 2]), sum(g for g in pos if g > 2),
        len([g for g in pos if g > 10]), sum(g for g in pos if g > 10)))
t0, t1 = steps[-1]
for g, i in sorted(gaps, reverse=True)[:12]:
    a, b = ks[i], ks[i + 1]
    ga, gb = int(a["End_Timestamp"]), int(b["Start_Timestamp"])
    others = [r["Kernel_Name"][:40] for r in sel if r["Queue_Id"] != netq and int(r["Start_Timestamp"]) < gb and int(r["End_Timestamp"]) > ga]
    print("gap %7.1f us at +%.0f us after %-50s before %-50s | other queues: %s" % (
        g, (ga - t0) / 1e3, a["Kernel_Name"][:50], b["Kernel_Name"][:50], collections.Counter(others).most_common(3)))
