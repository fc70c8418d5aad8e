"""Development-only: per-branch timeline of the steady-state steps of a rocprofv3 kernel trace (csv).
usage: trace_branches.py run_kernel_trace.csv [n_last_steps]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ENC = ("miopen", "igemm", "batched_transpose", "MIOpenBatchNorm", "launch_clamp", "bias_act", "max_pool_forward_nchw", "Cijk", "SubTensorOp", "transpose_NCHW", "naive_conv", "OpTensor")
CHAIN = ("nb_", "subsample", "rotate_", "pk_", "knn_", "unproject", "pad_tail", "offsets_from", "pad_points", "pad_index", "center_", "fa_gather")
def branch(n):
    if any(k in n for k in ENC): return "enc"
    if any(k in n for k in CHAIN): return "chain"
    return "net"
marks = [i for i, r in enumerate(rows) if "sgd_clip_kernel" in r["Kernel_Name"] or "multi_tensor_apply" in r["Kernel_Name"]]
ends = [int(rows[i]["End_Timestamp"]) for i in marks]
# a step = from the end of one optimiser launch to the end of the next
steps = [(ends[i], ends[i + 1]) for i in range(len(ends) - 1) if ends[i + 1] - ends[i] > 2e6][-nlast:]
for (t0, t1) in steps:
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
    out = ["step %.2f ms, %d launches" % ((t1 - t0) / 1e6, len(sel))]
    for b in ("net", "enc", "chain"):
        ks = [r for r in sel if branch(r["Kernel_Name"]) == b]
        if not ks: continue
        dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks) / 1e3
        span = (int(ks[-1]["End_Timestamp"]) - int(ks[0]["Start_Timestamp"])) / 1e3
        gaps = [(int(ks[i + 1]["Start_Timestamp"]) - int(ks[i]["End_Timestamp"])) / 1e3 for i in range(len(ks) - 1)]
        pos = [g for g in gaps if g > 0]
        out.append("%s: %d launches, busy %.0f us, span %.0f us (starts at +%.0f us), gaps>0: %d sum %.0f us, median %.1f, >5us: %d sum %.0f" % (
            b, len(ks), dur, span, (int(ks[0]["Start_Timestamp"]) - t0) / 1e3, len(pos), sum(pos), sorted(pos)[len(pos) // 2] if pos else 0,
            len([g for g in pos if g > 5]), sum(g for g in pos if g > 5)))
    print(" | ".join(out))
# biggest gaps of the network branch in the last step
t0, t1 = steps[-1]
ks = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1 and branch(r["Kernel_Name"]) == "net"]
gl = sorted([((int(ks[i + 1]["Start_Timestamp"]) - int(ks[i]["End_Timestamp"])) / 1e3, ks[i]["Kernel_Name"][:60], ks[i + 1]["Kernel_Name"][:60]) for i in range(len(ks) - 1)], reverse=True)
for g in gl[:15]: print("gap %.1f us after %s before %s" % g)
