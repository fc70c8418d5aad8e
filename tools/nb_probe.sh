# usage (GPU box): bash tools/nb_probe.sh <dbg variants...>: kernel trace of tools/nb_probe.py, mean duration of every
# kernel per variant (blocks of 10 launches in time order)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_nbp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_nbp -- python3 $R/tools/nb_probe.py "$@" > $R/gpurun_out/nb_probe.log 2>&1
F=$(ls $R/gpurun_out/prof_nbp/*/*_kernel_trace.csv | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seq = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:40]
    if n.startswith("nb_"):
        seq[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, d in seq.items():
    print("%-42s" % n, " ".join("%6.1f" % (sum(d[i:i + 10]) / len(d[i:i + 10])) for i in range(0, len(d), 10)))
PY
rm -rf $R/gpurun_out/prof_nbp
