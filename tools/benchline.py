"""Development-only: compact view of bench.py's JSON line(s) read from stdin."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith("DIAG"):
        print(line.split("{")[0].strip())
        line = line[line.index("{"):] if "{" in line else ""
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    c = d.get("cpu_baseline") or {}
    print("ms/step %.3f | %.0f %s | n_gpus %d | loss %s | gather %.1f us frac %.3f | mfma %s | cpu %s | %s" % (
        d["ms_per_step"], d["value"], d["unit"], d["n_gpus"], d["config"].get("final_loss"),
        r.get("avg_launch_us", 0), r.get("frac", 0), (d.get("contraction") or {}).get("achieved"), c.get("value"),
        d["config"].get("execution", "")[:40]))
