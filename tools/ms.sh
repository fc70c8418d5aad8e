#!/bin/bash
# Runs bench.py with the given arguments and prints only ms/step, points/s and the gather figures of its JSON line.
python bench.py "$@" 2>/dev/null | tail -1 | python -c '
import json,sys
d=json.loads(sys.stdin.read()); r=d["roofline"]
print("ms/step %.3f  points/s %.3e  gather %.1f us frac %.2f  loss %.4f  overflow %s" % (d["ms_per_step"], d["value"], r["avg_launch_us"], r["frac"], d["config"]["final_loss"], d["config"]["capacity_overflow"]))'
