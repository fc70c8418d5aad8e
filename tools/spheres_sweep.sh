#!/bin/bash
# bench.py over several batch sizes (spheres per GPU); prints ms/step, points/s and the roofline fraction
for s in "$@"; do
  python bench.py --spheres $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('spheres', $s, 'ms/step %.2f' % d['ms_per_step'], 'points/s %.0f' % d['value'], d['config']['execution'][:8], 'gather frac %.3f' % d['roofline']['frac'], 'launch us %.1f' % d['roofline']['avg_launch_us'])"
done
