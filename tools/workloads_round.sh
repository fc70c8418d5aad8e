# usage (GPU box): bash tools/workloads_round.sh  -- the bench line of every workload of DESIGN.md's tables (short form)
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; python3 $R/bench.py --no-cpu-baseline --steps 30 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; c = d.get('contraction') or {}
print('ms/step %.2f  points/s %.3g  pts/step %d  gather %s us frac %.2f  mfma %.1f TF  overflow %s' % (d['ms_per_step'], d['value'], d['config']['points_per_step_per_gpu'], round(r['avg_launch_us'],1), r['frac'], c.get('achieved', 0), d['config']['capacity_overflow']))"; }
run
run --spheres 2
run --spheres 5
run --spheres 8
run --workload baseline
run --workload middle --deformable --views 5
run --workload late --deformable --in-radius 2.0
