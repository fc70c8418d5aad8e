# usage (GPU box): bash tools/config5.sh -- BASELINE configs[4]: late fusion, deformable + modulated, 40 k-point sphere
# (--in-radius 1.7), f32 against the fp16-feature mode (streaming contraction on / off), plus a per-kernel table of each
R=$GRAFT_REPO_ROOT
run() { echo "== $*"; env $1 python3 $R/bench.py --no-cpu-baseline --steps 30 --workload late --deformable --in-radius 1.7 $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; c = d.get('contraction') or {}
print('ms/step %.3f  points/s %.3g  pts/step %d  gather %s us frac %.2f  contraction %.1f TF frac %.3f largest %s  overflow %s' % (d['ms_per_step'], d['value'], d['config']['points_per_step_per_gpu'], round(r['avg_launch_us'],1), r['frac'], c.get('achieved', 0), c.get('frac', 0), c.get('largest'), d['config']['capacity_overflow']))"; }
run "A=1" ""
run "A=1" "--features f16"
run "MVK_GEMM16_STREAM=0" "--features f16"
