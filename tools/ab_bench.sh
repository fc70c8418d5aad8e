# usage (GPU box): bash tools/ab_bench.sh "<label>=<ENV=.. ENV=..>|<bench args>" ...   -> one line per run: label ms_per_step value
# development: A/B of environment switches on the bench step
for spec in "$@"; do
  label="${spec%%=*}"; rest="${spec#*=}"; envs="${rest%%|*}"; args=""
  case "$rest" in *"|"*) args="${rest#*|}";; esac
  env $envs python bench.py --no-cpu-baseline $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['value'])"
done
