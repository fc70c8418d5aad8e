# usage (GPU box): bash tools/ab_bench.sh "<label>=<ENV=.. ENV=..>|<bench args>" ...   -> one line per run: label ms_per_step value
# development: A/B of environment switches on the bench step
# Every variant's stderr is kept (gpurun_out/ab_<label>.err) and its tail printed when no JSON line arrives: a forced plan
# without an instantiation, a removed switch or a GPU fault in one variant must not look like a missing row.
set -o pipefail
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%=*}"; rest="${spec#*=}"; envs="${rest%%|*}"; args=""
  case "$rest" in *"|"*) args="${rest#*|}";; esac
  case "$envs" in *MVK_BENCH_*) args="$args --dev";; esac      # knobs that change the captured step need the explicit flag
  err="gpurun_out/ab_${label}.err"
  out="$(env $envs python bench.py --no-cpu-baseline $args 2>"$err" | tail -1)"
  rc=$?
  if ! printf '%s' "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$label', d['ms_per_step'], d['value'])" 2>/dev/null; then
    echo "$label FAILED rc=$rc (no JSON line); last lines of $err:"
    tail -5 "$err" | sed 's/^/    /'
  fi
done
