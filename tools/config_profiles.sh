# usage (GPU box): bash tools/config_profiles.sh r05 -- BASELINE configs[3] / configs[4] per-rank steps (deformable nets): steady-state
# kernel table + per-grid durations of the KPConv / neighbour-search kernels (rocprofv3 kernel trace), then the bench line and the
# gather launches' achieved fractions of the HBM peak (HIP events inside bench.py, no profiler attached)
R=$GRAFT_REPO_ROOT; TAG=$1; O=$R/gpurun_out/profiles_$TAG; mkdir -p $O
one() {
  name=$1; shift
  bash $R/tools/prof_steady.sh $name "$@" > /dev/null 2>&1
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_kg
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_kg -- python3 $R/bench.py --no-eager-line --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2> $R/gpurun_out/prof_kg.err
  F=$(ls $R/gpurun_out/prof_kg/*/*_kernel_trace.csv | head -1)
  { echo "# python3 bench.py --no-eager-line --steps 20 --warmup 3 --no-cpu-baseline $* (rocprofv3 --kernel-trace --stats)"
    cat $R/gpurun_out/steady_$name.txt
    echo; echo "# kernels_by_grid (second trace of the same command, 10 steps): per (kernel, grid) launches and average duration"
    for k in kpconv_gather_mfma kpconv_gather_vec kpconv_gather_small kpconv_lane_channel kpconv_deform nb_query rev_ knn_pruned subsample_cloud; do
      echo "== $k"; python3 $R/tools/trace_by_grid.py $F $k | head -14
    done
    cd $R
    echo; echo "# bench line of the same workload without a profiler, and its gather launches (forward gathers; [dx] = gather-form feature"
    echo "# gradients over the reverse lists) with algorithmic GB/s and the fraction of the 8 TB/s HBM peak:"
    python3 bench.py --no-cpu-baseline --no-eager-line "$@" 2>/dev/null | tail -1
    python3 - <<'PY'
import json
d = json.load(open("gpurun_out/bench_detail.json"))
for g in d["detail"]["gather_launches"]:
    print("%-58s H %4d Cin %3d Nq %6d..%6d  n %3d  avg %7.1f us  %7.1f GB/s  frac %.3f" % (
        g["kernel"][:58], g["H"], g["Cin"], g["Nq_min"], g["Nq_max"], g["launches"], g["avg_us"], g["GBps"], g["frac_of_hbm_peak"]))
PY
  } > $O/${TAG}_steady_state_$name.txt 2>&1
  rm -rf $R/gpurun_out/prof_kg
}
one config4 --workload middle --deformable --views 5
one config5 --workload late --deformable --in-radius 1.7
ls -la $O | grep config
