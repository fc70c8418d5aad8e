# usage (on the GPU box): bash tools/prof_steady.sh <tag> [bench args]: rocprofv3 kernel trace of bench.py + steady-state summary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-eager-line --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
F=$(ls $R/gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_steady.py $F 10 200 > $R/gpurun_out/steady_$TAG.txt
python3 $R/tools/trace_branches.py $F 6 > $R/gpurun_out/branches_$TAG.txt
cp $(ls $R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1) $R/gpurun_out/kernel_stats_$TAG.csv
rm -rf $R/gpurun_out/prof_$TAG
head -3 $R/gpurun_out/steady_$TAG.txt
cat $R/gpurun_out/branches_$TAG.txt
