# usage (GPU box): bash tools/prof_netqueue.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-eager-line --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
F=$(ls $R/gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_netqueue.py $F > $R/gpurun_out/netqueue_$TAG.txt
python3 $R/tools/trace_steady.py $F 10 90 > $R/gpurun_out/steady_$TAG.txt
rm -rf $R/gpurun_out/prof_$TAG
