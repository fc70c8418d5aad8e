# usage (on the GPU box): bash tools/prof_chain.sh <tag> [bench args]: kernel trace of bench.py -> ordered listing of
# one steady step's network queue (tools/trace_chain.py), steady-state per-kernel table, per-queue summary
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
rm -rf $R/gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --dev --no-eager-line --steps 20 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.json 2> $R/gpurun_out/prof_$TAG.err
F=$(ls $R/gpurun_out/prof_$TAG/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_chain.py $F > $R/gpurun_out/chain_$TAG.txt
python3 $R/tools/trace_chain.py $F all > $R/gpurun_out/chain_all_$TAG.txt
python3 $R/tools/trace_steady.py $F 10 90 > $R/gpurun_out/steady_$TAG.txt
python3 $R/tools/trace_netqueue.py $F > $R/gpurun_out/netqueue_$TAG.txt
cp $(ls $R/gpurun_out/prof_$TAG/*/*_kernel_stats.csv | head -1) $R/gpurun_out/kernel_stats_$TAG.csv
rm -rf $R/gpurun_out/prof_$TAG
head -3 $R/gpurun_out/steady_$TAG.txt
head -8 $R/gpurun_out/netqueue_$TAG.txt
