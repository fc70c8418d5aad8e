# usage (GPU box): bash tools/profiles_round.sh r02 -- collects the judged profile summaries into profiles/ (copied back via gpurun_out/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; O=$R/gpurun_out/profiles_$TAG; rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
# 1. kernel trace + stats of the default bench command (graph mode), steady-state summary
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_under_profiler.json 2> $O/kt.err
F=$(ls $O/kt/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_steady.py $F 10 90 > $O/${TAG}_steady_state_per_step.txt
python3 $R/tools/trace_by_grid.py $F kpconv_gather_vec > $O/${TAG}_gather_by_grid.txt
cp $(ls $O/kt/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_graph_kernel_stats.csv
rm -rf $O/kt
# 2. HBM traffic of the KPConv kernels: separate FETCH_SIZE / WRITE_SIZE passes of the same command
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- $CMD > /dev/null 2> $O/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- $CMD > /dev/null 2> $O/pw.err
python3 $R/tools/pmc_traffic.py $(ls $O/pf/*/*_counter_collection.csv | head -1) $(ls $O/pw/*/*_counter_collection.csv | head -1) $O/${TAG}_pmc_gather.json "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline" > $O/pmc_traffic.log 2>&1
rm -rf $O/pf $O/pw
# 3. MFMA utilisation of the forward contractions
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm -- python3 $R/tools/mfma_probe.py > /dev/null 2> $O/pm.err
python3 $R/tools/pmc_mfma.py $(ls $O/pm/*/*_counter_collection.csv | head -1) $O/${TAG}_pmc_mfma.json > $O/pmc_mfma.log 2>&1
rm -rf $O/pm
ls -la $O; cat $O/pmc_mfma.log; head -5 $O/pmc_traffic.log; head -3 $O/${TAG}_steady_state_per_step.txt
