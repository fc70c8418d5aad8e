# usage (GPU box): bash tools/profiles_round.sh r02 -- collects the judged profile summaries into profiles/ (copied back via gpurun_out/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; O=$R/gpurun_out/profiles_$TAG; rm -rf $O; mkdir -p $O
CMD="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eager-line"
# 1. kernel trace + stats of the default bench command (graph mode), steady-state summary
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-eager-line > $O/bench_under_profiler.json 2> $O/kt.err
F=$(ls $O/kt/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_steady.py $F 10 90 > $O/${TAG}_steady_state_per_step.txt
python3 $R/tools/trace_by_grid.py $F kpconv_gather > $O/${TAG}_gather_by_grid.txt
for k in kpconv_gather_mfma kpconv_lane_channel kpconv_deform gemm_f32_mfma gemm_f32_stream subsample_cloud_kernel sub_ nb_query_kernel nb_build_kernel nb_hist_kernel nb_scan_kernel nb_scatter_kernel nb_cell_order_kernel rev_fill_kernel pk_count_kernel knn_pruned_kernel bn_finish_apply sgd_clip_kernel; do
  echo "== $k"; python3 $R/tools/trace_by_grid.py $F $k | head -24
done > $O/${TAG}_kernels_by_grid.txt
python3 $R/tools/trace_queues.py $F > $O/${TAG}_hw_queues_per_step.txt 2>&1
python3 $R/tools/trace_chain.py $F all > $O/${TAG}_step_all_queues_in_order.txt 2>&1
cp $(ls $O/kt/*/*_kernel_stats.csv | head -1) $O/${TAG}_bench_graph_kernel_stats.csv
rm -rf $O/kt
# 2. HBM traffic of the KPConv kernels: separate FETCH_SIZE / WRITE_SIZE passes of the same command
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pf -- $CMD > /dev/null 2> $O/pf.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pw -- $CMD > /dev/null 2> $O/pw.err
python3 $R/tools/pmc_traffic.py $(ls $O/pf/*/*_counter_collection.csv | head -1) $(ls $O/pw/*/*_counter_collection.csv | head -1) $O/${TAG}_pmc_gather.json "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-eager-line" > $O/pmc_traffic.log 2>&1
rm -rf $O/pf $O/pw
# 3. MFMA utilisation of the forward contractions
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pm -- python3 $R/tools/mfma_probe.py > /dev/null 2> $O/pm.err
python3 $R/tools/pmc_mfma.py $(ls $O/pm/*/*_counter_collection.csv | head -1) $O/${TAG}_pmc_mfma.json > $O/pmc_mfma.log 2>&1
rm -rf $O/pm
# 4. the network branch alone, kernel by kernel in launch order
MVK_BENCH_DIAG=noside rocprofv3 --kernel-trace --output-format csv -d $O/kn -- python3 $R/bench.py --dev --no-eager-line --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/kn.err
python3 $R/tools/trace_chain.py $(ls $O/kn/*/*_kernel_trace.csv | head -1) > $O/${TAG}_chain_noside.txt 2>&1
rm -rf $O/kn
# 5. operator benches (device time of graph-captured launches) and workload lines
python3 $R/tools/gemm_bench.py > $O/${TAG}_gemm_bench.txt 2>/dev/null
bash $R/tools/workloads_round.sh > $O/${TAG}_workloads.txt 2>&1
# 6. the gather's work list: launch time per order (1 and 8 spheres) and what each order fetches
python3 $R/tools/gather_order_bench.py 66 2>/dev/null > $O/${TAG}_gather_order_bench.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/po -- python3 $R/tools/gather_order_bench.py 66 pmc > /dev/null 2> $O/po.err
python3 $R/tools/pmc_gather_orders.py $(ls $O/po/*/*_counter_collection.csv | head -1) > $O/${TAG}_pmc_gather_orders.txt 2>&1
rm -rf $O/po
cd $R && python3 bench.py > $O/${TAG}_bench_line.json 2> $O/bench_line.err; cp $R/gpurun_out/bench_detail.json $O/${TAG}_bench_detail.json; cd /tmp
ls -la $O; cat $O/pmc_mfma.log; head -5 $O/pmc_traffic.log; head -3 $O/${TAG}_steady_state_per_step.txt
# 7. (round 4) the reference's batch shape: 5 spheres x 5 views per step (train_ScanNet_sphere.py:232, :338)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k5 -- python3 $R/bench.py --spheres 5 --views 5 --steps 10 --warmup 2 --no-cpu-baseline --no-eager-line > $O/${TAG}_bench_line_5spheres.json 2> $O/k5.err
F5=$(ls $O/k5/*/*_kernel_trace.csv | head -1)
python3 $R/tools/trace_steady.py $F5 5 90 > $O/${TAG}_steady_state_5spheres.txt
{ echo "== kpconv_gather (forward gathers and the gather-form feature gradients)"; python3 $R/tools/trace_by_grid.py $F5 kpconv_gather | head -30;
  for k in kpconv_gather_mfma kpconv_lane_channel kpconv_deform gemm_f32_mfma gemm_f32_stream gemm_f32_stream fa_gather_kernel subsample_cloud_kernel nb_query_kernel rev_fill_kernel rev_sort_kernel sgd_clip_kernel; do echo "== $k"; python3 $R/tools/trace_by_grid.py $F5 $k | head -16; done; } > $O/${TAG}_kernels_by_grid_5spheres.txt
rm -rf $O/k5
cd $R && python3 bench.py --spheres 5 --views 5 --steps 20 --warmup 3 --no-cpu-baseline > $O/${TAG}_bench_line_5spheres.json 2> $O/b5.err; cp $R/gpurun_out/bench_detail.json $O/${TAG}_bench_detail_5spheres.json; cd /tmp
# 8. (round 4) deterministic mode against the default, gather-form feature gradient against the atomic scatter
{ for e in X=0 MVK_DETERMINISTIC=1 MVK_REVERSE_DX=0 MVK_GATHER_MFMA=0 MVK_REV_FUSED=0 MVK_BN_FOLD=1 MVK_INPUTS_IN_GRAPH=0 MVK_BENCH_DUMMY_LAUNCHES=100; do echo -n "$e: "; env $e python3 $R/bench.py --dev --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step,', d['value'], d['unit'])"; done;
  for e in X=0 MVK_REVERSE_DX=0; do echo -n "8 spheres, $e: "; env $e python3 $R/bench.py --spheres 8 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step,', d['value'], d['unit'])"; done; } > $O/${TAG}_modes.txt 2>&1
# 9. (round 4) what each side branch costs the step (pieces left out of the captured step: timing only), the input kernels'
#    one-workgroup paths against the multi-workgroup front ends, and the level-0 neighbour kernels stand-alone
cd $R
{ for e in X=0 MVK_BENCH_SKIP=enc MVK_BENCH_SKIP=chain MVK_BENCH_SKIP=fa MVK_BENCH_SKIP=enc,chain,fa MVK_BENCH_SKIP=chain,fa MVK_BENCH_SKIP=enc,fa "MVK_SUB_MULTI_MIN=0 MVK_NB_MULTI_MIN=0" X=0; do echo -n "$e: "; env $e python3 $R/bench.py --dev --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step,', d['value'], d['unit'])"; done;
  for e in X=0 MVK_BENCH_SKIP=enc MVK_BENCH_SKIP=chain MVK_BENCH_SKIP=enc,chain,fa "MVK_SUB_MULTI_MIN=0 MVK_NB_MULTI_MIN=0"; do echo -n "5 spheres x 5 views, $e: "; env $e python3 $R/bench.py --dev --spheres 5 --views 5 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], 'ms/step,', d['value'], d['unit'])"; done; } > $O/${TAG}_side_branches.txt 2>&1
ls -la $O
# 10. (round 5) operator bench of the MFMA gather against the vector kernel (the deformable configurations: tools/config_profiles.sh)
cd $R
MVK_GATHER_MFMA=1 python3 tools/gather_mfma_bench.py mfma > $O/${TAG}_gather_mfma_bench.txt 2>/dev/null
MVK_GATHER_MFMA=0 python3 tools/gather_mfma_bench.py vec 2>/dev/null | grep "level" >> $O/${TAG}_gather_mfma_bench.txt
ls -la $O
