# development (GPU box): does any runtime knob change the step? usage: bash tools/env_knobs.sh
R=$GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $1 python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
run X=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run HIP_FORCE_DEV_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run GPU_FLUSH_ON_EXECUTION=1
run DEBUG_HIP_KERNARG_COPY_OPT=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run GPU_STREAMOPS_CP_WAIT=1
run AMD_DIRECT_DISPATCH=0
run X=1
