cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export MVK_GEMM_FORCE=5,1,1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU -d $R/gpurun_out/pmc_g1 --output-format csv -- python3 $R/tools/gemm_one.py 19464 64 990 0 0 > $R/gpurun_out/pmc_g1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM -d $R/gpurun_out/pmc_g2 --output-format csv -- python3 $R/tools/gemm_one.py 19464 64 990 0 0 > $R/gpurun_out/pmc_g2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R=os.environ["GRAFT_REPO_ROOT"]
for d in ("pmc_g1","pmc_g2"):
    for f in glob.glob(R+"/gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(lambda:[0,0.0])
        for r in csv.DictReader(open(f)):
            if "gemm_f32" in r["Kernel_Name"]:
                a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
        for k,v in acc.items(): print(d,k,v[1]/v[0])
PY
