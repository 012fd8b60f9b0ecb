#!/bin/bash
# PMC on the GEMM micro-benchmark: usage pmc_gemm.sh "<env assignments>" <gemm_bench custom args...>
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ENVS="$1"; shift
O=$ROOT/gpurun_out/pmc_gemm; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
env $ENVS ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/p1 -- python $ROOT/tools/gemm_bench.py custom "$@" > $O/log1.txt 2>&1
env $ENVS ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p3 -- python $ROOT/tools/gemm_bench.py custom "$@" > $O/log3.txt 2>&1
env $ENVS ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p4 -- python $ROOT/tools/gemm_bench.py custom "$@" > $O/log4.txt 2>&1
env $ENVS ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/p5 -- python $ROOT/tools/gemm_bench.py custom "$@" > $O/log5.txt 2>&1
env $ENVS ITERS=5 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $O/p2 -- python $ROOT/tools/gemm_bench.py custom "$@" > $O/log2.txt 2>&1
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-60:]
        if "gemm" not in k and "linear" not in k: continue
        agg[(k,r["Grid_Size"] if "Grid_Size" in r else "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print("   %-26s mean %.4g  (n=%d)"%(c,sum(vals)/len(vals),len(vals)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CU_CYCLES" in v:
        print("   mfma duty = %.3f"%(sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])/sum(v["SQ_BUSY_CU_CYCLES"])/4))
PY
grep custom $O/log1.txt
