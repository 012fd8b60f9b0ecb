#!/bin/bash
# round 3: PMC picture of the BigVGAN HBM-bound kernels (aa_conv / aa_act): where do the wave cycles go?
cd "$(dirname "$0")/../.."
O=$PWD/gpurun_out/r3; mkdir -p $O
ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
C="python $ROOT/bench.py --workload bigvgan --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pb1 -- $C > $O/pb1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pb2 -- $C > $O/pb2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM --output-format csv -d $O/pb3 -- $C > $O/pb3.log 2>&1
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
import re
for d in ("pb1","pb2","pb3"):
    for f in glob.glob("$O/"+d+"/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=re.sub(r"\(.*\)$","",r["Kernel_Name"].replace("void ","").replace("mi::",""))
            agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): cnt[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get("SQ_BUSY_CU_CYCLES",0))[:8]:
    print(k[:80], "dispatches", cnt[k])
    w=v.get("SQ_WAVE_CYCLES",1)
    for c in sorted(v): print("     %-28s %12.4g  /wave_cycles %.3f" % (c, v[c], v[c]/w))
PY
rm -rf $O/pb1 $O/pb2 $O/pb3
