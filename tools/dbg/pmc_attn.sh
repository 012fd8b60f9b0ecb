#!/bin/bash
# issue / wait / LDS counters of the attention kernels (fp32 one utterance, bf16 eight): tools/dbg/pmc_attn.sh
O=$PWD/gpurun_out/r4/pmc_attn; mkdir -p $O; ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt
for W in "f32 1 1" "bf16 8 1"; do
  set -- $W
  C="python $ROOT/tools/pmc_f5_eval.py $1 $2 $3"
  i=0
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$1_$i -- $C > $O/log_$1_$i.txt 2>&1
  done
  python $ROOT/tools/pmc_summary.py $O/p_$1_1 $O/p_$1_2 $O/p_$1_3 $O/p_$1_4 > $O/by_kernel_$1.json
  python - $O/by_kernel_$1.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "attn" in k:
        print(k[:60])
        for c, x in v.items():
            if isinstance(x, dict): print("   %-32s %14.0f" % (c, x["per_dispatch"]))
PY
  rm -rf $O/p_*
done
