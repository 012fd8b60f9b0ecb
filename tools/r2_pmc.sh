#!/bin/bash
# PMC passes on a SHORT command (2 DiT evaluations, ~520 launches): tools/pmc_f5_eval.py <dtype> <U> <reps>
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
DT=${1:-f32}; U=${2:-1}
O=$ROOT/gpurun_out/r2pmc_${DT}_u${U}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
C="python $ROOT/tools/pmc_f5_eval.py $DT $U 2"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/p_sq -- $C > $O/sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- $C > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/p_write -- $C > $O/write.log 2>&1
python $ROOT/tools/pmc_summary.py $O/p_sq $O/p_fetch $O/p_write > $O/pmc_by_kernel.json
rm -rf $O/p_sq $O/p_fetch $O/p_write
python - <<PY
import json
d=json.load(open("$O/pmc_by_kernel.json"))
for k,v in list(d.items())[:8]:
    print(k[:90])
    for c in ("SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CU_CYCLES","SQ_WAVE_CYCLES","SQ_WAIT_ANY","SQ_WAIT_INST_ANY","SQ_ACTIVE_INST_ANY","FETCH_SIZE","WRITE_SIZE"):
        if c in v: print("    %-28s per dispatch %.4g  (n=%d)"%(c,v[c]["per_dispatch"],v[c]["dispatches"]))
    for c in ("mfma_busy_over_busy_cu_cycles","hbm_GB_corrected_per_dispatch"):
        if c in v: print("    %-28s %.4g"%(c,v[c]))
PY
tail -3 $O/sq.log
