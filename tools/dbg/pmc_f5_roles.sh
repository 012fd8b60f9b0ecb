#!/bin/bash
# fabric traffic and L2 hit rate per kernel of one fp32 DiT evaluation (the linear_x3p instantiations are named per layer role)
O=$PWD/gpurun_out/r4/pmc_roles; mkdir -p $O; ROOT=$PWD
cd /tmp; export TMPDIR=/tmp
C="python $ROOT/tools/pmc_f5_eval.py f32 1 2"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/p_$n -- $C > $O/$n.log 2>&1
done
python $ROOT/tools/pmc_summary.py $O/p_FETCH_SIZE $O/p_WRITE_SIZE $O/p_TCC_HIT_sum $O/p_TCC_EA0_RDREQ_sum > $O/by_kernel.json
python - $O/by_kernel.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if "x3p" in k or "attn" in k:
        print(k[:70], {c: (round(x["per_dispatch"], 1) if isinstance(x, dict) else round(x, 4)) for c, x in v.items()})
PY
rm -rf $O/p_*
