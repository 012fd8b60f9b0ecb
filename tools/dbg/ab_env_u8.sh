#!/bin/bash
# same-box A/B of one environment switch on the bf16 x 8 utterances line (configs[3] shard): tools/dbg/ab_env_u8.sh VAR=VALUE
KV=$1; shift
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc --dtype bf16 --batch 8 --steps 4 --warmup 2 $*"
S='
import json,sys
ms=None; ks=[]
for l in sys.stdin:
    l=l.strip()
    if l.startswith("BENCH_DETAIL "):
        d=json.loads(l[13:]); ks=[(k["kernel"][:40], k["launches_per_step"], round(k["avg_launch_us"],1)) for k in d["roofline"].get("kernels",[])[:7]]
    elif l.startswith("{"):
        try: ms=json.loads(l)["ms_per_step"]
        except Exception: pass
print(sys.argv[1], round(ms,2) if ms else None, ks)'
for i in 1 2; do $B 2>&1 | python -c "$S" base; env $KV $B 2>&1 | python -c "$S" "$KV"; done
