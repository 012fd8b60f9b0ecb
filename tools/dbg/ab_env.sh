#!/bin/bash
# same-box A/B of one environment switch: tools/dbg/ab_env.sh VAR=VALUE [bench args]   (base = without it)
# prints ms_per_step and the per-kernel launch averages (the BENCH_DETAIL line bench.py writes to stderr)
KV=$1; shift
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc $*"
S='
import json,sys
ms=None; ks=[]
for l in sys.stdin:
    l=l.strip()
    if l.startswith("BENCH_DETAIL "):
        d=json.loads(l[13:]); ks=[(k["kernel"][:44], k["launches_per_step"], round(k["avg_launch_us"],1)) for k in d["roofline"].get("kernels",[])[:8]]
    elif l.startswith("{"):
        try: ms=json.loads(l)["ms_per_step"]
        except Exception: pass
print(sys.argv[1], round(ms,2) if ms else None, ks)'
for i in 1 2; do $B 2>&1 | python -c "$S" base; env $KV $B 2>&1 | python -c "$S" "$KV"; done
