#!/bin/bash
# same-box A/B of one environment switch: tools/dbg/ab_env.sh VAR=VALUE [bench args]   (base = without it)
KV=$1; shift
B="python bench.py --no-secondary --no-cpu-baseline --no-pmc $*"
S='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["ms_per_step"],2), [(k["kernel"][:40], round(k["avg_launch_us"],1)) for k in d["roofline"]["kernels"][:6]])'
for i in 1 2; do $B | python -c "$S" base; env $KV $B | python -c "$S" "$KV"; done
