#!/bin/bash
# same-box A/B of one AA environment switch on the BigVGAN fp16 (8,100,512) line: tools/dbg/aa_pipe_ab.sh VAR "v0 v1"
cd ${GRAFT_REPO_ROOT:-.}; V=$1; VALS=$2
for i in 1 2; do for v in $VALS; do echo -n "$V=$v "; env $V=$v python bench.py --workload bigvgan --no-cpu-baseline --no-pmc 2>&1 | python -c '
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith("BENCH_DETAIL "):
        d=json.loads(l[13:]); print(round(d["ms_per_step"],3), [(k["kernel"][:22], k["launches_per_step"], round(k["avg_launch_us"],1)) for k in d["roofline"]["kernels"][:6]])
'; done; done
