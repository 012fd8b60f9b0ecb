#!/bin/bash
cd /root/repo; mkdir -p gpurun_out/r3; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_f5.py -x -q -m gpu 2>&1 | tail -8
timeout 900 python bench.py --no-pmc > gpurun_out/r3/bench_x2_first.json 2> gpurun_out/r3/bench_x2_first.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_x2_first.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"].get("kernel"))
for k in d["roofline"].get("kernels", []): print(k["kernel"][:60], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
for n,s in d.get("secondary",{}).items(): print(n, round(s["ms_per_step"],1), round(s["value"],1))
PY
