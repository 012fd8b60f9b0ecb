#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_f5.py -x -q -rA -m gpu > gpurun_out/r3/pytest_attn_pairs.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3/pytest_attn_pairs.log; grep -E "F5 full size|DiT evaluation, fp16" gpurun_out/r3/pytest_attn_pairs.log | head
timeout 900 python bench.py --no-pmc > gpurun_out/r3/bench_attn_pairs.json 2> gpurun_out/r3/bench_attn_pairs.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_attn_pairs.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["peak"], d["roofline"].get("kernel"))
for k in d["roofline"].get("kernels", []): print(k["kernel"][:70], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
PY
