#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests/test_gpu_bigvgan.py tests/test_gpu_f5.py -x -q -rA -m gpu -k "fp16_pairs or full_size" > gpurun_out/r3/pytest_x2.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r3/pytest_x2.log; grep -E "F5 full size|relative rms error" gpurun_out/r3/pytest_x2.log | head -12
timeout 900 python bench.py --no-pmc > gpurun_out/r3/bench_x2_b.json 2> gpurun_out/r3/bench_x2_b.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_x2_b.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["peak"], d["roofline"].get("kernel"))
PY
