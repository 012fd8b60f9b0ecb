#!/bin/bash
# first GPU pass of the exact-fit kernel: numerics, bare timings, model tests, step A/B
mkdir -p gpurun_out/r6
timeout 600 python tools/dbg/x3d_check.py > gpurun_out/r6/x3d_check.txt 2>&1; echo "check rc $?" >> gpurun_out/r6/x3d_check.txt
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -k "full_size_fp32 or adaln_fold or dit_eval or end_to_end_golden" > gpurun_out/r6/x3d_pytest.txt 2>&1; echo "pytest rc $?" >> gpurun_out/r6/x3d_pytest.txt
for i in 1 2; do
  for on in 0 1; do
    MI355TTS_X3D=$on timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
    python - $on <<'PY' >> gpurun_out/r6/x3d_ab.txt
import json,sys
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or d["roofline"]["kernels"]
print("x3d="+sys.argv[1], round(d["ms_per_step"],2), [(k["kernel"][-40:], round(k["avg_launch_us"],1)) for k in ks[:5]])
PY
  done
done
tail -3 /tmp/b.err >> gpurun_out/r6/x3d_ab.txt
cat gpurun_out/r6/x3d_check.txt; tail -15 gpurun_out/r6/x3d_pytest.txt; cat gpurun_out/r6/x3d_ab.txt
