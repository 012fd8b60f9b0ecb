#!/bin/bash
mkdir -p gpurun_out/r6
ab() {  # label, env assignments...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python - "$label" <<'PY' >> gpurun_out/r6/vfast_ab.txt
import json,sys
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or d["roofline"]["kernels"]
print(sys.argv[1], round(d["ms_per_step"],2), [(k["kernel"][-30:], round(k["avg_launch_us"],1)) for k in ks[:4]])
PY
}
rm -f gpurun_out/r6/vfast_ab.txt
for i in 1 2 3; do ab vfast_on MI355TTS_GEMM_DBG=0; ab vfast_off MI355TTS_GEMM_DBG=64; done
cat gpurun_out/r6/vfast_ab.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6/tests_gpu_mid.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/tests_gpu_mid.log
tail -8 gpurun_out/r6/tests_gpu_mid.log
timeout 900 python bench.py > gpurun_out/r6/bench_default_mid.json 2> gpurun_out/r6/bench_default_mid.err; echo "bench rc $?"
cp bench_detail.json gpurun_out/r6/bench_default_mid_detail.json
cat gpurun_out/r6/bench_default_mid.json
