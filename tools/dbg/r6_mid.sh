#!/bin/bash
mkdir -p gpurun_out/r6
O=gpurun_out/r6/run12.txt; rm -f $O
ab() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python - "$label" <<'PY' >> gpurun_out/r6/run12.txt
import json,sys
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or []
at=[k for k in d["roofline"]["kernels"] if "attn" in k["kernel"]]
print(sys.argv[1], round(d["ms_per_step"],2), d["config"]["arithmetic_kind"], d["config"]["saturation_events"], [(k["kernel"][-22:], round(k["avg_launch_us"],1)) for k in ks[:4]], [(k["kernel"][-30:], round(k["avg_launch_us"],1)) for k in at])
PY
}
for i in 1 2 3; do ab il_off MI355TTS_QKV_IL=0; ab il_on MI355TTS_QKV_IL=1; done
cat $O
timeout 1800 python -m pytest tests -x -q -m gpu > gpurun_out/r6/tests_gpu_mid2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/tests_gpu_mid2.log
tail -6 gpurun_out/r6/tests_gpu_mid2.log
