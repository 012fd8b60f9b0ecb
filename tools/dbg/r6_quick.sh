#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -k "attention or full_size_fp32_against or golden" 2>&1 | tail -3
for i in 1 2 3; do
  timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python - <<'PY'
import json
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or []
at=[k for k in d["roofline"]["kernels"] if "attn" in k["kernel"]]
print(round(d["ms_per_step"],2), d["config"]["saturation_events"], [(k["kernel"][-22:], round(k["avg_launch_us"],1)) for k in ks[:4]], [(k["kernel"][-30:], round(k["avg_launch_us"],1)) for k in at])
PY
done
