#!/bin/bash
# tile form of the second (remainder) launches of the row split, bf16 x 8 utterances: 64x64 two-buffer (default, <= 256 tiles of 128) or 128x128 four-stage ring
for o in 256 150 100 256; do
  python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --option gemm_small16_max=$o > /tmp/a.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("/tmp/a.json").read().strip().splitlines()[-1])
print("gemm_small16_max=$o", round(d["ms_per_step"],1), "ms")
for k in d["roofline"]["kernels"][:8]:
    if "ph8" in k["kernel"] or "attn" in k["kernel"]: continue
    print("     ", k["kernel"][:70], k["launches_per_step"], round(k["avg_launch_us"],1), round(k["ms_per_step"],1))
P
done
