#!/bin/bash
# stream-K for the small fp32 linear layers (proj_out: 18 tiles): A/B of gemm_sk_min_tiles 64 (default) | 16 | 8
for o in 64 16 8; do
  python bench.py --steps 4 --warmup 2 --no-secondary --no-cpu-baseline --no-pmc --option gemm_sk_min_tiles=$o > /tmp/g.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("/tmp/g.json").read().strip().splitlines()[-1])
print("gemm_sk_min_tiles=$o", round(d["ms_per_step"],2), "ms")
for k in d["roofline"]["kernels"][:9]: print("   ", k["kernel"][:84], k["launches_per_step"], round(k["avg_launch_us"],1), round(k["ms_per_step"],2))
P
done
