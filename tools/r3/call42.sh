#!/bin/bash
# second look at attn_xcd_map: alternating order, bf16 x 8 utterances and fp32 x 1, attention launch time by HIP events
for o in 1 0 1 0; do
  python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --option attn_xcd_map=$o > /tmp/a.json 2>/dev/null
  python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --option attn_xcd_map=$o > /tmp/b.json 2>/dev/null
  python - <<P
import json
for f, n in (("/tmp/a.json", "bf16 x 8"), ("/tmp/b.json", "fp32 x 1")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    a=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("attn")][0]
    print("attn_xcd_map=$o", n, round(d["ms_per_step"],1), "ms; attention", round(a["avg_launch_us"],2), "us")
P
done
