#!/bin/bash
# ablation of gconv_pairs_kernel (MI355TTS_GCONV_DBG): per-launch time in the model (one utterance) with parts switched off
for d in 0 1 2 4 8 16 24 31; do
  MI355TTS_GCONV_DBG=$d python bench.py --steps 3 --warmup 1 --no-secondary --no-cpu-baseline --no-pmc > /tmp/g.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("/tmp/g.json").read().strip().splitlines()[-1])
k=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("gconv")]
a=[x for x in d["roofline"]["kernels"] if x["kernel"].startswith("attn")]
print("dbg=$d", "gconv", round(k[0]["avg_launch_us"],1) if k else None, "us   (attention on this box:", round(a[0]["avg_launch_us"],1), "us)")
P
done
