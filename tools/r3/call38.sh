#!/bin/bash
# QKV row split (m_off in the QKV epilogues): parity of the 16-bit engines at 8 utterances and one utterance, then the A/B
python -m pytest tests/test_gpu_f5.py -q -x -k "lowp or fp16_transformer or attention or batch or lowp_gate" -rA 2>&1 | grep -E "F5 full|passed|failed|Error" | tail -12
for o in 0 1; do
  python bench.py --dtype bf16 --batch 8 --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-secondary --option gemm_row_split=$o > gpurun_out/s6_u8b_split$o.json 2> gpurun_out/s6_u8b_split$o.err
  python - <<P
import json
d=json.loads(open("gpurun_out/s6_u8b_split$o.json").read().strip().splitlines()[-1])
print("row_split=$o", round(d["ms_per_step"],1), "ms", round(d["value"],1))
for k in d["roofline"]["kernels"][:8]: print("   ", k["kernel"][:80], k["launches_per_step"], round(k["avg_launch_us"],1), round(k["ms_per_step"],1))
P
done
