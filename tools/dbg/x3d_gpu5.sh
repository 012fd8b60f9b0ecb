#!/bin/bash
mkdir -p gpurun_out/r6
O=gpurun_out/r6/run6.txt; rm -f $O
echo "== numerics + attention tests" >> $O
timeout 300 python tools/dbg/x3d_check.py 2>&1 | grep -v "^T[0-9]" >> $O
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -k "attention or full_size or range_watch or adaln" >> $O 2>&1
ab() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python - "$label" <<'PY' >> gpurun_out/r6/run6.txt
import json,sys
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or []
at=[k for k in d["roofline"]["kernels"] if "attn" in k["kernel"]]
print(sys.argv[1], round(d["ms_per_step"],2), d["config"]["arithmetic_kind"], d["config"]["saturation_events"], [(k["kernel"][-22:], round(k["avg_launch_us"],1)) for k in ks[:4]], [(k["kernel"][-30:], round(k["avg_launch_us"],1)) for k in at])
PY
}
for i in 1 2; do ab lpt_off MI355TTS_ATTN_LPT=0; ab lpt_auto MI355TTS_ATTN_LPT=1; ab cuts_7_14 MI355TTS_ATTN_CUTS=7,14; ab cuts_8_14 MI355TTS_ATTN_CUTS=8,14; ab cuts_7_13 MI355TTS_ATTN_CUTS=7,13; ab cuts_7_14_17 MI355TTS_ATTN_CUTS=7,14,17; ab cuts_6_12_15 MI355TTS_ATTN_CUTS=6,12,15; done
cat $O
