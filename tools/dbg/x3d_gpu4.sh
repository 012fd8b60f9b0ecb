#!/bin/bash
mkdir -p gpurun_out/r6
O=gpurun_out/r6/run5.txt; rm -f $O
echo "== SK attention test" >> $O
timeout 600 python -m pytest tests/test_gpu_f5.py -x -q -k "attention or full_size_fp32_against or range_watch" >> $O 2>&1
echo "== bare GEMM, hot weights" >> $O
X3D_TIMING_ONLY=1 timeout 300 python tools/dbg/x3d_check.py >> $O 2>&1
echo "== bare GEMM, 40 weight sets (HBM-cold weights)" >> $O
X3D_TIMING_ONLY=1 MI355TTS_BENCH_WSETS=40 timeout 300 python tools/dbg/x3d_check.py >> $O 2>&1
ab() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-secondary --no-cpu-baseline --no-pmc --steps 8 --warmup 3 > /tmp/b.json 2>/tmp/b.err
  python - "$label" <<'PY' >> gpurun_out/r6/run5.txt
import json,sys
d=json.load(open("bench_detail.json"))
ks=d["roofline"].get("instantiations") or []
at=[k for k in d["roofline"]["kernels"] if "attn" in k["kernel"]]
print(sys.argv[1], round(d["ms_per_step"],2), d["config"]["arithmetic_kind"], d["config"]["saturation_events"], [(k["kernel"][-22:], round(k["avg_launch_us"],1)) for k in ks[:4]], [(k["kernel"][-30:], round(k["avg_launch_us"],1)) for k in at])
PY
}
for i in 1 2 3; do ab base MI355TTS_ATTN_SK=0 MI355TTS_GEMM_DBG=64; ab sk MI355TTS_ATTN_SK=1 MI355TTS_GEMM_DBG=64; ab sk_vfast MI355TTS_ATTN_SK=1 MI355TTS_GEMM_DBG=0; done
ab mainloop_only MI355TTS_ATTN_SK=1 MI355TTS_GEMM_DBG=68
cat $O
