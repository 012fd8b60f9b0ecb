#!/bin/bash
# round 3, GPU call 6: whole GPU suite on the current build, default bench (all legs), MFMA-on-real-operands ceiling of x3p
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
{
for dbg in 0 16 7 4; do echo "== x3p (tuning build) MI355TTS_GEMM_DBG=$dbg"; MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_tune.so MI355TTS_GEMM_DBG=$dbg ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH; done
} > $O/x3p_ablation4.txt 2>&1
cat $O/x3p_ablation4.txt
timeout 2400 python -m pytest tests -x -q -rA -m gpu > $O/pytest_c6.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_c6.log
timeout 1200 python bench.py > $O/bench_c6.json 2> $O/bench_c6.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/bench_c6.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["roofline"]["traffic"])
for k in d["roofline"]["kernels"]: print(k["kernel"][:50], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
for n,s in d["secondary"].items(): print(n, round(s["ms_per_step"],1), round(s["value"],1))
print(d["cpu_baseline"]["value"], d["config"]["host_io_ms_per_step"])
PY
