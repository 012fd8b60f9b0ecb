#!/bin/bash
# round 3, GPU call 4: x3p with the DMA slots of the two waves of a SIMD at opposite ends + fused producers (rownorm / attention / FF1 -> planes)
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
{
for dbg in 0 4 1 8 12 7; do echo "== x3p MI355TTS_GEMM_DBG=$dbg"; MI355TTS_GEMM_DBG=$dbg ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH; done
} > $O/x3p_ablation2.txt 2>&1
cat $O/x3p_ablation2.txt
timeout 900 python -m pytest tests/test_gpu_f5.py tests/test_gpu_bigvgan.py -x -q -rA -m gpu -k "full_size_fp32 or golden or batched_utterances or panel_planes" > $O/pytest_c4.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_c4.log
timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 5 --warmup 2 > $O/b4.json 2> $O/b4.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/b4.json").read().strip().splitlines()[-1])
print(d["ms_per_step"])
for k in d["roofline"]["kernels"]: print(k["kernel"][:50], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
PY
