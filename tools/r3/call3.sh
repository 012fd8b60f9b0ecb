#!/bin/bash
# round 3, GPU call 3: gemm_x3p ablations per DiT shape (micro-benchmark), model bench with fused rownorm->planes + PMC traffic
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
{
for dbg in 0 4 1 2 3 8 5 7 12; do echo "== x3p MI355TTS_GEMM_DBG=$dbg"; MI355TTS_GEMM_DBG=$dbg ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH; done
echo "== x3p noalign"; MI355TTS_X3P_NOALIGN=1 ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH
for g in 1 2 4 8; do echo "== x3p grid $g"; MI355TTS_X3P_GRID=$g ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH; done
echo "== x3 (round 2 kernel)"; MI355TTS_F32_X3P=0 ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH
echo "== 3 utterances"; ITERS=100 timeout 300 python tools/gemm_bench.py custom f32 6 1126 1024 3072 1 1 f32 6 1126 1024 1024 1 1 f32 6 1126 1024 2048 1 1 f32 6 1126 2048 1024 1 1
} > $O/x3p_ablation.txt 2>&1
cat $O/x3p_ablation.txt
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -rA -m gpu -k "full_size_fp32 or golden or batched_utterances" > $O/pytest_c3.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_c3.log
timeout 900 python bench.py --no-cpu-baseline --no-secondary --steps 5 --warmup 2 > $O/b3_x3p_fusedln.json 2> $O/b3.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3/b3_x3p_fusedln.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["traffic"], d["roofline"].get("traffic_detail"))
for k in d["roofline"]["kernels"]: print(k["kernel"][:50], round(k["avg_launch_us"],1), k["launches_per_step"], round(k["ms_per_step"],1))
PY
