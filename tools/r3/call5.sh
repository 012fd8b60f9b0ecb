#!/bin/bash
# round 3, GPU call 5: x3p placement variants (MI355TTS_X3P_VAR) + eight-wave fix-up / epilogue
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
{
for v in 0 1 2 3; do
  echo "== x3p VAR=$v"; MI355TTS_X3P_VAR=$v ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH
  echo "== x3p VAR=$v no fix-up / epilogue"; MI355TTS_X3P_VAR=$v MI355TTS_GEMM_DBG=4 ITERS=200 timeout 300 python tools/gemm_bench.py custom $SH
done
} > $O/x3p_ablation3.txt 2>&1
cat $O/x3p_ablation3.txt
for v in 0 1 2 3; do
  MI355TTS_X3P_VAR=$v timeout 600 python -m pytest tests/test_gpu_f5.py tests/test_gpu_bigvgan.py -x -q -m gpu -k "bf16x3-splits or (panel_planes and 0-0)" > $O/pytest_c5_v$v.log 2>&1; echo "VAR=$v pytest rc=$?"; tail -1 $O/pytest_c5_v$v.log
done
for v in 0 1 2 3; do
MI355TTS_X3P_VAR=$v timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-pmc --steps 5 --warmup 2 > $O/b5_v$v.json 2> $O/b5.err; echo "bench VAR=$v rc=$?"
done
python - <<'PY'
import json
for v in range(4):
    d=json.loads(open(f"gpurun_out/r3/b5_v{v}.json").read().strip().splitlines()[-1])
    print(v, round(d["ms_per_step"],1), " | ".join(f"{k['kernel'][:30]} {k['avg_launch_us']:.1f}" for k in d["roofline"]["kernels"][:3]))
PY
