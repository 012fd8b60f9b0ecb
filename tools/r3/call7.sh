#!/bin/bash
# round 3, GPU call 7: attention with K / V^T pre-split by the QKV epilogue
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_f5.py -x -q -rA -m gpu > $O/pytest_c7.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_c7.log
B="timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 5 --warmup 2"
$B > $O/b7_kvp.json 2> $O/b7.err; echo "kvp rc=$?"
MI355TTS_ATTN_KVP=0 $B > $O/b7_nokvp.json 2>> $O/b7.err; echo "nokvp rc=$?"
python - <<'PY'
import json
for f in ("b7_kvp","b7_nokvp"):
    d=json.loads(open(f"gpurun_out/r3/{f}.json").read().strip().splitlines()[-1])
    print(f, round(d["ms_per_step"],1), " | ".join(f"{k['kernel'][:40]} {k['avg_launch_us']:.1f}" for k in d["roofline"]["kernels"][:4]))
PY
