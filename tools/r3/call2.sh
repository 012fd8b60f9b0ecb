#!/bin/bash
# round 3, GPU call 2: first run of gemm_x3p (panel planes): unit parity, F5 full-size parity, A/B benches
cd "$(dirname "$0")/../.."
O=gpurun_out/r3; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bigvgan.py -x -q -rA -m gpu -k "panel_planes or exact_bf16" > $O/pytest_c2a.log 2>&1; echo "pytest a rc=$?"
tail -4 $O/pytest_c2a.log
timeout 900 python -m pytest tests/test_gpu_f5.py -x -q -rA -m gpu -k "full_size_fp32 or golden or batched_utterances" > $O/pytest_c2b.log 2>&1; echo "pytest b rc=$?"
tail -6 $O/pytest_c2b.log
B="timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-secondary --steps 5 --warmup 2"
$B > $O/b2_x3p.json 2> $O/b2_x3p.err; echo "x3p rc=$?"
MI355TTS_F32_X3P=0 $B > $O/b2_x3.json 2> $O/b2_x3.err; echo "x3 rc=$?"
MI355TTS_X3P_NOALIGN=1 $B > $O/b2_x3p_noalign.json 2>/dev/null
for g in 1 2 4 8; do MI355TTS_X3P_GRID=$g $B > $O/b2_x3p_grid$g.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r3/b2_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    ks=" | ".join(f"{k['kernel'][:34]} {k['avg_launch_us']:.1f}us x{k['launches_per_step']:.0f}" for k in d["roofline"]["kernels"][:5])
    print(f"{f.split('/')[-1]:28s} {d['ms_per_step']:8.2f} ms  {ks}")
PY
