#!/bin/bash
mkdir -p gpurun_out/r3
cd /root/repo
export TMPDIR=/tmp
timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_new2.npz 2>&1 | tail -1
python tools/r3/gpt_ab.py --cmp gpurun_out/r3/ab_new.npz gpurun_out/r3/ab_new2.npz
timeout 1500 python -m pytest tests/test_gpu_gpt.py tests/test_gpu_compat.py tests/test_gpu_indextts_a.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --workload indextts --steps 3 --warmup 2 2>/dev/null | tail -1 > gpurun_out/r3/bench_indextts_decode_v2.json; cat gpurun_out/r3/bench_indextts_decode_v2.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['gpt_leg_ms'], d['roofline']['frac'])"
timeout 600 python bench.py --workload indextts --batch 8 --steps 3 --warmup 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('b8', d['ms_per_step'], d['value'])"
