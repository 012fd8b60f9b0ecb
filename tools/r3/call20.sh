#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
for w in 0 2; do echo "== MI355TTS_X2_WIDE=$w"; MI355TTS_X2_WIDE=$w ITERS=400 timeout 300 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu; done
timeout 1500 python -m pytest tests/test_gpu_bigvgan.py -x -q -m gpu -k "panel_planes and wide" 2>&1 | tail -5
