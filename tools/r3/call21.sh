#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 2048 1024 1 1 f32 6 1126 1024 3072 1 1"
for w in 0 2; do for d in 0 4; do echo "== MI355TTS_X2_WIDE=$w DBG=$d"; MI355TTS_GEMM_DBG=$d MI355TTS_X2_WIDE=$w ITERS=300 timeout 300 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu; done; done
