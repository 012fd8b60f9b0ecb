#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_tail.so
for d in 0 256 0 256; do echo "== dbg=$d (4: no fix-up no epilogue; 64: no fix-up; 128: no epilogue)"; MI355TTS_GEMM_DBG=$d ITERS=300 timeout 300 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu; done
