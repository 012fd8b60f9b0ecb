#!/bin/bash
cd /root/repo; export TMPDIR=/tmp; mkdir -p gpurun_out/r3
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 2048 1024 1 1"
export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_tune.so
{
for dbg in 0 4 16 3 7 8 12 1 2 9 10; do echo "== x2p (tuning build) MI355TTS_GEMM_DBG=$dbg"; MI355TTS_GEMM_DBG=$dbg ITERS=300 timeout 300 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu; done
} > gpurun_out/r3/x2p_ablation1.txt 2>&1
cat gpurun_out/r3/x2p_ablation1.txt
