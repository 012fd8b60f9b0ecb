#!/bin/bash
cd /root/repo; export TMPDIR=/tmp
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
for v in nst3 product nst5; do
  if [ $v = product ]; then unset MI355TTS_LIB; else export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_$v.so; fi
  echo "== $v"; ITERS=400 timeout 300 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu
done
