#!/bin/bash
# package power / sclk while the fp32 QKV linear layer runs in a loop: three bf16 planes, two fp16 planes, and the tuning
# ablations of the two-plane kernel (MFMA only on real operands, data movement only)
cd /root/repo; mkdir -p gpurun_out/r3
SH="f32 2 1126 1024 3072 1 1"
run() {  # label, env...
  local label=$1; shift
  (for i in $(seq 1 6); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Package Power|Socket Power" | tr '\n' ' ' | sed 's/  */ /g'; echo; sleep 1.5; done) > /tmp/pt.txt &
  S=$!
  sleep 0.5
  env "$@" ITERS=150000 timeout 120 python tools/gemm_bench.py custom $SH 2>&1 | grep -v amdgpu > /tmp/pg.txt
  wait $S
  echo "== $label"; cat /tmp/pg.txt; sed -n 3,5p /tmp/pt.txt
}
T=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_tune.so
run "three bf16 planes (product)" MI355TTS_F32_PLANES=3
run "two fp16 planes (product)" MI355TTS_F32_PLANES=2
run "two planes, MFMA only on real operands" MI355TTS_LIB=$T MI355TTS_GEMM_DBG=16
run "two planes, MFMA only on zero operands" MI355TTS_LIB=$T MI355TTS_GEMM_DBG=3
run "two planes, data movement only" MI355TTS_LIB=$T MI355TTS_GEMM_DBG=8
