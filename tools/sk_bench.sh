#!/bin/bash
cd "$(dirname "$0")/.."
SH="f32 2 1126 1024 3072 1 1 f32 2 1126 1024 1024 1 1 f32 2 1126 1024 2048 1 1 f32 2 1126 2048 1024 1 1"
for cfg in "0 0" "1 3" "1 4" "1 2"; do set -- $cfg; echo "== MI355TTS_SK=$1 stages=$2"; MI355TTS_SK=$1 MI355TTS_SK_STAGES=$2 ITERS=50 python tools/gemm_bench.py custom $SH; done
SH16="bf16 2 1126 1024 3072 1 1 bf16 2 1126 1024 1024 1 1 bf16 2 1126 1024 2048 1 1 bf16 2 1126 2048 1024 1 1 bf16 16 1126 1024 3072 1 1 bf16 16 1126 1024 1024 1 1 bf16 16 1126 2048 1024 1 1"
for cfg in "0 0" "2 2" "2 3"; do set -- $cfg; echo "== bf16 MI355TTS_SK=$1 stages=$2"; MI355TTS_SK=$1 MI355TTS_SK_STAGES=$2 ITERS=100 python tools/gemm_bench.py custom $SH16; done
