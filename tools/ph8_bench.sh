#!/bin/bash
# A/B of the 256x256 eight-phase kernel (gemm_ph8.hip) against the 128x128 kernels on the DiT linear layers of 8 utterances
cd "$(dirname "$0")/.."
SH16="bf16 16 1126 1024 3072 1 1 bf16 16 1126 1024 2048 1 1 bf16 16 1126 1024 1024 1 1 bf16 16 1126 2048 1024 1 1 bf16 1 4096 4096 4096 1 1 bf16 1 8192 8192 8192 1 1 f16 16 1126 1024 1024 1 1"
for cfg in "0 1" "1 1" "1 0"; do set -- $cfg; echo "== MI355TTS_PH8=$1 order=$2"; MI355TTS_PH8=$1 MI355TTS_PH8_ORDER=$2 MI355TTS_PH8_MIN=64 ITERS=${ITERS:-50} timeout 300 python tools/gemm_bench.py custom $SH16; done
for dbg in 1 2 3 4 8; do echo "== ablation MI355TTS_GEMM_DBG=$dbg (1 no DMA, 2 no LDS reads, 4 no epilogue, 8 DMA issued out of range)"; MI355TTS_GEMM_DBG=$dbg MI355TTS_PH8=1 MI355TTS_PH8_MIN=64 ITERS=${ITERS:-50} timeout 300 python tools/gemm_bench.py custom bf16 16 1126 1024 3072 1 1 bf16 16 1126 1024 1024 1 1 bf16 1 8192 8192 8192 1 1; done
