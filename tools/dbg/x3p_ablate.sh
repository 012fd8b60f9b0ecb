#!/bin/bash
# Ablation of linear_x3p NP = 2 on the four DiT shapes of one utterance (tuning build: build.py --tuning --variant tune):
#   MI355TTS_GEMM_DBG bits: 1 no LDS-DMA, 2 no fragment reads, 8 no MFMA, 16 MFMA on real operands only ; 4 (run time) no fix-up / epilogue
# usage: tools/dbg/x3p_ablate.sh [iters]
cd "$(dirname "$0")/../.."
export MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_tune.so
IT=${1:-40}
for sh in "2 1126 1024 3072" "2 1126 1024 1024" "2 1126 1024 2048" "2 1126 2048 1024"; do
  set -- $sh
  echo "== f32 B$1 T$2 K$3 N$4"
  for d in ${DBGS:-0 4 1 5 2 6 3 7 8 12 9 13 10 14 16 20 36 52 38}; do
    printf "dbg %2d: " $d
    MI355TTS_GEMM_DBG=$d ITERS=$IT timeout 120 python tools/gemm_bench.py custom f32 $1 $2 $3 $4 1 1 | sed 's/custom *//'
  done
done
