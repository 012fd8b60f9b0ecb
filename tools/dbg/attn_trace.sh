#!/bin/bash
# Where one wave of the fp32 attention kernel spends its cycles: s_memtime stamps at the phase boundaries (build.py with
# MI355TTS_EXTRA_FLAGS=-DMI355TTS_ATTN_TRACE --variant atrace), one line per launch from workgroup (3, 5), four waves.
#   tools/dbg/attn_trace.sh [utterances]      (1: 128-query workgroups x 3 key slices ; 4: 1152 unsliced workgroups)
MI355TTS_LIB=$PWD/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_atrace.so python tools/pmc_f5_eval.py f32 ${1:-1} 1 2>&1 | grep ATTN_TRACE | head -${2:-12}
