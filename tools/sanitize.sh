#!/bin/bash
# Run a command against the sanitizer build of the library (host code under AddressSanitizer + UBSan):
#   python text-to-speech-tts-onnx_amd/build.py --sanitize -j 8
#   tools/sanitize.sh python -m pytest tests/test_capi_symbols.py -q              (CPU: loader, config parsing, error paths)
#   tools/sanitize.sh python -m pytest tests/test_gpu_f5.py -m gpu -q -k small    (GPU box: the engines' host side)
# Leak checking is off (python and the HIP runtime hold memory until exit); protect_shadow_gap=0 is what the ROCm runtime needs.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export MI355TTS_LIB=$ROOT/text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_asan.so
[ -f "$MI355TTS_LIB" ] || { echo "no $MI355TTS_LIB: run build.py --sanitize first" >&2; exit 2; }
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
LD_PRELOAD=$RT exec "$@"
