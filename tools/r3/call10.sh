#!/bin/bash
# decode-step kernels A/B: HEAD build (base) vs gemv_d / pipelined attn1
mkdir -p gpurun_out/r3
cd /root/repo
B=text-to-speech-tts-onnx_amd/mi355tts/libmi355tts_base.so
MI355TTS_LIB=$PWD/$B timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_base.npz 2>&1 | tail -3
timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_new.npz 2>&1 | tail -3
python tools/r3/gpt_ab.py --cmp gpurun_out/r3/ab_base.npz gpurun_out/r3/ab_new.npz
AB_DTYPE=f32 AB_TOKENS=64 MI355TTS_LIB=$PWD/$B timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_base32.npz 2>&1 | tail -1
AB_DTYPE=f32 AB_TOKENS=64 timeout 300 python tools/r3/gpt_ab.py gpurun_out/r3/ab_new32.npz 2>&1 | tail -1
python tools/r3/gpt_ab.py --cmp gpurun_out/r3/ab_base32.npz gpurun_out/r3/ab_new32.npz
