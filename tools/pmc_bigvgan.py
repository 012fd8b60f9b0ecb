#!/usr/bin/env python
"""A SHORT command for rocprofv3 --pmc passes over the vocoder (the last measured traffic of the BigVGAN
forward dated from round 1): `reps` forwards of BigVGAN-v2 24khz_100band_256x on a mel (B, 100, 512), one stream (side streams
would interleave the dispatches of the three AMP blocks; the bytes are the same).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- python tools/pmc_bigvgan.py f16 8 1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "text-to-speech-tts-onnx_amd")]
from mi355tts import _lib                        # noqa: E402
from mi355tts.config import BigVGANConfig        # noqa: E402
from mi355tts import weights as W                # noqa: E402
from mi355tts.bigvgan import BigVGANVocoder      # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = BigVGANConfig()
voc = BigVGANVocoder(cfg, W.synth_state(W.bigvgan_spec(cfg), 9527, fast=True), dtype=dtype)
_lib.set_option("bigvgan_streams", 1)
mel = W.bigvgan_synthetic_mel(cfg, B, 512, 0)
for _ in range(reps):
    voc.run(mel)
voc.close()
print("done", dtype, B, reps)
