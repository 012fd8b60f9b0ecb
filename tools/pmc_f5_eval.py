#!/usr/bin/env python
"""A SHORT command for rocprofv3 --pmc passes (a PMC pass costs ~40 ms per dispatch here: the full bench is 60 k dispatches):
`reps` DiT evaluations of the bench utterance (F5Config(), N = 1126, CFG batch 2 = ~260 launches each) on the HIP engine.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d out -- python tools/pmc_f5_eval.py f32 1 2
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "text-to-speech-tts-onnx_amd")]
import numpy as np                      # noqa: E402
from mi355tts.config import F5Config    # noqa: E402
from mi355tts import weights as W       # noqa: E402
from mi355tts.f5 import F5Engine        # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
U = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cfg = F5Config()
eng = F5Engine(cfg, W.synth_state(W.f5_spec(cfg), 9527, fast=True), dtype=dtype)
N = 1126
noise = np.stack([W.synth_normal_fast(1 + u, "n", (N, cfg.mel_dim)) for u in range(U)])
cmt = np.stack([W.synth_normal_fast(20 + u, "c", (N, cfg.mel_dim + cfg.text_dim), std=0.7) for u in range(U)])
cmtd = np.stack([W.synth_normal_fast(40 + u, "d", (N, cfg.mel_dim + cfg.text_dim), std=0.7) for u in range(U)])
for r in range(reps):
    eng.dit_eval(noise, cmt, cmtd, 3 + r)
eng.close()
print("done", dtype, U, reps)
