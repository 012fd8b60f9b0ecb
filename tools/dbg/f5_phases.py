#!/usr/bin/env python
"""Where the fp32 step's time goes by phase (graph A = preprocess, the 31-evaluation loop, graph C = decode), host-visible time of
each engine call with everything synchronised between them: tools/dbg/f5_phases.py [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts import weights as W
from mi355tts.config import F5Config
from mi355tts.f5 import F5Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
cfg = F5Config()
raw = W.synth_state(W.f5_spec(cfg), 9527, fast=True) if "fast" in W.synth_state.__code__.co_varnames else W.synth_state(W.f5_spec(cfg), 9527)
audio, ids, N, noise = W.f5_synthetic_inputs(cfg, 1, 0)
eng = F5Engine(cfg, raw, dtype="f32")
t = {"preprocess": [], "loop": [], "decode": [], "synthesize": []}
for r in range(reps + 2):
    t0 = time.perf_counter(); o = eng.preprocess(audio.reshape(1, 1, -1), ids.reshape(1, -1), np.array([N]), noise=noise[0])
    t1 = time.perf_counter(); x = eng.sample(o["noise"], o["cat_mel_text"], o["cat_mel_text_drop"])
    t2 = time.perf_counter(); w = eng.decode(x, o["ref_signal_len"])
    t3 = time.perf_counter(); eng.synthesize(audio, ids, N, noise=noise)
    t4 = time.perf_counter()
    if r >= 2:
        for k, v in zip(t, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)): t[k].append(v * 1e3)
for k, v in t.items(): print(f"{k:12s} {np.median(v):8.2f} ms (median of {reps}; host-visible, numpy in / out)")
eng.close()
