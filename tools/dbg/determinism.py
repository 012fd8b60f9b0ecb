"""Run-to-run identity of the full-size F5 waveform (graph replays included): stream-K fix-ups, split-tail slabs, attention
tickets and key slices must not make the result depend on timing."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts.config import F5Config
from mi355tts import weights as W
from mi355tts.f5 import F5Engine
cfg = F5Config()
raw = W.synth_state(W.f5_spec(cfg), 9527)
for dtype, U, reps in (("f32", 1, 6), ("bf16", 8, 4), ("f32", 3, 3)):
    eng = F5Engine(cfg, raw, dtype=dtype)
    audio, ids, N, noise = W.f5_synthetic_inputs(cfg, U, 0)
    ref = eng.synthesize(audio, ids, N, noise=noise)
    same = all(np.array_equal(ref, eng.synthesize(audio, ids, N, noise=noise)) for _ in range(reps))
    print(dtype, U, "identical over", reps + 1, "runs:", same, "rms", float(np.sqrt(np.mean(ref.astype(np.float64) ** 2))), flush=True)
    eng.close()
