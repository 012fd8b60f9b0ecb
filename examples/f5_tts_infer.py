#!/usr/bin/env python
"""Prompt wav + two texts in, WAVEX file out: the F5-TTS driver flow on the MI355X engine.

What the reference's F5_TTS/F5-TTS-ONNX-Inference.py does with three ONNX Runtime sessions (:223-316), written
against this repo's onnxruntime-shaped module so the body reads the same: session A (preprocess), NFE-1 calls of
session B, session C (decode), then the WAVEX write.  With no checkpoint on disk (`--synthetic`, the default) the
weights are the seeded synthetic ones — the output is noise-like audio, but every shape, dtype and call is the real one.

    python examples/f5_tts_infer.py --prompt prompt.wav --ref-text "..." --gen-text "..." --out generated.wav
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))

import mi355tts.ort_compat as onnxruntime          # noqa: E402   (the reference: `import onnxruntime`)
from mi355tts import audio_io, text, weights       # noqa: E402
from mi355tts.config import F5Config                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt", help="reference audio (RIFF/WAVE); default: a synthetic 3 s tone")
    ap.add_argument("--ref-text", default="the quick brown fox jumps over the lazy dog.")
    ap.add_argument("--gen-text", default="pack my box with five dozen liquor jugs.")
    ap.add_argument("--out", default="generated.wav")
    ap.add_argument("--models", help="directory holding F5_Preprocess / F5_Transformer / F5_Decode .mi355.json manifests")
    ap.add_argument("--dtype", default="bf16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--small", action="store_true", help="tiny synthetic model (smoke runs)")
    ap.add_argument("--seed", type=int, default=9527)
    ap.add_argument("--speed", type=float, default=1.0)
    a = ap.parse_args()

    cfg = F5Config.small() if a.small else F5Config()
    tmp = None
    if a.models:
        paths = {k: os.path.join(a.models, f"{k}.mi355.json") for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")}
        vocab = text.load_vocab(os.path.join(a.models, "vocab.txt"))
    else:                                            # synthetic weights -> manifests in a scratch directory
        tmp = tempfile.mkdtemp(prefix="mi355tts_")
        wfile = os.path.join(tmp, "f5_weights.npy")
        np.save(wfile, weights.pack_f5(cfg, weights.synth_state(weights.f5_spec(cfg), a.seed, fast=not a.small)))
        paths = {k: onnxruntime.save_model(os.path.join(tmp, f"{k}.mi355.json"), k, cfg, wfile, a.dtype)
                 for k in ("F5_Preprocess", "F5_Transformer", "F5_Decode")}
        vocab = weights.synth_vocab(cfg.text_num_embeds)

    if a.prompt:
        audio = audio_io.load_prompt(a.prompt, cfg.sample_rate)
    else:
        t = np.arange(3 * cfg.sample_rate)
        audio = (0.1 * 32767 * np.sin(2 * np.pi * 220 * t / cfg.sample_rate)).astype(np.int16).reshape(1, 1, -1)

    onnxruntime.set_seed(a.seed)
    opts = onnxruntime.SessionOptions()
    sess_a = onnxruntime.InferenceSession(paths["F5_Preprocess"], sess_options=opts)
    sess_b = onnxruntime.InferenceSession(paths["F5_Transformer"], sess_options=opts)
    sess_c = onnxruntime.InferenceSession(paths["F5_Decode"], sess_options=opts)
    in_a, out_a = [i.name for i in sess_a.get_inputs()], [o.name for o in sess_a.get_outputs()]
    in_b, out_b = [i.name for i in sess_b.get_inputs()], [o.name for o in sess_b.get_outputs()]
    in_c, out_c = [i.name for i in sess_c.get_inputs()], [o.name for o in sess_c.get_outputs()]

    ids = text.list_str_to_idx(text.convert_char_to_pinyin([a.ref_text + a.gen_text]), vocab)
    max_duration = np.array([text.max_duration(audio.shape[-1], a.ref_text, a.gen_text, cfg.hop_length, a.speed)],
                            dtype=np.int64)

    t0 = time.time()
    noise, cq, sq, ck, sk, cmt, cmtd, rsl = sess_a.run(out_a, {in_a[0]: audio, in_a[1]: ids.astype(np.int32), in_a[2]: max_duration})
    step = np.array([0], dtype=np.int32)
    for _ in range(cfg.nfe_step - 1):
        noise, step = sess_b.run(out_b, {in_b[0]: noise, in_b[1]: cq, in_b[2]: sq, in_b[3]: ck, in_b[4]: sk,
                                         in_b[5]: cmt, in_b[6]: cmtd, in_b[7]: step})
    wav = sess_c.run(out_c, {in_c[0]: noise, in_c[1]: rsl})[0]
    dt = time.time() - t0
    audio_io.write_wavex(a.out, wav.reshape(-1), cfg.sample_rate)
    secs = wav.size / cfg.sample_rate
    print(f"{a.out}: {secs:.2f} s of audio in {dt:.3f} s (RTF {dt / max(secs, 1e-9):.4f}), {cfg.nfe_step - 1} DiT evaluations")
    if tmp:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
