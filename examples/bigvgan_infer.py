#!/usr/bin/env python
"""Mel spectrogram in, WAVEX file out: the BigVGAN-v2 vocoder session on the MI355X engine.

The run at the end of the reference's BigVGAN/Export_BigVGAN.py (:153-175) — `InferenceSession`, `_inputs_meta[0].type / .shape`,
`OrtValue.ortvalue_from_numpy(mel, device_type, DEVICE_ID)`, `run_with_ort_values` — against this repo's onnxruntime-shaped module.
With no checkpoint on disk the weights are the seeded synthetic ones.

    python examples/bigvgan_infer.py [--mel mel.npy] [--out generated.wav] [--dtype f16] [--device-type cuda] [--small]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))

import mi355tts.ort_compat as onnxruntime          # noqa: E402   (the reference: `import onnxruntime`)
from mi355tts import audio_io, weights             # noqa: E402
from mi355tts.config import BigVGANConfig           # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mel", help=".npy holding a log-mel (1, num_mels, frames) float32; default: a synthetic one of 512 frames")
    ap.add_argument("--out", default="generated.wav")
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "bf16"])
    ap.add_argument("--device-type", default="cpu", choices=["cpu", "cuda"])
    ap.add_argument("--small", action="store_true")
    a = ap.parse_args()
    cfg = BigVGANConfig.small() if a.small else BigVGANConfig()
    tmp = tempfile.mkdtemp(prefix="mi355tts_bv_")
    wfile = os.path.join(tmp, "bigvgan.npy")
    np.save(wfile, weights.pack_bigvgan(cfg, weights.synth_state(weights.bigvgan_spec(cfg), 9527, fast=not a.small)))
    onnx_model_A = onnxruntime.save_model(os.path.join(tmp, "BigVGAN.mi355.json"), "BigVGAN", cfg, wfile, a.dtype)
    ort_session_A = onnxruntime.InferenceSession(onnx_model_A, sess_options=onnxruntime.SessionOptions(), providers=[], provider_options=None)
    print(f"Usable Providers: {ort_session_A.get_providers()[0]}")
    in_name_A0, out_name_A0 = ort_session_A.get_inputs()[0].name, ort_session_A.get_outputs()[0].name
    mel = np.load(a.mel).astype(np.float32) if a.mel else weights.bigvgan_synthetic_mel(cfg, 1, 512)
    mel = onnxruntime.OrtValue.ortvalue_from_numpy(mel, a.device_type, 0)
    ort_session_A.run_with_ort_values([out_name_A0], {in_name_A0: mel})                      # first call: allocations
    start_time = time.time()
    output = ort_session_A.run_with_ort_values([out_name_A0], {in_name_A0: mel})
    wav = onnxruntime.OrtValue.numpy(output[0])
    dt = time.time() - start_time
    audio_io.write_wavex(a.out, wav.reshape(-1), cfg.sampling_rate)
    secs = wav.size / cfg.sampling_rate
    print(f"{a.out}: {secs:.2f} s of audio in {dt * 1e3:.2f} ms (RTF {dt / max(secs, 1e-9):.5f})")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
