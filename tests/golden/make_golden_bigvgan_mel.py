#!/usr/bin/env python
"""Fixture of the reference's bigvgan-type mel front end (F5_TTS/modeling_modified/F5/modules.py:30-72,
``get_bigvgan_mel_spectrogram``): run in the build container, where /root/reference is mounted; writes
tests/golden/f5_bigvgan_mel.npz (data only: int16 audio + the reference function's log-mel).

The function is IMPORTED from the reference file where it lies.  Its one un-vendored dependency, ``librosa.filters.mel``, is the
slaney restatement of _ref_import.py, cross-checked here against ``transformers.audio_utils.mel_filter_bank`` (installed).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, ROOT, os.path.join(ROOT, "text-to-speech-tts-onnx_amd")]
import _ref_import as R                      # noqa: E402
from mi355tts import weights as W            # noqa: E402
from mi355tts.config import F5Config         # noqa: E402


def main():
    from transformers.audio_utils import mel_filter_bank
    basis = R.slaney_mel_basis(24000, 1024, 100, 0, None)
    other = mel_filter_bank(513, 100, 0.0, 12000.0, 24000, norm="slaney", mel_scale="slaney").T
    d = float(np.abs(basis - other).max())
    assert basis.shape == other.shape == (100, 513) and d < 1e-6 * float(np.abs(other).max()) + 1e-9, d
    modules = R.load_f5_ref()[0]
    cfg = F5Config()
    out = {"basis_check_max_abs": np.float64(d)}
    audio_syn = W.f5_synthetic_inputs(cfg, 1, 0)[0][0][:24000 + 300]          # 1 s of the bench prompt (+ an odd tail)
    z = np.load(os.path.join(HERE, "zh_prompt.npz"))
    for name, pcm in (("syn", audio_syn), ("zh", z["pcm"])):
        wav = torch.from_numpy(pcm.astype(np.float32) * (1.0 / 32768.0))[None]
        with torch.no_grad():
            mel = modules.get_bigvgan_mel_spectrogram(wav, n_fft=cfg.n_fft, n_mel_channels=cfg.mel_dim, target_sample_rate=cfg.sample_rate,
                                                      hop_length=cfg.hop_length, win_length=cfg.n_fft, fmin=0, fmax=None, center=False)
        out[f"{name}_pcm"] = pcm.astype(np.int16)
        out[f"{name}_logmel"] = mel[0].numpy().astype(np.float32)               # (100, frames)
        print(name, pcm.shape, "->", tuple(mel.shape), float(mel.min()), float(mel.max()))
    np.savez_compressed(os.path.join(HERE, "f5_bigvgan_mel.npz"), **out)


if __name__ == "__main__":
    main()
