"""BigVGAN-v2 vocoder: host-side mirror of the reference's BigVGAN graph.

``BigVGANVocoder.run(mel)`` is what ``ort_session_A.run_with_ort_values([generated_wav],
{mel_features: ...})`` does in /root/reference BigVGAN/Export_BigVGAN.py:165-175 — mel
(1, 100, F) float32 in, int16 (1, 1, 256*F + 30) out — executed by hand-written gfx950 kernels
through the C-ABI.  B > 1 is this engine's extension (the reference graph is batch-1).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from . import _lib
from .config import BigVGANConfig
from .weights import pack_bigvgan


class BigVGANVocoder:
    def __init__(self, cfg: BigVGANConfig, state: Optional[dict] = None, *, blob: Optional[np.ndarray] = None,
                 dtype: str = "f32", device: int = 0):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        self._h = None
        L = _lib.load()
        _lib.init(device)
        if blob is None:
            if state is None:
                raise ValueError("BigVGANVocoder needs a state dict or a packed blob")
            blob = pack_bigvgan(cfg, state)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
        expect = L.mi_bigvgan_param_count(_lib.i32p(ci), len(ci))
        if expect != blob.size:
            raise _lib.MiError(f"weight blob has {blob.size} floats, config needs {expect}")
        self._h = L.mi_bigvgan_create(_lib.i32p(ci), len(ci), _lib.f32p(blob), blob.size, _lib.DTYPES[dtype], device)
        if not self._h:
            raise _lib.MiError("mi_bigvgan_create: " + L.mi_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mi_bigvgan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def out_len(self, frames: int) -> int:
        return int(_lib.load().mi_bigvgan_out_len(self._h, frames))

    # ---- numpy in / numpy out (the reference's `.run` shape) ----------------------------------
    def run(self, mel: np.ndarray) -> np.ndarray:
        mel = self._check_mel(mel)
        B, _, F = mel.shape
        out = np.empty((B, 1, self.out_len(F)), dtype=np.int16)
        _lib.check(_lib.load().mi_bigvgan_forward(self._h, mel.ctypes.data, B, F, out.ctypes.data, _lib.MI_HOST),
                   "mi_bigvgan_forward")
        return out

    def run_float(self, mel: np.ndarray) -> np.ndarray:
        """float waveform in [-1, 1] before the int16 conversion (tests)."""
        mel = self._check_mel(mel)
        B, _, F = mel.shape
        out = np.empty((B, 1, self.out_len(F)), dtype=np.float32)
        _lib.check(_lib.load().mi_bigvgan_forward_f32(self._h, mel.ctypes.data, B, F, out.ctypes.data, _lib.MI_HOST),
                   "mi_bigvgan_forward_f32")
        return out

    # ---- torch-ROCm tensors, zero copy (device pointers cross the C-ABI) ----------------------
    def run_torch(self, mel, out=None):
        import torch
        if not (mel.is_cuda and mel.dtype == torch.float32 and mel.is_contiguous()):
            raise ValueError("mel must be a contiguous float32 tensor on the GPU")
        if mel.dim() != 3 or mel.shape[1] != self.cfg.num_mels:
            raise ValueError(f"mel must be (B, {self.cfg.num_mels}, F)")
        B, _, F = mel.shape
        if out is None:
            out = torch.empty((B, 1, self.out_len(F)), dtype=torch.int16, device=mel.device)
        torch.cuda.current_stream(mel.device).synchronize()
        _lib.check(_lib.load().mi_bigvgan_forward(self._h, mel.data_ptr(), B, F, out.data_ptr(), _lib.MI_DEVICE),
                   "mi_bigvgan_forward")
        return out

    def _check_mel(self, mel):
        mel = np.asarray(mel)
        if mel.ndim != 3 or mel.shape[1] != self.cfg.num_mels or mel.shape[2] < 1 or mel.shape[0] < 1:
            raise ValueError(f"mel_features must be (B, {self.cfg.num_mels}, F>=1), got {mel.shape}")
        if mel.dtype == np.float16:
            mel = mel.astype(np.float32)
        if mel.dtype != np.float32:
            raise ValueError(f"mel_features must be float32/float16, got {mel.dtype}")
        return np.ascontiguousarray(mel)


# ---- unit-level ops (tests) --------------------------------------------------------------------
def aa_activation1d(x, alpha_log, beta_log, *, post=False, logscale=True, dtype="f32"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, Cc, T = x.shape
    a = np.ascontiguousarray(alpha_log, dtype=np.float32)
    b = np.ascontiguousarray(beta_log, dtype=np.float32)
    y = np.empty((B, Cc, T + (30 if post else 0)), dtype=np.float32)
    _lib.check(_lib.load().mi_aa_activation1d(_lib.f32p(x), B, Cc, T, _lib.f32p(a), _lib.f32p(b), int(logscale),
                                              int(post), _lib.DTYPES[dtype], _lib.f32p(y)), "mi_aa_activation1d")
    return y


def conv1d(x, w, b=None, *, dilation=1, padding=0, groups=1, dtype="f32"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, Ci, T = x.shape
    Co, _, k = w.shape
    To = T + 2 * padding - dilation * (k - 1)
    y = np.empty((B, Co, To), dtype=np.float32)
    bp = _lib.f32p(np.ascontiguousarray(b, dtype=np.float32)) if b is not None else None
    _lib.check(_lib.load().mi_conv1d(_lib.f32p(x), B, Ci, T, _lib.f32p(w), bp, Co, k, dilation, padding, groups,
                                     _lib.DTYPES[dtype], _lib.f32p(y)), "mi_conv1d")
    return y


def conv_transpose1d(x, w, b=None, *, stride, padding, dtype="f32"):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, Ci, T = x.shape
    _, Co, k = w.shape
    y = np.empty((B, Co, (T - 1) * stride - 2 * padding + k), dtype=np.float32)
    bp = _lib.f32p(np.ascontiguousarray(b, dtype=np.float32)) if b is not None else None
    _lib.check(_lib.load().mi_conv_transpose1d(_lib.f32p(x), B, Ci, T, _lib.f32p(w), bp, Co, k, stride, padding,
                                               _lib.DTYPES[dtype], _lib.f32p(y)), "mi_conv_transpose1d")
    return y
