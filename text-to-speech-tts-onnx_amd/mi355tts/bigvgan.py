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
                 blob_device=None, dtype: str = "f32", device: int = 0):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        self._h = None
        L = _lib.load()
        _lib.init(device)
        if blob_device is not None:          # packed fp32 blob as a CUDA tensor (e.g. filled by an RCCL broadcast)
            import torch
            t = blob_device
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == device):
                raise ValueError("blob_device must be a contiguous float32 CUDA tensor on the engine's device")
            ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
            torch.cuda.current_stream(t.device).synchronize()
            self._h = L.mi_bigvgan_create_mem(_lib.i32p(ci), len(ci), t.data_ptr(), t.numel(), _lib.DTYPES[dtype], device,
                                              _lib.MI_DEVICE)
            if not self._h:
                raise _lib.MiError("mi_bigvgan_create_mem: " + L.mi_last_error().decode())
            return
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

    # ---- IndexTTS graph F: latent + speaker conditioning in, waveform out ------------------------------------
    def run_latent(self, latent: np.ndarray, conds, return_float: bool = False):
        """latent (T_codes, gpt_dim) float32 ('save_hidden_state'); conds = [save_bigvgan_conds_0..n-1,
        bigvgan_cond_layer_speaker_embedding], each (1, C, 1) / (C,).  Returns int16 (1, 1, (T_codes-2)*hop + 30)."""
        cfg = self.cfg
        if not (cfg.pre_layernorm and cfg.speaker_cond):
            raise ValueError("this vocoder was not created from an IndexTTS graph-F config")
        latent = np.ascontiguousarray(latent, dtype=np.float32)
        if latent.ndim != 2 or latent.shape[1] != cfg.num_mels or latent.shape[0] < 3:
            raise ValueError(f"save_hidden_state must be (T_codes >= 3, {cfg.num_mels}), got {latent.shape}")
        want = [cfg.stage_channels(i) for i in range(cfg.num_upsamples)] + [cfg.upsample_initial_channel]
        if len(conds) != len(want):
            raise ValueError(f"expected {len(want)} conditioning vectors")
        flat = []
        for c, n in zip(conds, want):
            c = np.asarray(c, dtype=np.float32).reshape(-1)
            if c.size != n:
                raise ValueError(f"conditioning vector has {c.size} values, expected {n}")
            flat.append(c)
        flat = np.ascontiguousarray(np.concatenate(flat))
        T = latent.shape[0]
        n = (T - 2) * cfg.hop + 30
        out = np.empty((1, 1, n), np.int16)
        outf = np.empty((1, 1, n), np.float32) if return_float else None
        _lib.check(_lib.load().mi_bigvgan_forward_latent(self._h, latent.ctypes.data, T, flat.ctypes.data, flat.size,
                                                         out.ctypes.data, None if outf is None else outf.ctypes.data,
                                                         _lib.MI_HOST), "mi_bigvgan_forward_latent")
        return (out, outf) if return_float else out

    def run_latent_torch(self, latent, conds_flat, out=None):
        """Device-resident variant: latent (T_codes, gpt_dim) float32 CUDA tensor, conds_flat = the concatenated
        conditioning vectors (float32 CUDA tensor, total_cond values)."""
        import torch
        cfg = self.cfg
        assert latent.is_cuda and latent.dtype == torch.float32 and latent.is_contiguous()
        assert conds_flat.is_cuda and conds_flat.dtype == torch.float32 and conds_flat.is_contiguous()
        T = latent.shape[0]
        if out is None:
            out = torch.empty((1, 1, (T - 2) * cfg.hop + 30), dtype=torch.int16, device=latent.device)
        torch.cuda.current_stream(latent.device).synchronize()
        _lib.check(_lib.load().mi_bigvgan_forward_latent(self._h, latent.data_ptr(), T, conds_flat.data_ptr(),
                                                         conds_flat.numel(), out.data_ptr(), None, _lib.MI_DEVICE),
                   "mi_bigvgan_forward_latent")
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


def aa_conv1d(x, alpha_log, beta_log, w, b, *, dilation=1, res=None, logscale=True, dtype="f32"):
    """The fused kernel of the low-channel stages: conv(Activation1d(x)) (+ res), "same" padding; C % 8 == 0, C <= 96."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    B, Cc, T = x.shape
    if w.shape[0] != Cc or w.shape[1] != Cc:
        raise ValueError("aa_conv1d: weight must be (C, C, k)")
    a = np.ascontiguousarray(alpha_log, dtype=np.float32)
    bb = np.ascontiguousarray(beta_log, dtype=np.float32)
    bias = np.ascontiguousarray(b, dtype=np.float32)
    r = np.ascontiguousarray(res, dtype=np.float32) if res is not None else None
    y = np.empty_like(x)
    _lib.check(_lib.load().mi_aa_conv1d(_lib.f32p(x), B, Cc, T, _lib.f32p(a), _lib.f32p(bb), int(logscale), _lib.f32p(w),
                                        _lib.f32p(bias), w.shape[2], dilation, _lib.f32p(r) if r is not None else None,
                                        _lib.DTYPES[dtype], _lib.f32p(y)), "mi_aa_conv1d")
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
