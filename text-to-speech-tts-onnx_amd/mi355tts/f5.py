"""F5-TTS engine: host-side mirror of the reference's three F5 graphs.

    F5Engine.preprocess        <-> ort_session_A.run   F5_TTS/F5-TTS-ONNX-Inference.py:247-253
    F5Engine.transformer_step  <-> ort_session_B.run   :292-303  (one call of the NFE loop)
    F5Engine.sample            <-> the whole loop      :291-304  (kept on the device)
    F5Engine.decode            <-> ort_session_C.run   :306-311
    F5Engine.synthesize        <-> lines :247-311 end to end, no host round trips

numpy in / numpy out with the ONNX tensor layouts.  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from . import _lib
from .config import F5Config
from .weights import pack_f5


def _p(a):
    return None if a is None else a.ctypes.data


class F5Engine:
    def __init__(self, cfg: F5Config, state: Optional[dict] = None, *, blob: Optional[np.ndarray] = None,
                 blob_device=None, dtype: str = "f32", device: int = 0):
        """`state`: upstream-named (unfolded) tensors | `blob`: packed fp32 numpy blob | `blob_device`: the same blob as a
        float32 CUDA tensor on `device` (e.g. the buffer torch.distributed.broadcast filled) — consumed in place."""
        self.cfg, self.dtype, self.device, self._h = cfg, dtype, device, None
        L = _lib.load()
        _lib.init(device)
        self._ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
        self._cf = np.asarray(cfg.to_float_array(), dtype=np.float32)
        if blob_device is not None:
            import torch
            t = blob_device
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == device):
                raise ValueError("blob_device must be a contiguous float32 CUDA tensor on the engine's device")
            expect = L.mi_f5_param_count(_lib.i32p(self._ci), len(self._ci), _lib.f32p(self._cf), len(self._cf))
            if expect != t.numel():
                raise _lib.MiError(f"weight blob has {t.numel()} floats, config needs {expect}")
            torch.cuda.current_stream(t.device).synchronize()
            self._h = L.mi_f5_create_mem(_lib.i32p(self._ci), len(self._ci), _lib.f32p(self._cf), len(self._cf),
                                         t.data_ptr(), t.numel(), _lib.DTYPES[dtype], device, _lib.MI_DEVICE)
            if not self._h:
                raise _lib.MiError("mi_f5_create_mem: " + L.mi_last_error().decode())
            return
        if blob is None:
            if state is None:
                raise ValueError("F5Engine needs an (unfolded, upstream-named) state dict or a packed blob")
            blob = pack_f5(cfg, state)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        self._ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
        self._cf = np.asarray(cfg.to_float_array(), dtype=np.float32)
        expect = L.mi_f5_param_count(_lib.i32p(self._ci), len(self._ci), _lib.f32p(self._cf), len(self._cf))
        if expect != blob.size:
            raise _lib.MiError(f"weight blob has {blob.size} floats, config needs {expect}")
        self._h = L.mi_f5_create(_lib.i32p(self._ci), len(self._ci), _lib.f32p(self._cf), len(self._cf),
                                 _lib.f32p(blob), blob.size, _lib.DTYPES[dtype], device)
        if not self._h:
            raise _lib.MiError("mi_f5_create: " + L.mi_last_error().decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mi_f5_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> dict:
        """What the engine is doing (mi_f5_info): the fp32 arithmetic IN USE (an fp16-pair engine whose weights or activations
        leave the fp16 range switches itself to the exact three-plane bf16 split), whether that happened during a call, and
        whether the AdaLN fold is available."""
        from .config import F32_ARITHMETIC_NAMES
        L = _lib.load()
        a = int(L.mi_f5_info(self._h, b"f32_arithmetic"))
        return {"f32_arithmetic": F32_ARITHMETIC_NAMES.get(a), "saturation_events": int(L.mi_f5_info(self._h, b"saturation_events")),
                "adaln_fold": bool(L.mi_f5_info(self._h, b"adaln_fold"))}

    # ---- load-time tables -----------------------------------------------------------------------
    def tables(self):
        te = np.empty((self.cfg.nfe_step, self.cfg.dim), np.float32)
        dt = np.empty((self.cfg.nfe_step - 1,), np.float32)
        _lib.check(_lib.load().mi_f5_tables(self._h, _lib.f32p(te), _lib.f32p(dt)), "mi_f5_tables")
        return te, dt

    # ---- graph A ----------------------------------------------------------------------------------
    def preprocess(self, audio, text_ids, max_duration, noise=None, seed: int = 9527):
        """audio (1,1,L) / (L,) int16 ; text_ids (1,T) / (T,) int32 ; max_duration int / (1,) int64.
        Returns the reference's 8 outputs as a dict (names = ONNX output names)."""
        cfg = self.cfg
        audio = np.ascontiguousarray(np.asarray(audio).reshape(-1))
        if audio.dtype != np.int16:
            raise ValueError("audio must be int16")
        text_ids = np.ascontiguousarray(np.asarray(text_ids).reshape(-1), dtype=np.int32)
        N = int(np.asarray(max_duration).reshape(-1)[0])
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32).reshape(N, cfg.mel_dim)
        out_noise = np.empty((1, N, cfg.mel_dim), np.float32)
        rc = np.empty((N, cfg.dim_head), np.float32)
        rs = np.empty((N, cfg.dim_head), np.float32)
        cd = cfg.mel_dim + cfg.text_dim
        cmt = np.empty((1, N, cd), np.float32)
        cmtd = np.empty((1, N, cd), np.float32)
        R = C.c_int64(0)
        _lib.check(_lib.load().mi_f5_preprocess(self._h, audio.ctypes.data, audio.size, text_ids.ctypes.data, text_ids.size,
                                                N, _p(noise), seed, out_noise.ctypes.data, rc.ctypes.data, rs.ctypes.data,
                                                cmt.ctypes.data, cmtd.ctypes.data, C.byref(R), _lib.MI_HOST),
                   "mi_f5_preprocess")
        H = cfg.heads
        cq = np.broadcast_to(rc[None, None], (2, H, N, cfg.dim_head))
        sq = np.broadcast_to(rs[None, None], (2, H, N, cfg.dim_head))
        return {"noise": out_noise, "rope_cos_q": cq, "rope_sin_q": sq, "rope_cos_k": cq.transpose(0, 1, 3, 2),
                "rope_sin_k": sq.transpose(0, 1, 3, 2), "cat_mel_text": cmt, "cat_mel_text_drop": cmtd,
                "ref_signal_len": np.int64(R.value)}

    def stft(self, audio):
        """STFT-B of graph A alone (STFT_Process.py:144-157): audio (L,) int16 -> (real, imag), each (n_fft/2+1, L//hop+1)."""
        cfg = self.cfg
        audio = np.ascontiguousarray(np.asarray(audio).reshape(-1))
        if audio.dtype != np.int16:
            raise ValueError("audio must be int16")
        nb, R = cfg.n_fft // 2 + 1, audio.size // cfg.hop_length + 1
        spec = np.empty((R, 2 * nb), np.float32)
        _lib.check(_lib.load().mi_f5_stft(self._h, audio.ctypes.data, audio.size, spec.ctypes.data, _lib.MI_HOST), "mi_f5_stft")
        return np.ascontiguousarray(spec[:, :nb].T), np.ascontiguousarray(spec[:, nb:].T)

    # ---- graph B ----------------------------------------------------------------------------------
    def _cond(self, noise, cmt, cmtd):
        cfg = self.cfg
        noise = np.ascontiguousarray(noise, dtype=np.float32)
        if noise.ndim != 3 or noise.shape[2] != cfg.mel_dim:
            raise ValueError(f"noise must be (U, N, {cfg.mel_dim})")
        U, N, _ = noise.shape
        cd = cfg.mel_dim + cfg.text_dim
        cmt = np.ascontiguousarray(cmt, dtype=np.float32)
        cmtd = np.ascontiguousarray(cmtd, dtype=np.float32)
        if cmt.shape != (U, N, cd) or cmtd.shape != (U, N, cd):
            raise ValueError(f"cat_mel_text(_drop) must be ({U}, {N}, {cd})")
        return noise, cmt, cmtd, U, N

    def transformer_step(self, noise, cat_mel_text, cat_mel_text_drop, time_step, fuse: int = 1):
        """One ort_session_B.run: returns (denoised, time_step + fuse); inputs are not modified."""
        noise, cmt, cmtd, U, N = self._cond(noise, cat_mel_text, cat_mel_text_drop)
        x = noise.copy()
        ts = np.asarray(time_step, dtype=np.int32).reshape(-1).copy()
        _lib.check(_lib.load().mi_f5_transformer_step(self._h, x.ctypes.data, cmt.ctypes.data, cmtd.ctypes.data, U, N,
                                                      _lib.i32p(ts), fuse, _lib.MI_HOST), "mi_f5_transformer_step")
        return x, ts

    def transformer_step_device(self, x, cat_mel_text, cat_mel_text_drop, k: int, fuse: int = 1) -> int:
        """ort_session_B.run_with_iobinding with every operand in HBM: float32 CUDA tensors x (U,N,100) — advanced IN PLACE —,
        cat_mel_text(_drop) (U,N,612); `k` = the grid index *time_step holds.  Returns k + fuse."""
        import torch
        for t in (x, cat_mel_text, cat_mel_text_drop):
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == self.device):
                raise ValueError("operands must be contiguous float32 CUDA tensors on the engine's device")
        U, N, M = x.shape
        cd = self.cfg.mel_dim + self.cfg.text_dim
        if M != self.cfg.mel_dim or tuple(cat_mel_text.shape) != (U, N, cd) or tuple(cat_mel_text_drop.shape) != (U, N, cd):
            raise ValueError(f"x must be (U, N, {self.cfg.mel_dim}), cat_mel_text(_drop) (U, N, {cd})")
        torch.cuda.current_stream(x.device).synchronize()
        ts = C.c_int32(int(k))
        _lib.check(_lib.load().mi_f5_transformer_step(self._h, x.data_ptr(), cat_mel_text.data_ptr(), cat_mel_text_drop.data_ptr(),
                                                      U, N, C.byref(ts), int(fuse), _lib.MI_DEVICE), "mi_f5_transformer_step")
        return int(ts.value)

    def sample(self, noise, cat_mel_text, cat_mel_text_drop, k0: int = 0, n_steps: Optional[int] = None):
        noise, cmt, cmtd, U, N = self._cond(noise, cat_mel_text, cat_mel_text_drop)
        x = noise.copy()
        if n_steps is None:
            n_steps = self.cfg.nfe_step - 1 - k0
        _lib.check(_lib.load().mi_f5_sample(self._h, x.ctypes.data, cmt.ctypes.data, cmtd.ctypes.data, U, N, k0, n_steps,
                                            _lib.MI_HOST), "mi_f5_sample")
        return x

    def dit_eval(self, noise, cat_mel_text, cat_mel_text_drop, k: int):
        noise, cmt, cmtd, U, N = self._cond(noise, cat_mel_text, cat_mel_text_drop)
        pred = np.empty((2 * U, N, self.cfg.mel_dim), np.float32)
        _lib.check(_lib.load().mi_f5_dit_eval(self._h, noise.ctypes.data, cmt.ctypes.data, cmtd.ctypes.data, U, N, k,
                                              pred.ctypes.data, _lib.MI_HOST), "mi_f5_dit_eval")
        return pred

    # ---- graph C ----------------------------------------------------------------------------------
    def decode(self, denoised, ref_signal_len, return_float: bool = False):
        cfg = self.cfg
        den = np.ascontiguousarray(denoised, dtype=np.float32)
        if den.ndim != 3 or den.shape[2] != cfg.mel_dim:
            raise ValueError(f"denoised must be (U, N, {cfg.mel_dim})")
        U, N, _ = den.shape
        R = int(ref_signal_len)
        n = (N - R - 1) * cfg.hop_length
        if n < 0:
            raise ValueError("decode needs at least one generated frame")
        if n == 0:          # one generated frame: the reference's graph C returns an empty waveform ((N - R - 1) * hop samples)
            out0 = np.empty((U, 1, 0), np.int16)
            return (out0, np.empty((U, 1, 0), np.float32)) if return_float else out0
        out = np.empty((U, 1, n), np.int16)
        outf = np.empty((U, 1, n), np.float32) if return_float else None
        ln = C.c_int64(0)
        _lib.check(_lib.load().mi_f5_decode(self._h, den.ctypes.data, U, N, R, out.ctypes.data, _p(outf), C.byref(ln),
                                            _lib.MI_HOST), "mi_f5_decode")
        assert ln.value == n
        return (out, outf) if return_float else out

    # ---- A -> loop -> C ----------------------------------------------------------------------------------
    def synthesize(self, audio, text_ids, max_duration, noise=None, seed: int = 9527):
        """audio (U,L) int16, text_ids (U,T) int32 -> int16 (U, 1, (N-R-1)*hop)."""
        cfg = self.cfg
        audio = np.ascontiguousarray(np.atleast_2d(np.asarray(audio)))
        if audio.dtype != np.int16:
            raise ValueError("audio must be int16")
        text_ids = np.ascontiguousarray(np.atleast_2d(np.asarray(text_ids)), dtype=np.int32)
        U, Ln = audio.shape
        if text_ids.shape[0] != U:
            raise ValueError("audio / text_ids batch mismatch")
        N = int(max_duration)
        R = cfg.ref_frames(Ln)
        n = (N - R - 1) * cfg.hop_length
        if n < 0:
            raise ValueError("max_duration leaves no generated frames")
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32).reshape(U, N, cfg.mel_dim)
        out = np.empty((U, 1, max(n, 0)), np.int16)
        ln = C.c_int64(0)
        _lib.check(_lib.load().mi_f5_synthesize(self._h, U, audio.ctypes.data, Ln, text_ids.ctypes.data, text_ids.shape[1],
                                                N, _p(noise), seed, out.ctypes.data, C.byref(ln), _lib.MI_HOST),
                   "mi_f5_synthesize")
        assert ln.value == n
        return out

    def synthesize_mel(self, audio, text_ids, max_duration, noise=None, seed: int = 9527):
        """A -> loop, generated frames as a vocoder mel: float32 (U, 100, N - R), channels first — what
        ``BigVGANVocoder.run`` takes (the "F5-TTS + BigVGAN" pipeline; see include/mi355tts.h mi_f5_synthesize_mel)."""
        cfg = self.cfg
        audio = np.ascontiguousarray(np.atleast_2d(np.asarray(audio)))
        if audio.dtype != np.int16:
            raise ValueError("audio must be int16")
        text_ids = np.ascontiguousarray(np.atleast_2d(np.asarray(text_ids)), dtype=np.int32)
        U, Ln = audio.shape
        if text_ids.shape[0] != U:
            raise ValueError("audio / text_ids batch mismatch")
        N = int(max_duration)
        F = N - cfg.ref_frames(Ln)
        if F < 1:
            raise ValueError("max_duration leaves no generated frames")
        if noise is not None:
            noise = np.ascontiguousarray(noise, dtype=np.float32).reshape(U, N, cfg.mel_dim)
        mel = np.empty((U, cfg.mel_dim, F), np.float32)
        nf = C.c_int64(0)
        _lib.check(_lib.load().mi_f5_synthesize_mel(self._h, U, audio.ctypes.data, Ln, text_ids.ctypes.data, text_ids.shape[1],
                                                    N, _p(noise), seed, mel.ctypes.data, C.byref(nf), _lib.MI_HOST),
                   "mi_f5_synthesize_mel")
        assert nf.value == F
        return mel

    def synthesize_mel_torch(self, audio, text_ids, max_duration, noise=None, seed: int = 9527, out=None):
        """Device-resident variant of synthesize_mel: CUDA tensors in, float32 (U, 100, N - R) CUDA tensor out."""
        import torch
        cfg = self.cfg
        U, Ln = audio.shape
        N = int(max_duration)
        F = N - cfg.ref_frames(Ln)
        if out is None:
            out = torch.empty((U, cfg.mel_dim, F), dtype=torch.float32, device=audio.device)
        assert audio.is_cuda and audio.dtype == torch.int16 and audio.is_contiguous()
        assert text_ids.is_cuda and text_ids.dtype == torch.int32 and text_ids.is_contiguous()
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and tuple(out.shape) == (U, cfg.mel_dim, F)
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
        torch.cuda.current_stream(audio.device).synchronize()
        nf = C.c_int64(0)
        _lib.check(_lib.load().mi_f5_synthesize_mel(self._h, U, audio.data_ptr(), Ln, text_ids.data_ptr(), text_ids.shape[1], N,
                                                    None if noise is None else noise.data_ptr(), seed, out.data_ptr(),
                                                    C.byref(nf), _lib.MI_DEVICE), "mi_f5_synthesize_mel")
        return out

    def synthesize_torch(self, audio, text_ids, max_duration, noise=None, seed: int = 9527, out=None):
        """Device-resident variant: torch int16 (U,L) / int32 (U,T) / float32 (U,N,100) CUDA tensors."""
        import torch
        cfg = self.cfg
        U, Ln = audio.shape
        N = int(max_duration)
        R = cfg.ref_frames(Ln)
        n = (N - R - 1) * cfg.hop_length
        if n < 0:
            raise ValueError("max_duration leaves no generated frames")
        if out is None:
            out = torch.empty((U, 1, n), dtype=torch.int16, device=audio.device)
        assert audio.is_cuda and audio.dtype == torch.int16 and audio.is_contiguous()
        assert text_ids.is_cuda and text_ids.dtype == torch.int32 and text_ids.is_contiguous()
        assert out.is_cuda and out.dtype == torch.int16 and out.is_contiguous() and out.numel() == U * n
        if noise is not None:
            assert noise.is_cuda and noise.dtype == torch.float32 and noise.is_contiguous()
        torch.cuda.current_stream(audio.device).synchronize()
        ln = C.c_int64(0)
        # one generated frame -> an empty waveform like graph C (and like synthesize()): an empty tensor has no storage, so the
        # library (which requires a destination) gets a one-sample scratch buffer it writes nothing to
        dst = out if n > 0 else torch.empty((1,), dtype=torch.int16, device=audio.device)
        _lib.check(_lib.load().mi_f5_synthesize(self._h, U, audio.data_ptr(), Ln, text_ids.data_ptr(), text_ids.shape[1], N,
                                                None if noise is None else noise.data_ptr(), seed, dst.data_ptr(),
                                                C.byref(ln), _lib.MI_DEVICE), "mi_f5_synthesize")
        assert ln.value == n
        return out
