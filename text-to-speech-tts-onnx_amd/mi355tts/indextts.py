"""IndexTTS acoustic GPT-2: host-side mirror of the reference's graphs B, C, D, E and of the per-sentence decode loop.

``IndexGPT`` is what ``ort_session_B/_C/_D/_E`` are in /root/reference IndexTTS/Inference_IndexTTS_ONNX.py:619-675
(graph definitions: IndexTTS/Export_IndexTTS.py:203-289), executed by hand-written gfx950 kernels through the C-ABI:

    text_embed(text_ids)                      graph B   (:723-727)
    mel_embed(gpt_id, gen_len)                graph C   (:729-734, :775-780)
    concat(conds_latent, text_h, mel_h)       graph D   (:736-742)  — a host concatenate, nothing to accelerate
    step(hidden_state, ...)                   graph E   (:754)      — KV cache resident in the handle
    generate(conds_latent, text_ids)          the loop  (:716-783)  — all tokens without leaving the device

There is no CPU fallback: without libmi355tts.so and an MI355X every call raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .config import IndexGPTConfig
from .weights import pack_gpt


class IndexGPT:
    def __init__(self, cfg: IndexGPTConfig, state: Optional[dict] = None, *, blob: Optional[np.ndarray] = None,
                 blob_device=None, dtype: str = "f32", device: int = 0):
        self.cfg = cfg
        self.dtype = dtype
        self.device = device
        self._h = None
        L = _lib.load()
        _lib.init(device)
        self.repeat_penality = np.ones((1, cfg.mel_codes), np.float32)
        if blob_device is not None:          # packed fp32 blob as a CUDA tensor (e.g. filled by an RCCL broadcast)
            import torch
            t = blob_device
            if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.device.index == device):
                raise ValueError("blob_device must be a contiguous float32 CUDA tensor on the engine's device")
            ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
            torch.cuda.current_stream(t.device).synchronize()
            self._h = L.mi_gpt_create_mem(_lib.i32p(ci), len(ci), t.data_ptr(), t.numel(), _lib.DTYPES[dtype], device,
                                          _lib.MI_DEVICE)
            if not self._h:
                raise _lib.MiError("mi_gpt_create_mem: " + L.mi_last_error().decode())
            return
        if blob is None:
            if state is None:
                raise ValueError("IndexGPT needs a state dict or a packed blob")
            blob = pack_gpt(cfg, state)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        ci = np.asarray(cfg.to_int_array(), dtype=np.int32)
        expect = L.mi_gpt_param_count(_lib.i32p(ci), len(ci))
        if expect != blob.size:
            raise _lib.MiError(f"weight blob has {blob.size} floats, config needs {expect}")
        self._h = L.mi_gpt_create(_lib.i32p(ci), len(ci), _lib.f32p(blob), blob.size, _lib.DTYPES[dtype], device)
        if not self._h:
            raise _lib.MiError("mi_gpt_create: " + L.mi_last_error().decode())
        # the reference initialises repeat_penality once and carries it across sentences (:685)
        self.repeat_penality = np.ones((1, cfg.mel_codes), np.float32)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mi_gpt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- graph B ------------------------------------------------------------------------------------------------
    def text_embed(self, text_ids) -> np.ndarray:
        ids = np.ascontiguousarray(np.asarray(text_ids).reshape(-1), dtype=np.int32)
        out = np.empty((1, ids.size + 2, self.cfg.hidden), np.float32)
        _lib.check(_lib.load().mi_gpt_text_embed(self._h, ids.ctypes.data, ids.size, out.ctypes.data, _lib.MI_HOST),
                   "mi_gpt_text_embed")
        return out

    # ---- graph C ------------------------------------------------------------------------------------------------
    def mel_embed(self, gpt_id, gen_len):
        g = int(np.asarray(gen_len).reshape(-1)[0])
        out = np.empty((1, 1, self.cfg.hidden), np.float32)
        _lib.check(_lib.load().mi_gpt_mel_embed(self._h, int(np.asarray(gpt_id).reshape(-1)[0]), g, out.ctypes.data,
                                                _lib.MI_HOST), "mi_gpt_mel_embed")
        return out, np.array([g + 1], np.int64)

    # ---- graph D ------------------------------------------------------------------------------------------------
    @staticmethod
    def concat(embed_x, embed_y, embed_z):
        c = np.ascontiguousarray(np.concatenate([embed_x, embed_y, embed_z], axis=1), dtype=np.float32)
        return c, np.array([c.shape[1]], np.int64)

    # ---- graph E ------------------------------------------------------------------------------------------------
    @property
    def history_len(self) -> int:
        return int(_lib.load().mi_gpt_history_len(self._h))

    def reset(self):
        _lib.check(_lib.load().mi_gpt_reset(self._h), "mi_gpt_reset")

    def step(self, hidden_state, repeat_penality=None, attention_mask: int = 0, return_logits: bool = False):
        """hidden_state (1, ids_len, hidden) appended after the handle's history.  Returns (kv_seq_len,
        last_hidden_state (1, hidden), max_logit_id (1, 1) int32[, logits (1, mel_codes)])."""
        hs = np.ascontiguousarray(hidden_state, dtype=np.float32)
        if hs.ndim != 3 or hs.shape[0] != 1 or hs.shape[2] != self.cfg.hidden or hs.shape[1] < 1:
            raise ValueError(f"hidden_state must be (1, ids_len >= 1, {self.cfg.hidden}), got {hs.shape}")
        pen = None
        if repeat_penality is not None:
            pen = np.ascontiguousarray(repeat_penality, dtype=np.float32).reshape(-1)
            if pen.size != self.cfg.mel_codes:
                raise ValueError(f"repeat_penality must hold {self.cfg.mel_codes} values")
        last = np.empty((1, self.cfg.hidden), np.float32)
        tok = np.empty((1, 1), np.int32)
        logits = np.empty((1, self.cfg.mel_codes), np.float32) if return_logits else None
        _lib.check(_lib.load().mi_gpt_step(self._h, hs.ctypes.data, hs.shape[1], None if pen is None else pen.ctypes.data,
                                           int(attention_mask), last.ctypes.data, tok.ctypes.data,
                                           None if logits is None else logits.ctypes.data, _lib.MI_HOST), "mi_gpt_step")
        kv = np.array([self.history_len], np.int64)
        return (kv, last, tok, logits) if return_logits else (kv, last, tok)

    def kv_read(self, layer: int):
        """(keys (H, 64, history), values (H, history, 64)) of one layer, in the reference's out_key_i / out_value_i
        layouts."""
        c, hist = self.cfg, self.history_len
        k = np.zeros((c.heads, c.head_dim, hist), np.float32)
        v = np.zeros((c.heads, hist, c.head_dim), np.float32)
        if hist:
            _lib.check(_lib.load().mi_gpt_kv_read(self._h, layer, k.ctypes.data, v.ctypes.data, _lib.MI_HOST),
                       "mi_gpt_kv_read")
        return k, v

    def kv_write(self, keys: Sequence[np.ndarray], values: Sequence[np.ndarray]):
        """Load a cache given as the reference's in_key_i / in_value_i lists; history_len := keys[0].shape[2]."""
        c = self.cfg
        if len(keys) != c.layers or len(values) != c.layers:
            raise ValueError(f"expected {c.layers} key and value tensors")
        hist = int(np.asarray(keys[0]).shape[2])
        for i in range(c.layers):
            k = np.ascontiguousarray(keys[i], dtype=np.float32)
            v = np.ascontiguousarray(values[i], dtype=np.float32)
            if k.shape != (c.heads, c.head_dim, hist) or v.shape != (c.heads, hist, c.head_dim):
                raise ValueError(f"layer {i}: keys {k.shape} / values {v.shape} do not match history {hist}")
            _lib.check(_lib.load().mi_gpt_kv_write(self._h, i, k.ctypes.data, v.ctypes.data, hist, _lib.MI_HOST),
                       "mi_gpt_kv_write")

    # ---- the per-sentence loop ------------------------------------------------------------------------------------
    def generate_from_prompt(self, prompt, max_new: int, *, stop_tokens=None, repeat_value=None, penalty_range=None,
                             repeat_penality=None):
        """prompt (1, P, hidden) = graph D's output.  Returns (tokens (n,), hidden (n, hidden), repeat_penality)."""
        c = self.cfg
        p = np.ascontiguousarray(prompt, dtype=np.float32)
        if p.ndim != 3 or p.shape[0] != 1 or p.shape[2] != c.hidden or p.shape[1] < 1:
            raise ValueError(f"prompt must be (1, P >= 1, {c.hidden}), got {p.shape}")
        stops = np.ascontiguousarray([c.stop_mel_token] if stop_tokens is None else list(stop_tokens), dtype=np.int32)
        pen = np.ascontiguousarray(self.repeat_penality if repeat_penality is None else repeat_penality,
                                   dtype=np.float32).reshape(1, -1).copy()
        max_new = int(max_new)
        toks = np.zeros((max(max_new, 1),), np.int32)
        hid = np.zeros((max(max_new, 1), c.hidden), np.float32)
        import ctypes as C
        n = C.c_int32(0)
        _lib.check(_lib.load().mi_gpt_generate(
            self._h, p.ctypes.data, p.shape[1], max_new, stops.ctypes.data if stops.size else None, stops.size,
            float(c.repeat_penalty if repeat_value is None else repeat_value),
            int(c.penalty_range if penalty_range is None else penalty_range), pen.ctypes.data, toks.ctypes.data,
            hid.ctypes.data, C.byref(n), _lib.MI_HOST), "mi_gpt_generate")
        if repeat_penality is None:
            self.repeat_penality = pen
        return toks[: n.value].copy(), hid[: n.value].copy(), pen

    def generate_torch(self, prompt, max_new: int, tokens, hidden, *, stop_tokens=None, repeat_value=None,
                       penalty_range=None, repeat_penality=None) -> int:
        """Device-resident variant: prompt (P, hidden) float32 CUDA tensor; tokens (>= max_new) int32 and hidden
        (>= max_new, hidden) float32 CUDA tensors are filled; repeat_penality (mel_codes) float32 CUDA tensor or None
        (= ones, not written back).  Returns the number of tokens produced."""
        import ctypes as C
        import torch
        c = self.cfg
        assert prompt.is_cuda and prompt.dtype == torch.float32 and prompt.is_contiguous() and prompt.shape[-1] == c.hidden
        assert tokens.is_cuda and tokens.dtype == torch.int32 and tokens.numel() >= max_new
        assert hidden.is_cuda and hidden.dtype == torch.float32 and hidden.is_contiguous() and hidden.shape[0] >= max_new
        stops = [c.stop_mel_token] if stop_tokens is None else list(stop_tokens)
        st = torch.tensor(stops, dtype=torch.int32, device=prompt.device) if stops else None
        torch.cuda.current_stream(prompt.device).synchronize()
        n = C.c_int32(0)
        _lib.check(_lib.load().mi_gpt_generate(
            self._h, prompt.data_ptr(), prompt.shape[-2], int(max_new), st.data_ptr() if st is not None else None,
            len(stops), float(c.repeat_penalty if repeat_value is None else repeat_value),
            int(c.penalty_range if penalty_range is None else penalty_range),
            repeat_penality.data_ptr() if repeat_penality is not None else None, tokens.data_ptr(), hidden.data_ptr(),
            C.byref(n), _lib.MI_DEVICE), "mi_gpt_generate")
        return int(n.value)

    def generate_batch(self, prompts, max_new, *, stop_tokens=None, repeat_value=None, penalty_range=None,
                       repeat_penality=None):
        """Several sentences at once (engine extension: the reference decodes one sentence at a time).  prompts = list
        of (1, P_b, hidden) graph-D outputs, max_new = list of per-sentence limits.  Every decode step streams the
        weights once for all sentences.  Returns a list of (tokens, hidden) and the (nb, mel_codes) penalty matrix."""
        c = self.cfg
        nb = len(prompts)
        if nb < 1 or nb > c.max_batch or len(max_new) != nb:
            raise ValueError(f"batch of {nb} sentences; this engine was created with max_batch = {c.max_batch}")
        ps = [np.ascontiguousarray(p, dtype=np.float32).reshape(-1, c.hidden) for p in prompts]
        rows = np.ascontiguousarray([p.shape[0] for p in ps], dtype=np.int32)
        cat = np.ascontiguousarray(np.concatenate(ps, axis=0))
        mx = np.ascontiguousarray([int(m) for m in max_new], dtype=np.int32)
        cap = max(int(mx.max()), 1)
        stops = np.ascontiguousarray([c.stop_mel_token] if stop_tokens is None else list(stop_tokens), dtype=np.int32)
        pen = (np.ones((nb, c.mel_codes), np.float32) if repeat_penality is None
               else np.ascontiguousarray(repeat_penality, dtype=np.float32).reshape(nb, c.mel_codes).copy())
        toks = np.zeros((nb, cap), np.int32)
        hid = np.zeros((nb, cap, c.hidden), np.float32)
        n = np.zeros((nb,), np.int32)
        _lib.check(_lib.load().mi_gpt_generate_batch(
            self._h, nb, cat.ctypes.data, _lib.i32p(rows), _lib.i32p(mx), stops.ctypes.data if stops.size else None,
            stops.size, float(c.repeat_penalty if repeat_value is None else repeat_value),
            int(c.penalty_range if penalty_range is None else penalty_range), pen.ctypes.data, toks.ctypes.data,
            hid.ctypes.data, cap, _lib.i32p(n), _lib.MI_HOST), "mi_gpt_generate_batch")
        return [(toks[b, : n[b]].copy(), hid[b, : n[b]].copy()) for b in range(nb)], pen

    def generate_batch_torch(self, prompts_cat, prompt_rows, max_new, tokens, hidden, *, stop_tokens=None,
                             repeat_value=None, penalty_range=None):
        """Device-resident variant: prompts_cat (sum P_b, hidden) float32 CUDA tensor; tokens (nb, cap) int32 and hidden
        (nb, cap, hidden) float32 CUDA tensors are filled.  Returns the per-sentence token counts."""
        import torch
        c = self.cfg
        nb = len(prompt_rows)
        assert prompts_cat.is_cuda and prompts_cat.dtype == torch.float32 and prompts_cat.is_contiguous()
        assert tokens.is_cuda and tokens.dtype == torch.int32 and tokens.is_contiguous() and tokens.shape[0] == nb
        assert hidden.is_cuda and hidden.dtype == torch.float32 and hidden.is_contiguous() and hidden.shape[:2] == tokens.shape
        rows = np.ascontiguousarray(prompt_rows, dtype=np.int32)
        mx = np.ascontiguousarray(max_new, dtype=np.int32)
        stops = [c.stop_mel_token] if stop_tokens is None else list(stop_tokens)
        st = torch.tensor(stops, dtype=torch.int32, device=prompts_cat.device) if stops else None
        torch.cuda.current_stream(prompts_cat.device).synchronize()
        n = np.zeros((nb,), np.int32)
        _lib.check(_lib.load().mi_gpt_generate_batch(
            self._h, nb, prompts_cat.data_ptr(), _lib.i32p(rows), _lib.i32p(mx), st.data_ptr() if st is not None else None,
            len(stops), float(c.repeat_penalty if repeat_value is None else repeat_value),
            int(c.penalty_range if penalty_range is None else penalty_range), None, tokens.data_ptr(), hidden.data_ptr(),
            int(tokens.shape[1]), _lib.i32p(n), _lib.MI_DEVICE), "mi_gpt_generate_batch")
        return n

    def generate(self, conds_latent, text_ids, *, max_generate_length=None, **kw):
        """Inference_IndexTTS_ONNX.py:723-783 for one sentence: B, C, D then E until a stop token or
        MAX_GENERATE_LENGTH - concat_len tokens.  Returns (tokens, save_last_hidden_state (n, hidden), penalty)."""
        c = self.cfg
        text_h = self.text_embed(text_ids)
        mel_h, _ = self.mel_embed(c.start_mel_token, 0)
        prompt, concat_len = self.concat(np.asarray(conds_latent, np.float32), text_h, mel_h)
        limit = (c.max_generate_length if max_generate_length is None else int(max_generate_length)) - int(concat_len[0])
        return self.generate_from_prompt(prompt, limit, **kw)


class IndexCond:
    """IndexTTS graph A (ort_session_A of Inference_IndexTTS_ONNX.py:700-712; IndexTTS_A, Export_IndexTTS.py:74-200): int16
    prompt audio -> (conds_latent, vocoder conditioning).  ``state``: upstream-named tensors of weights.cond_spec (folds applied
    here), or ``blob``: the packed array of weights.pack_cond."""

    def __init__(self, cfg, state=None, blob=None, device: int = 0):
        from .config import IndexCondConfig
        from . import weights as W
        assert isinstance(cfg, IndexCondConfig)
        self.cfg = cfg
        _lib.init(device)
        if blob is None:
            if state is None:
                raise ValueError("IndexCond needs a state dict or a packed blob")
            blob = W.pack_cond(cfg, state)
        blob = np.ascontiguousarray(blob, dtype=np.float32)
        arr = np.asarray(cfg.to_int_array(), dtype=np.int32)
        L = _lib.load()
        n = L.mi_indextts_cond_param_count(_lib.i32p(arr), arr.size)
        if n != blob.size:
            raise ValueError(f"IndexCond: blob has {blob.size} weights, the config needs {n}")
        self._h = L.mi_indextts_cond_create(_lib.i32p(arr), arr.size, blob.ctypes.data_as(C.POINTER(C.c_float)), blob.size, device)
        if not self._h:
            raise _lib.MiError("mi_indextts_cond_create: " + L.mi_last_error().decode())
        self.ncond = cfg.voc_initial + sum(cfg.voc_channels)

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mi_indextts_cond_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, audio, return_mel: bool = False):
        """audio int16 (L,) or (1, 1, L) -> (conds (ncond,) = cond_layer | conds_0 | ..., conds_latent (latents, model_dim)[, mel])."""
        a = np.ascontiguousarray(np.asarray(audio).reshape(-1))
        if a.dtype != np.int16:
            raise ValueError("audio must be int16")
        cfg = self.cfg
        conds = np.empty((self.ncond,), np.float32)
        lat = np.empty((cfg.latents, cfg.model_dim), np.float32)
        mel = np.empty((cfg.frames(a.size), cfg.n_mels), np.float32) if return_mel else None
        _lib.check(_lib.load().mi_indextts_cond_run(self._h, a.ctypes.data, a.size, conds.ctypes.data, lat.ctypes.data,
                                                    None if mel is None else mel.ctypes.data, _lib.MI_HOST), "mi_indextts_cond_run")
        return (conds, lat, mel) if return_mel else (conds, lat)

    def split_conds(self, conds):
        """The graph's separate outputs: (save_bigvgan_conds_0..n-1 each (1, C_i, 1), bigvgan_cond_layer_speaker_embedding (1, C0, 1))."""
        cfg = self.cfg
        o = cfg.voc_initial
        outs = []
        for ch in cfg.voc_channels:
            outs.append(conds[o:o + ch].reshape(1, ch, 1)); o += ch
        return outs, conds[:cfg.voc_initial].reshape(1, cfg.voc_initial, 1)
