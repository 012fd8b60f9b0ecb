"""TEST INFRASTRUCTURE — CPU oracle (numpy restatement) of the reference F5-TTS graphs.

Checker only: imported by tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg;
never by the product path.

Restates (paths relative to /root/reference):
  F5Preprocess.forward      F5_TTS/Export_F5.py:117-141   (+ tables :99-115)
  F5Transformer.forward     F5_TTS/Export_F5.py:167-182   (+ tables :145-165)
  F5Decode.forward          F5_TTS/Export_F5.py:193-203
  TextEmbedding / DiT       F5_TTS/modeling_modified/F5/dit.py:49-73, 85-87, 205-220
  DiTBlock / attention      F5_TTS/modeling_modified/F5/modules.py:217-261, 301-340, 421-468, 599-613
  Vocos backbone / head     F5_TTS/modeling_modified/vocos/models.py:78-83, modules.py:43-51, heads.py:55-59
  STFT_Process              F5_TTS/STFT_Process.py:86-133, 144-166

Parity pin: validated against the reference's own module code run in the build container
(tests/golden/make_golden_f5.py -> tests/golden/f5_small.npz), see tests/test_oracle_f5.py.
Un-vendored formula with no in-tree source: torchaudio.functional.melscale_fbanks (HTK, norm=None)
— restated from its published definition, cross-checked against
transformers.audio_utils.mel_filter_bank in the golden generator ("parity unpinned" for that table).

Weights: a dict with the *folded* tensors of mi355tts.weights.fold_f5 (q/k pre-scaled, Vocos norm
weights * sqrt(C), gamma folded into pwconv2), i.e. what the exported graphs hold.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.special import erf as _erf

F32 = np.float32

# ------------------------------------------------------------------------------------------------
# threads (bench.py's cpu_baseline leg): numpy runs its element-wise passes and the per-head attention loop on ONE core,
# which made the timed baseline 5x slower than the reference's own torch modules on the same 8 cores.  With
# set_threads(n > 1) the independent pieces — rows of an element-wise pass, (branch, head) pairs of attention — are spread
# over n Python threads (numpy releases the GIL inside ufuncs and BLAS); BLAS itself is limited to one thread inside
# the head loop and to n threads elsewhere.  Each piece is the single-threaded arithmetic; only OpenBLAS's own
# blocking changes with its thread count, so results for different n agree to fp32 rounding, not bit for bit
# (tests/test_oracle_f5.py checks both settings against the reference fixture).
# ------------------------------------------------------------------------------------------------
_THREADS = 1
_POOL = None


def set_threads(n: int) -> int:
    """Number of worker threads for the row-parallel passes and the attention head loop (1 = plain numpy)."""
    global _THREADS, _POOL
    n = max(1, int(n))
    if _POOL is not None:
        _POOL.shutdown(wait=True)
        _POOL = None
    _THREADS = n
    if n > 1:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=n)
    return n


def get_threads() -> int:
    return _THREADS


def _blas_limit(n):
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=n, user_api="blas")
    except Exception:                                  # threadpoolctl missing: BLAS keeps its own setting
        import contextlib
        return contextlib.nullcontext()


def _rows(fn, x):
    """fn applied to row blocks of x (all leading axes flattened) on the worker threads; fn must act row by row."""
    if _POOL is None or x.ndim < 2 or x.shape[0] * int(np.prod(x.shape[1:-1], dtype=np.int64)) < 4 * _THREADS:
        return fn(x)
    flat = x.reshape(-1, x.shape[-1])
    n = flat.shape[0]
    step = -(-n // (_THREADS * 2))
    parts = list(_POOL.map(lambda i: fn(flat[i:i + step]), range(0, n, step)))
    return np.concatenate(parts, axis=0).reshape(x.shape[:-1] + parts[0].shape[-1:])


# ------------------------------------------------------------------------------------------------
# small math
# ------------------------------------------------------------------------------------------------
def gelu_erf(x):
    return (F32(0.5) * x * (F32(1.0) + _erf(x * F32(0.7071067811865476)).astype(F32))).astype(F32)


def _gelu_tanh1(x):
    k0, k1 = F32(0.7978845608028654), F32(0.044715)
    return (F32(0.5) * x * (F32(1.0) + np.tanh(k0 * (x + k1 * x * x * x)))).astype(F32)


def gelu_tanh(x):
    return _rows(_gelu_tanh1, x)


def silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def mish(x):
    sp = np.where(x > 20.0, x, np.log1p(np.exp(np.minimum(x, 20.0))))
    return (x * np.tanh(sp)).astype(F32)


def _layer_norm1(x, eps=1e-6):
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    return ((x - mu) / np.sqrt(var + eps)).astype(F32)


def layer_norm(x, eps=1e-6):
    """LayerNorm over the last axis, no affine, biased variance."""
    return _rows(lambda r: _layer_norm1(r, eps), x)


def linear(x, w, b=None):
    y = x @ w.T
    if b is not None:
        y += b
    return y if y.dtype == F32 else y.astype(F32)


# ------------------------------------------------------------------------------------------------
# STFT / ISTFT — STFT_Process.py
# ------------------------------------------------------------------------------------------------
def hann_periodic(n):
    """torch.hann_window(n) (periodic), evaluated in float32 like torch: cos(k * fl32(2pi/n)) * -0.5 + 0.5."""
    ang = (np.arange(n, dtype=F32) * F32(2.0 * math.pi / n)).astype(F32)
    return (np.cos(ang).astype(F32) * F32(-0.5) + F32(0.5)).astype(F32)


def stft_kernels(n_fft=1024):
    """cos_kernel / sin_kernel of STFT_Process.__init__ (:86-98), evaluated in fp32 like torch does:
    omega = 2*pi*f*t/n_fft in float32, then cos/sin of the rounded argument."""
    t = np.arange(n_fft, dtype=F32)[None, :]
    f = np.arange(n_fft // 2 + 1, dtype=F32)[:, None]
    omega = ((F32(2.0 * math.pi) * f) * t) / F32(n_fft)
    w = hann_periodic(n_fft)[None, :]
    return (np.cos(omega) * w).astype(F32), (-np.sin(omega) * w).astype(F32)


def stft_b(audio_f32, n_fft=1024, hop=256):
    """audio (L,) float32 -> real, imag each (n_fft/2+1, L//hop + 1); reflect padding (:144-157)."""
    half = n_fft // 2
    xp = np.pad(audio_f32.astype(F32), (half, half), mode="reflect")
    nfr = (len(xp) - n_fft) // hop + 1
    idx = np.arange(nfr)[:, None] * hop + np.arange(n_fft)[None, :]
    frames = xp[idx]                                              # (F, n_fft)
    ck, sk = stft_kernels(n_fft)
    return (ck @ frames.T).astype(F32), (sk @ frames.T).astype(F32)


def istft_basis(n_fft=1024, hop=256):
    """inverse_basis (n_fft+2, n_fft) = window * pinv(fourier_basis * n_fft / hop).T (:100-112),
    in closed form: pinv of the one-sided real DFT basis is the irfft weight table."""
    n = np.arange(n_fft, dtype=np.float64)[None, :]
    k = np.arange(n_fft // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n / n_fft
    scale = np.full((n_fft // 2 + 1, 1), 2.0 / n_fft)
    scale[0, 0] = scale[-1, 0] = 1.0 / n_fft
    w = hann_periodic(n_fft).astype(np.float64)[None, :]
    cosb = w * (hop / n_fft) * scale * np.cos(ang)
    sinb = w * (hop / n_fft) * scale * (-np.sin(ang))
    return np.concatenate([cosb, sinb], axis=0).astype(F32)


def window_sum_inv(n_fft=1024, hop=256, max_frames=4096):
    n = n_fft + hop * (max_frames - 1)
    ws = np.zeros(n, dtype=F32)
    w = hann_periodic(n_fft)
    win_sq = (w / np.abs(w).max()) ** 2
    for i in range(max_frames):
        ws[i * hop:i * hop + n_fft] += win_sq
    return (F32(n_fft) / (ws * F32(hop) + F32(1e-7))).astype(F32)


def istft_a(mag, phase, n_fft=1024, hop=256, max_frames=4096, _cache={}):
    """mag, phase (n_fft/2+1, F) -> (F-1)*hop samples (:160-166)."""
    key = (n_fft, hop, max_frames)
    if key not in _cache:
        _cache[key] = (istft_basis(n_fft, hop), window_sum_inv(n_fft, hop, max_frames))
    basis, wsi = _cache[key]
    inp = np.concatenate([mag * np.cos(phase), mag * np.sin(phase)], axis=0).astype(F32)   # (n_fft+2, F)
    Fr = inp.shape[1]
    fr = (inp.T @ basis).astype(F32)                                # (F, n_fft)
    out = np.zeros((Fr - 1) * hop + n_fft, dtype=F32)
    for f in range(Fr):
        out[f * hop:f * hop + n_fft] += fr[f]
    s, e = n_fft // 2, len(out) - n_fft // 2
    return (out[s:e] * wsi[s:e]).astype(F32)


def melscale_fbanks_htk(n_freqs=513, f_min=0.0, f_max=12000.0, n_mels=100, sample_rate=24000):
    """torchaudio.functional.melscale_fbanks(..., norm=None, mel_scale='htk') -> (n_freqs, n_mels)."""
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = np.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return np.maximum(0.0, np.minimum(down, up)).astype(F32)


def mel_basis_slaney(n_fft=1024, n_mels=100, sample_rate=24000, fmin=0.0, fmax=None):
    """librosa.filters.mel (slaney scale, slaney norm) -> (n_mels, n_fft/2+1): the basis of the reference's bigvgan-type mel
    (modeling_modified/F5/modules.py:45; librosa is un-vendored: restated from its published definition)."""
    fmax = sample_rate / 2.0 if fmax is None else float(fmax)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    h2m = lambda f: f / f_sp if f < min_log_hz else min_log_mel + math.log(f / min_log_hz) / logstep
    mels = np.linspace(h2m(fmin), h2m(fmax), n_mels + 2)
    mel_f = np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    freqs = np.linspace(0.0, sample_rate / 2.0, n_fft // 2 + 1)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - freqs[None, :]
    w = np.maximum(0.0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
    return (w * (2.0 / (mel_f[2:] - mel_f[:-2]))[:, None]).astype(F32)


def bigvgan_mel(audio_f32, n_fft=1024, n_mels=100, sample_rate=24000, hop=256):
    """get_bigvgan_mel_spectrogram (modules.py:30-72): reflect pad (n_fft - hop) / 2 each side, torch.stft(center=False) with the
    periodic Hann window, sqrt(re^2 + im^2 + 1e-9), slaney mel basis, log(clamp(., 1e-5)).  audio (L,) float -> (frames, n_mels)."""
    pad = (n_fft - hop) // 2
    xp = np.pad(audio_f32.astype(np.float64), (pad, pad), mode="reflect")
    nfr = (len(xp) - n_fft) // hop + 1
    idx = np.arange(nfr)[:, None] * hop + np.arange(n_fft)[None, :]
    spec = np.fft.rfft(xp[idx] * hann_periodic(n_fft).astype(np.float64)[None, :], axis=1)           # (F, n_fft/2+1)
    mag = np.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9).astype(F32)
    return np.log(np.maximum(mag @ mel_basis_slaney(n_fft, n_mels, sample_rate).T, F32(1e-5))).astype(F32)


# ------------------------------------------------------------------------------------------------
# tables of the exported graphs
# ------------------------------------------------------------------------------------------------
def rope_tables(n, head_dim=64):
    """cos/sin (n, head_dim) with interleaved-pair frequencies, rounded through fp16
    (Export_F5.py:107-112)."""
    # torch: 1.0 / (10000.0 ** (arange(0, D, 2).float() / D)) — a correctly rounded fp32 pow, then an fp32 division.  numpy's own
    # float32 power differs from it in the last bit for 2 of the 32 exponents, which at n ~ 4000 moves the angle by more than an
    # fp16 ulp of the table (found by the N = 4096 limit test, round 4): form the power in float64 and round it to fp32 first.
    expo = (np.arange(0, head_dim, 2, dtype=F32) / F32(head_dim)).astype(np.float64)
    inv_freq = (F32(1.0) / np.power(np.float64(10000.0), expo).astype(F32)).astype(F32)
    freqs = np.outer(np.arange(n, dtype=F32), inv_freq).astype(F32)
    freqs = np.repeat(freqs, 2, axis=-1)
    return np.cos(freqs).astype(np.float16).astype(F32), np.sin(freqs).astype(np.float16).astype(F32)


def text_pos_table(n, dim):
    """precompute_freqs_cis(dim, ...)[:n] (modules.py:196-207): [cos || sin], dim wide."""
    freqs = (1.0 / (10000.0 ** (np.arange(0, dim, 2, dtype=F32)[: dim // 2] / F32(dim)))).astype(F32)
    ang = np.outer(np.arange(n, dtype=F32), freqs).astype(F32)
    return np.concatenate([np.cos(ang), np.sin(ang)], axis=-1).astype(F32)


def time_tables(cfg, st):
    """sway-sampled grid, delta_t and the time-MLP table (Export_F5.py:145-165)."""
    steps = cfg.nfe_step
    t = np.linspace(0, 1, steps, dtype=F32)
    ts = (t + F32(cfg.sway_coef) * (np.cos(F32(math.pi) * F32(0.5) * t) - F32(1.0) + t)).astype(F32)
    delta = np.diff(ts).astype(F32)
    half = cfg.freq_embed_dim // 2
    ef = F32(math.log(10000) / (half - 1))
    ef = (F32(1000.0) * np.exp(np.arange(half, dtype=F32) * -ef)).astype(F32)
    out = np.zeros((steps, cfg.dim), dtype=F32)
    p = "transformer.time_embed.time_mlp."
    for i in range(steps):
        emb = (ts[i] * ef).astype(F32)
        emb = np.concatenate([np.sin(emb), np.cos(emb)]).astype(F32)
        h = silu(linear(emb[None], st[p + "0.weight"], st[p + "0.bias"]))
        out[i] = linear(h, st[p + "2.weight"], st[p + "2.bias"])[0]
    return ts, delta, out


# ------------------------------------------------------------------------------------------------
# text embedding — dit.py:49-73, modules.py:217-261
# ------------------------------------------------------------------------------------------------
def dwconv7(x, w, b):
    """depthwise Conv1d k7 pad 3 over the sequence; x (N, C), w (C,1,7)."""
    N, C = x.shape
    xp = np.zeros((N + 6, C), dtype=F32)
    xp[3:3 + N] = x
    y = np.zeros((N, C), dtype=F32)
    for j in range(7):
        y += xp[j:j + N] * w[:, 0, j][None, :]
    return (y + b[None, :]).astype(F32)


def convnext_v2_block(x, st, p):
    r = x
    y = dwconv7(x, st[p + "dwconv.weight"], st[p + "dwconv.bias"])
    y = layer_norm(y) * st[p + "norm.weight"] + st[p + "norm.bias"]
    y = gelu_erf(linear(y.astype(F32), st[p + "pwconv1.weight"], st[p + "pwconv1.bias"]))
    gx = np.sqrt((y.astype(np.float64) ** 2).sum(axis=0, keepdims=True)).astype(F32)      # L2 over the sequence
    nx = gx / (gx.mean(axis=-1, keepdims=True) + F32(1e-6))
    y = (st[p + "grn.gamma"].reshape(1, -1) * (y * nx) + st[p + "grn.beta"].reshape(1, -1) + y).astype(F32)
    y = linear(y, st[p + "pwconv2.weight"], st[p + "pwconv2.bias"])
    return (r + y).astype(F32)


def text_embed(cfg, st, ids, n):
    """ids (n,) int (already +1 and zero padded) -> text, text_drop (n, text_dim)."""
    emb = st["transformer.text_embed.text_embed.weight"]
    mask = (ids == 0)[:, None]
    pos = text_pos_table(n, cfg.text_dim)
    outs = []
    for src in (emb[ids], np.repeat(emb[0:1], n, axis=0)):
        x = (src + pos).astype(F32)
        x = np.where(mask, F32(0), x)
        for l in range(cfg.conv_layers):
            x = convnext_v2_block(x, st, f"transformer.text_embed.text_blocks.{l}.")
            x = np.where(mask, F32(0), x)
        outs.append(x.astype(F32))
    return outs[0], outs[1]


# ------------------------------------------------------------------------------------------------
# F5Preprocess.forward — Export_F5.py:117-141
# ------------------------------------------------------------------------------------------------
def preprocess(cfg, st, audio_i16, text_ids, max_duration, noise):
    """audio (L,) int16, text_ids (T,) int32, noise (N,100) injected (the reference draws it inside
    ORT).  Returns dict with the graph's eight outputs (batch axes dropped)."""
    N = int(max_duration)
    a = audio_i16.astype(F32) * F32(1.0 / 32768.0)
    if getattr(cfg, "mel_spec_type", "vocos") == "bigvgan":       # the F5 *_bigvgan checkpoints' prompt features (modules.py:30-72)
        mel = bigvgan_mel(a, cfg.n_fft, cfg.mel_dim, cfg.sample_rate, cfg.hop_length)
    else:
        re, im = stft_b(a, cfg.n_fft, cfg.hop_length)
        fb = melscale_fbanks_htk(cfg.n_freq, 0.0, cfg.sample_rate // 2, cfg.mel_dim, cfg.sample_rate).T   # (100, 513)
        mel = np.log(np.maximum((fb @ np.sqrt(re * re + im * im)).T, F32(1e-5))).astype(F32)             # (R, 100)
    R = mel.shape[0]
    mel_pad = np.zeros((N, cfg.mel_dim), dtype=F32)
    mel_pad[:R] = mel
    ids = np.zeros(N, dtype=np.int64)
    ids[:len(text_ids)] = text_ids.astype(np.int64) + 1
    text, drop = text_embed(cfg, st, ids, N)
    cos, sin = rope_tables(N, cfg.dim_head)
    return {
        "noise": noise.astype(F32), "rope_cos": cos, "rope_sin": sin,
        "cat_mel_text": np.concatenate([mel_pad, text], axis=-1).astype(F32),
        "cat_mel_text_drop": np.concatenate([np.zeros_like(mel_pad), drop], axis=-1).astype(F32),
        "ref_signal_len": R, "mel": mel,
    }


# ------------------------------------------------------------------------------------------------
# DiT — dit.py:205-220, modules.py
# ------------------------------------------------------------------------------------------------
def grouped_conv31(x, w, b, groups):
    """x (N, C) channels-last; w (C, C/g, k) ; zero pad k//2.  Per group: im2col (a strided window view of the padded,
    contiguous group slab) and one sgemm against the (k*cg, cg) matrix of that group's taps."""
    N, C = x.shape
    k = w.shape[2]
    cg = C // groups
    y = np.empty((N, C), dtype=F32)
    for g in range(groups):
        xg = np.zeros((N + k - 1, cg), dtype=F32)
        xg[k // 2:k // 2 + N] = x[:, g * cg:(g + 1) * cg]
        cols = np.lib.stride_tricks.as_strided(xg, shape=(N, k * cg), strides=(xg.strides[0], xg.strides[1]), writeable=False)
        wg = np.ascontiguousarray(w[g * cg:(g + 1) * cg].transpose(2, 1, 0).reshape(k * cg, cg))   # [(tap, ci)][co]
        y[:, g * cg:(g + 1) * cg] = np.ascontiguousarray(cols) @ wg      # materialise: overlapping strides are not BLAS-able
    return (y + b[None, :]).astype(F32)


def input_embed(cfg, st, x, cond):
    p = "transformer.input_embed."
    h = linear(np.concatenate([x, cond], axis=-1), st[p + "proj.weight"], st[p + "proj.bias"])
    c = mish(grouped_conv31(h, st[p + "conv_pos_embed.conv1d.0.weight"], st[p + "conv_pos_embed.conv1d.0.bias"],
                            cfg.pos_conv_groups))
    c = mish(grouped_conv31(c, st[p + "conv_pos_embed.conv1d.2.weight"], st[p + "conv_pos_embed.conv1d.2.bias"],
                            cfg.pos_conv_groups))
    return (c + h).astype(F32)


def rope_apply(z, cos, sin):
    """z (..., N, D): z*cos + rot(z)*sin with rot(z)[2j] = -z[2j+1], rot(z)[2j+1] = z[2j]."""
    rot = np.empty_like(z)
    rot[..., 0::2] = -z[..., 1::2]
    rot[..., 1::2] = z[..., 0::2]
    return (z * cos + rot * sin).astype(F32)


def attention(cfg, st, p, u, cos, sin):
    """AttnProcessor.__call__ (modules.py:449-468); u (2, N, d).  q/k weights are pre-scaled.  softmax(q k^T) v is formed one
    (branch, head) at a time so that the N x N logits stay cache resident (same arithmetic as the batched form)."""
    B, N, d = u.shape
    H, D = cfg.heads, cfg.dim_head
    q = linear(u, st[p + "to_q.weight"], st[p + "to_q.bias"]).reshape(B, N, H, D).transpose(0, 2, 1, 3)
    k = linear(u, st[p + "to_k.weight"], st[p + "to_k.bias"]).reshape(B, N, H, D).transpose(0, 2, 1, 3)
    v = linear(u, st[p + "to_v.weight"], st[p + "to_v.bias"]).reshape(B, N, H, D).transpose(0, 2, 1, 3)
    o = np.empty((B, N, H, D), dtype=F32)

    def head(j):
        bi, h = divmod(j, H)
        qh, kh = rope_apply(q[bi, h], cos, sin), rope_apply(k[bi, h], cos, sin)
        s = qh @ kh.T                                                   # (N, N) fp32
        if getattr(cfg, "ref_fp16_attn", False):                        # fp16/modules.py:467: fp16 scores, .float() * 100.0
            s = s.astype(np.float16).astype(F32) * F32(100.0)
        s -= s.max(axis=-1, keepdims=True)
        np.exp(s, out=s)
        s /= s.sum(axis=-1, keepdims=True)
        o[bi, :, h, :] = s @ np.ascontiguousarray(v[bi, h])

    if _POOL is None:
        for j in range(B * H):
            head(j)
    else:
        with _blas_limit(1):
            list(_POOL.map(head, range(B * H)))
    return linear(o.reshape(B, N, H * D), st[p + "to_out.0.weight"], st[p + "to_out.0.bias"])


def dit_block(cfg, st, i, x, t_emb, cos, sin):
    p = f"transformer.transformer_blocks.{i}."
    emb = linear(silu(t_emb)[None], st[p + "attn_norm.linear.weight"], st[p + "attn_norm.linear.bias"])[0]
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = np.split(emb, 6)
    u = layer_norm(x) * (F32(1) + sc_a) + sh_a
    x = (x + g_a * attention(cfg, st, p + "attn.", u.astype(F32), cos, sin)).astype(F32)
    u = (layer_norm(x) * (F32(1) + sc_m) + sh_m).astype(F32)
    ff = linear(gelu_tanh(linear(u, st[p + "ff.ff.0.0.weight"], st[p + "ff.ff.0.0.bias"])),
                st[p + "ff.ff.2.weight"], st[p + "ff.ff.2.bias"])
    return (x + g_m * ff).astype(F32)


def dit_forward(cfg, st, x, cond, cond_drop, t_emb, cos, sin, taps=None):
    """x (N,100), cond/cond_drop (N,612), t_emb (dim,) -> (2, N, 100)."""
    h = np.stack([input_embed(cfg, st, x, cond), input_embed(cfg, st, x, cond_drop)], axis=0)
    if taps is not None:
        taps["input_embed"] = h
    for i in range(cfg.depth):
        h = dit_block(cfg, st, i, h, t_emb, cos, sin)
        if taps is not None:
            taps[f"block.{i}"] = h
    emb = linear(silu(t_emb)[None], st["transformer.norm_out.linear.weight"], st["transformer.norm_out.linear.bias"])[0]
    sc, sh = np.split(emb, 2)
    h = (layer_norm(h) * (F32(1) + sc) + sh).astype(F32)
    return linear(h, st["transformer.proj_out.weight"], st["transformer.proj_out.bias"])


def transformer_step(cfg, st, tables, noise, pre, k):
    """One F5Transformer.forward call at grid index k (Export_F5.py:167-182).  Returns new noise."""
    _, delta, texp = tables
    pred = dit_forward(cfg, st, noise, pre["cat_mel_text"], pre["cat_mel_text_drop"], texp[k], pre["rope_cos"],
                       pre["rope_sin"])
    return (noise + (pred[0] + (pred[0] - pred[1]) * F32(cfg.cfg_strength)) * delta[k]).astype(F32)


def sample(cfg, st, pre, tables=None):
    tables = tables or time_tables(cfg, st)
    x = pre["noise"].copy()
    for k in range(cfg.nfe_step - 1):
        x = transformer_step(cfg, st, tables, x, pre, k)
    return x


# ------------------------------------------------------------------------------------------------
# F5Decode.forward — Export_F5.py:193-203 ; vocos/*
# ------------------------------------------------------------------------------------------------
def l2norm_affine(x, w, b):
    """w' * x / ||x||_2 (over channels, no eps) + b ; x (F, C) channels-last."""
    nrm = np.sqrt((x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True))
    return (w[None, :] * (x / nrm).astype(F32) + b[None, :]).astype(F32)


def conv1d_cl(x, w, b, pad):
    """dense Conv1d on channels-last x (T, Ci); w (Co, Ci, k)."""
    T, Ci = x.shape
    k = w.shape[2]
    xp = np.zeros((T + 2 * pad, Ci), dtype=F32)
    xp[pad:pad + T] = x
    y = np.zeros((T, w.shape[0]), dtype=F32)
    for j in range(k):
        y += xp[j:j + T] @ w[:, :, j].T
    return (y + b[None, :]).astype(F32)


def vocos_decode(cfg, st, mel):
    """mel (F, 100) channels-last -> mag, phase each (513, F)."""
    p = "vocos.backbone."
    h = conv1d_cl(mel, st[p + "embed.weight"], st[p + "embed.bias"], 3)
    h = l2norm_affine(h, st[p + "norm.weight"], st[p + "norm.bias"])
    for l in range(cfg.vocos_layers):
        q = p + f"convnext.{l}."
        z = dwconv7(h, st[q + "dwconv.weight"], st[q + "dwconv.bias"])
        z = l2norm_affine(z, st[q + "norm.weight"], st[q + "norm.bias"])
        z = gelu_erf(linear(z, st[q + "pwconv1.weight"], st[q + "pwconv1.bias"]))
        z = linear(z, st[q + "pwconv2.weight"], st[q + "pwconv2.bias"])
        h = (h + z).astype(F32)
    h = l2norm_affine(h, st[p + "final_layer_norm.weight"], st[p + "final_layer_norm.bias"])
    s = linear(h, st["vocos.head.out.weight"], st["vocos.head.out.bias"])          # (F, n_fft+2)
    nb = cfg.n_freq
    mag = np.minimum(np.exp(s[:, :nb]), F32(100.0)).T.astype(F32)
    return mag, s[:, nb:].T.astype(F32)


def decode(cfg, st, denoised, ref_signal_len, return_float=False):
    """denoised (N,100) -> int16 ((N-R-1)*hop,)  [clamp, *32767, truncate]."""
    mel = denoised[int(ref_signal_len):]
    mag, ph = vocos_decode(cfg, st, mel)
    sig = istft_a(mag, ph, cfg.n_fft, cfg.hop_length, cfg.max_signal_length)
    if return_float:
        return sig
    return (np.clip(sig, -1.0, 1.0) * F32(32767.0)).astype(np.int16)


# ------------------------------------------------------------------------------------------------
# host-side text front end (F5-TTS-ONNX-Inference.py:140-148, 227-231) restated for ASCII prompts
# ------------------------------------------------------------------------------------------------
def list_str_to_idx(chars, vocab):
    return np.asarray([vocab.get(c, 0) for c in chars], dtype=np.int32)


def max_duration(audio_len, ref_text, gen_text, hop=256, speed=1.0):
    ref_len = len(ref_text.encode("utf-8"))
    gen_len = len(gen_text.encode("utf-8"))
    ref_audio_len = audio_len // hop + 1
    return ref_audio_len + int(ref_audio_len / ref_len * gen_len / speed)
