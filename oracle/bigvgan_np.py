"""TEST INFRASTRUCTURE — CPU oracle (numpy restatement) of the reference BigVGAN-v2 generator.

This file is a checker only.  It may be imported by tests/, __graft_entry__.smoke() and
bench.py's ``cpu_baseline`` leg; the product path (text-to-speech-tts-onnx_amd/) never
imports it.

Parity pin: validated against outputs of the reference's own module code
(/root/reference/BigVGAN/modeling_modified/*.py imported in the build container by
tests/golden/make_golden.py) on seeded inputs; fixtures in tests/golden/bigvgan_*.npz,
checked by tests/test_oracle_bigvgan.py.  The un-vendored pieces (NVIDIA BigVGAN
``activations.SnakeBeta``, ``utils.get_padding``) have no in-tree reference source:
for those two formulas parity is *unpinned*  — the golden generator
restates them from the published upstream definition.

Layout here is the reference's: channels-first (B, C, T), float32.
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


# ---------------------------------------------------------------------------------------
# filter taps — BigVGAN/modeling_modified/filter.py:30-62 (kaiser_sinc_filter1d)
# ---------------------------------------------------------------------------------------
def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> np.ndarray:
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    # torch.kaiser_window(L, beta, periodic=False) == np.kaiser(L, beta)
    window = np.kaiser(kernel_size, beta)
    if even:
        time = np.arange(-half_size, half_size) + 0.5
    else:
        time = np.arange(kernel_size) - half_size
    if cutoff == 0:
        return np.zeros(kernel_size, dtype=F32)
    filt = 2 * cutoff * window * np.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.astype(F32)


def aa_filter() -> np.ndarray:
    """The one 12-tap filter every Activation1d uses: ratio 2 => cutoff 0.25, half-width 0.3
    (resample.py:12-27 UpSample1d, :37-48 DownSample1d)."""
    return kaiser_sinc_filter1d(0.5 / 2, 0.6 / 2, 12)


# ---------------------------------------------------------------------------------------
# convolutions (torch.nn.functional.conv1d / conv_transpose1d semantics)
# ---------------------------------------------------------------------------------------
def conv1d(x: np.ndarray, w: np.ndarray, b=None, dilation: int = 1, padding: int = 0) -> np.ndarray:
    """x (B,Ci,T), w (Co,Ci,k) -> (B,Co,T + 2p - d(k-1)); zero padding."""
    B, Ci, T = x.shape
    Co, _, k = w.shape
    To = T + 2 * padding - dilation * (k - 1)
    xp = np.zeros((B, Ci, T + 2 * padding), dtype=F32)
    xp[:, :, padding:padding + T] = x
    out = np.zeros((B, Co, To), dtype=F32)
    for j in range(k):
        out += np.einsum("oc,bct->bot", w[:, :, j], xp[:, :, j * dilation:j * dilation + To], optimize=True)
    if b is not None:
        out += b[None, :, None]
    return out


def conv_transpose1d(x: np.ndarray, w: np.ndarray, b, stride: int, padding: int) -> np.ndarray:
    """x (B,Ci,T), w (Ci,Co,k) -> (B,Co,(T-1)s - 2p + k)."""
    B, Ci, T = x.shape
    _, Co, k = w.shape
    full = np.zeros((B, Co, (T - 1) * stride + k), dtype=F32)
    for j in range(k):
        full[:, :, j:j + (T - 1) * stride + 1:stride] += np.einsum("co,bct->bot", w[:, :, j], x, optimize=True)
    out = full[:, :, padding:full.shape[2] - padding]
    if b is not None:
        out = out + b[None, :, None]
    return np.ascontiguousarray(out, dtype=F32)


# ---------------------------------------------------------------------------------------
# anti-aliased SnakeBeta — act.py:25-29, resample.py:30-34, filter.py:94-98
# ---------------------------------------------------------------------------------------
def snake_beta(x: np.ndarray, alpha_log: np.ndarray, beta_log: np.ndarray, logscale: bool = True) -> np.ndarray:
    """NVIDIA BigVGAN activations.SnakeBeta (un-vendored; published formula):
    x + 1/(beta + 1e-9) * sin^2(alpha x); alpha,beta = exp(param) when logscale."""
    a = np.exp(alpha_log) if logscale else alpha_log
    b = np.exp(beta_log) if logscale else beta_log
    a = a.astype(F32)[None, :, None]
    b = b.astype(F32)[None, :, None]
    s = np.sin(x * a)
    return (x + (F32(1.0) / (b + F32(1e-9))) * (s * s)).astype(F32)


def aa_upsample(x: np.ndarray, h: np.ndarray, pad: int) -> np.ndarray:
    """zero-pad ``pad`` both sides -> depthwise conv_transpose(stride 2) * 2 -> crop [15:-15]
    (resample.py:30-34; pad=5 in blocks, 15 for activation_post: bigvgan.py:370)."""
    B, C, T = x.shape
    P = np.zeros((B, C, T + 2 * pad), dtype=F32)
    P[:, :, pad:pad + T] = x
    L = P.shape[2]
    full = np.zeros((B, C, (L - 1) * 2 + 12), dtype=F32)
    for t in range(12):
        full[:, :, t:t + (L - 1) * 2 + 1:2] += h[t] * P
    full *= F32(2.0)
    return full[:, :, 15:full.shape[2] - 15]


def aa_downsample(x: np.ndarray, h: np.ndarray, pad_l: int, pad_r: int) -> np.ndarray:
    """zero-pad (pad_l, pad_r) -> depthwise conv stride 2 (filter.py:94-98)."""
    B, C, T = x.shape
    Q = np.zeros((B, C, T + pad_l + pad_r), dtype=F32)
    Q[:, :, pad_l:pad_l + T] = x
    To = (Q.shape[2] - 12) // 2 + 1
    out = np.zeros((B, C, To), dtype=F32)
    for t in range(12):
        out += h[t] * Q[:, :, t:t + 2 * (To - 1) + 1:2]
    return out


def activation1d(x, alpha_log, beta_log, h, post: bool = False, logscale: bool = True):
    """Activation1d.forward (act.py:25-29).  post=True uses the pad-15 tables of
    bigvgan.py:370,381-382 and returns T+30 samples."""
    if post:
        u = aa_upsample(x, h, 15)
        s = snake_beta(u, alpha_log, beta_log, logscale)
        return aa_downsample(s, h, 15, 15)
    u = aa_upsample(x, h, 5)
    s = snake_beta(u, alpha_log, beta_log, logscale)
    return aa_downsample(s, h, 5, 6)


# ---------------------------------------------------------------------------------------
# generator — bigvgan.py:132-140 (AMPBlock1.forward), :384-410 (BigVGAN.forward)
# ---------------------------------------------------------------------------------------
def get_padding(kernel_size: int, dilation: int = 1) -> int:
    return int((kernel_size * dilation - dilation) / 2)


def amp_block1(x, st, n: int, k: int, dils, h, logscale=True):
    p = f"resblocks.{n}."
    for l, d in enumerate(dils):
        xt = activation1d(x, st[p + f"activations.{2 * l}.act.alpha"], st[p + f"activations.{2 * l}.act.beta"], h,
                          logscale=logscale)
        xt = conv1d(xt, st[p + f"convs1.{l}.weight"], st[p + f"convs1.{l}.bias"], dilation=d,
                    padding=get_padding(k, d))
        xt = activation1d(xt, st[p + f"activations.{2 * l + 1}.act.alpha"],
                          st[p + f"activations.{2 * l + 1}.act.beta"], h, logscale=logscale)
        xt = conv1d(xt, st[p + f"convs2.{l}.weight"], st[p + f"convs2.{l}.bias"], dilation=1,
                    padding=get_padding(k, 1))
        x = xt + x
    return x


def generator(cfg, st, mel: np.ndarray, taps=None, conds=None) -> np.ndarray:
    """BigVGAN.forward: mel (B, num_mels, F) float32 -> (B, 1, F*hop + 30) float32 in [-1, 1].
    conds (IndexTTS graph F, Export_IndexTTS.py:300-314): list [cond_0 .. cond_{n-1}, cond_pre], each (C,) — added after
    every upsampler / after conv_pre."""
    h = aa_filter()
    x = conv1d(mel.astype(F32), st["conv_pre.weight"], st["conv_pre.bias"], padding=3)
    if conds is not None:
        x = x + conds[-1].astype(F32)[None, :, None]
    if taps is not None:
        taps["conv_pre"] = x
    nk = cfg.num_kernels
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = conv_transpose1d(x, st[f"ups.{i}.0.weight"], st[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        if conds is not None:
            x = (x + conds[i].astype(F32)[None, :, None]).astype(F32)
        if taps is not None:
            taps[f"ups.{i}"] = x
        xs = None
        for j in range(nk):
            y = amp_block1(x, st, i * nk + j, cfg.resblock_kernel_sizes[j], cfg.resblock_dilation_sizes[j], h,
                           cfg.snake_logscale)
            xs = y if xs is None else xs + y
        x = xs * F32(1.0 / nk)
        if taps is not None:
            taps[f"stage.{i}"] = x
    x = activation1d(x, st["activation_post.act.alpha"], st["activation_post.act.beta"], h, post=True,
                     logscale=cfg.snake_logscale)
    x = conv1d(x, st["conv_post.weight"], st.get("conv_post.bias") if cfg.use_bias_at_final else None, padding=3)
    if cfg.use_tanh_at_final:
        x = np.tanh(x)
    else:
        x = np.clip(x, -1.0, 1.0)
    return x.astype(F32)


def bigvgan_int16(cfg, st, mel: np.ndarray) -> np.ndarray:
    """BIGVGAN.forward (BigVGAN/Export_BigVGAN.py:44-49): x32767 -> clamp -> int16 (truncation)."""
    w = generator(cfg, st, mel) * F32(32767.0)
    if cfg.use_tanh_at_final:
        w = np.clip(w, -32768.0, 32767.0)
    return w.astype(np.int16)      # numpy float->int16 cast truncates toward zero, like torch .to(int16)


def indextts_f_int16(cfg, st, latent: np.ndarray, conds) -> np.ndarray:
    """IndexTTS_F.forward (IndexTTS/Export_IndexTTS.py:300-314): latent (T_codes, gpt_dim) channels-last ->
    LayerNorm(latent[:-2]) -> generator with speaker-conditioning biases -> tanh -> clamp(-1,1)*32767 -> int16."""
    x = latent[:-2].astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    x = ((x - mu) / np.sqrt(var + cfg.ln_eps)).astype(F32) * st["final_norm.weight"] + st["final_norm.bias"]
    y = generator(cfg, st, x.T[None].astype(F32), conds=conds)
    return (np.clip(y, -1.0, 1.0) * F32(32767.0)).astype(np.int16)


def indextts_f_float(cfg, st, latent: np.ndarray, conds) -> np.ndarray:
    x = latent[:-2].astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    x = ((x - mu) / np.sqrt(var + cfg.ln_eps)).astype(F32) * st["final_norm.weight"] + st["final_norm.bias"]
    return generator(cfg, st, x.T[None].astype(F32), conds=conds)
