"""TEST INFRASTRUCTURE — CPU oracle (numpy restatement) of IndexTTS graph A.

Checker only: imported by tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg; never by the product path.

Restates (paths relative to /root/reference):
  IndexTTS_A.forward            IndexTTS/Export_IndexTTS.py:131-200   (the wrapper: mel front end, the Conformer layer loop with the
                                rel-pos attention spelled out, the Perceiver loop, ECAPA attentive statistics pooling, cond layers)
  rel_shift / _compute_statistics   IndexTTS/Export_IndexTTS.py:60-71
  IndexTTS_A.__init__ folds     :88-129  (applied by mi355tts.weights.fold_cond: this file takes the FOLDED state)
  STFT (constant padding)       IndexTTS/STFT_Process.py:86-98, 147-160

Parity pin: the wrapper lines above are exec'd where they lie by tests/golden/make_golden_indextts_a.py over stand-in modules and
this file is checked against that fixture (tests/test_oracle_indextts_a.py).  The SUB-MODULES the wrapper calls as black boxes —
wenet's Conv2dSubsampling2 / PositionwiseFeedForward / ConvolutionModule pieces / RelPositionalEncoding table, lucidrains'
PerceiverResampler FeedForward (GEGLU) and RMSNorm, speechbrain's TDNNBlock / Res2NetBlock / SEBlock / BatchNorm1d — have no source
in the reference tree (un-vendored ``indextts`` package): they are restated from their published definitions, once here and once
(as torch modules) in the generator — "parity unpinned" for those definitions, pinned for everything the wrapper itself computes.
"""
from __future__ import annotations

import math

import numpy as np

from . import f5_np as F

F32 = np.float32


def layer_norm_affine(x, w, b, eps):
    mu = x.mean(axis=-1, keepdims=True, dtype=np.float64)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True, dtype=np.float64)
    return (((x - mu) / np.sqrt(var + eps)).astype(F32) * w + b).astype(F32)


def silu(x):
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def gelu_erf(x):
    return F.gelu_erf(x)


def softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return (e / e.sum(axis=axis, keepdims=True)).astype(F32)


def rel_pos_table(max_len, d):
    """wenet RelPositionalEncoding.pe (published definition): pe[p, 2i] = sin(p / 10000^(2i/d)), pe[p, 2i+1] = cos(...), kept as
    fp16 by the wrapper (:87) and widened again in forward (:139)."""
    pos = np.arange(max_len, dtype=F32)[:, None]
    div = np.exp(np.arange(0, d, 2, dtype=F32) * F32(-(math.log(10000.0) / d))).astype(F32)
    pe = np.zeros((max_len, d), dtype=F32)
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe.astype(np.float16).astype(F32)


def rel_shift(x):
    """rel_shift (:67-71) on (h, T, T): prepend a zero column, reinterpret the (T, T+1) block as (T+1, T), drop its first row."""
    h, T, _ = x.shape
    xp = np.concatenate([np.zeros((h, T, 1), dtype=F32), x], axis=-1).reshape(h, T + 1, T)
    return xp[:, 1:].reshape(h, T, T)


def conv1d_ncl(x, w, b, dilation=1, pad=0, pad_mode="zeros", groups=1):
    """x (C, T) channels first, w (Co, Ci/g, k)."""
    Co, cig, k = w.shape
    C, T = x.shape
    if pad:
        x = np.pad(x, ((0, 0), (pad, pad)), mode="reflect" if pad_mode == "reflect" else "constant")
    To = x.shape[1] - dilation * (k - 1)
    y = np.zeros((Co, To), dtype=F32)
    cog = Co // groups
    for g in range(groups):
        xs = x[g * cig:(g + 1) * cig]
        for j in range(k):
            y[g * cog:(g + 1) * cog] += w[g * cog:(g + 1) * cog, :, j] @ xs[:, j * dilation:j * dilation + To]
    if b is not None:
        y += b[:, None]
    return y.astype(F32)


def batch_norm(x, st, p, eps):
    """BatchNorm1d in eval mode on (C, T)."""
    return ((x - st[p + "running_mean"][:, None]) / np.sqrt(st[p + "running_var"][:, None] + F32(eps)) * st[p + "weight"][:, None]
            + st[p + "bias"][:, None]).astype(F32)


def tdnn(cfg, st, p, x, k, d):
    """speechbrain TDNNBlock: norm(relu(conv(x))), Conv1d with 'same' reflect padding."""
    pad = d * (k - 1) // 2
    y = conv1d_ncl(x, st[p + "conv.conv.weight"], st[p + "conv.conv.bias"], dilation=d, pad=pad, pad_mode="reflect")
    return batch_norm(np.maximum(y, 0), st, p + "norm.norm.", cfg.bn_eps)


def se_res2net(cfg, st, p, x, k, d):
    """speechbrain SERes2NetBlock (in == out channels: no shortcut conv)."""
    y = tdnn(cfg, st, p + "tdnn1.", x, 1, 1)
    sc = cfg.spk_res2net_scale
    parts = np.split(y, sc, axis=0)
    outs = [parts[0]]
    prev = None
    for i in range(1, sc):
        inp = parts[i] if i == 1 else (parts[i] + prev).astype(F32)
        prev = tdnn(cfg, st, p + f"res2net_block.blocks.{i - 1}.", inp, k, d)
        outs.append(prev)
    y = np.concatenate(outs, axis=0)
    y = tdnn(cfg, st, p + "tdnn2.", y, 1, 1)
    s = y.mean(axis=1, keepdims=True)
    s = np.maximum(conv1d_ncl(s, st[p + "se_block.conv1.conv.weight"], st[p + "se_block.conv1.conv.bias"]), 0)
    s = F32(1.0) / (F32(1.0) + np.exp(-conv1d_ncl(s, st[p + "se_block.conv2.conv.weight"], st[p + "se_block.conv2.conv.bias"])))
    return (s * y + x).astype(F32)


def compute_statistics(x, m):
    """_compute_statistics (:60-63): x (C, T), m (T,) or (C, T) weights."""
    mean = (m * x).sum(axis=1, keepdims=True)
    std = np.sqrt(np.maximum((m * (x - mean) ** 2).sum(axis=1, keepdims=True), F32(1e-6)))
    return mean.astype(F32), std.astype(F32)


def mel_front_end(cfg, st, audio_i16):
    """:132-135.  Returns mel (n_mels, T)."""
    a = audio_i16.astype(F32) * F32(1.0 / 32768.0)
    a = np.concatenate([st["audio_pad"].astype(F32), a])
    half = cfg.n_fft // 2
    xp = np.concatenate([np.zeros(half, F32), a, np.zeros(half, F32)])           # 'constant' padding (STFT_Process.py:147-150)
    nfr = (len(xp) - cfg.n_fft) // cfg.hop + 1
    idx = np.arange(nfr)[:, None] * cfg.hop + np.arange(cfg.n_fft)[None, :]
    frames = xp[idx]
    ck, sk = F.stft_kernels(cfg.n_fft)
    re, im = (ck @ frames.T).astype(F32), (sk @ frames.T).astype(F32)
    fb = F.melscale_fbanks_htk(cfg.n_fft // 2 + 1, 0.0, cfg.sample_rate // 2, cfg.n_mels, cfg.sample_rate).T
    return np.log(np.maximum(fb @ np.sqrt(re * re + im * im), F32(1e-5))).astype(F32)


def conformer(cfg, st, mel):
    """:136-165.  mel (n_mels, T) -> (T2, enc_dim) after after_norm."""
    p = "gpt.conditioning_encoder."
    d, h, dk = cfg.enc_dim, cfg.enc_heads, cfg.enc_dk
    # embed.conv = Conv2d(1, d, 3, 2) + ReLU on (1, 1, T, n_mels); then (T2, d * F2) -> Linear (wenet Conv2dSubsampling2)
    x = mel.T                                                                      # (T, n_mels)
    T2, F2 = (x.shape[0] - 3) // 2 + 1, cfg.sub_freq
    w = st[p + "embed.conv.0.weight"][:, 0]                                        # (d, 3, 3)
    y = np.zeros((d, T2, F2), dtype=F32)
    for i in range(3):
        for j in range(3):
            y += w[:, i, j][:, None, None] * x[i:i + 2 * T2 - 1:2, j:j + 2 * F2 - 1:2][None]
    y = np.maximum(y + st[p + "embed.conv.0.bias"][:, None, None], 0)
    x = (y.transpose(1, 0, 2).reshape(T2, d * F2) @ st[p + "embed.out.0.weight"].T + st[p + "embed.out.0.bias"]).astype(F32)
    pos = rel_pos_table(cfg.max_signal_len, d)[:T2]                                 # :139
    for i in range(cfg.enc_blocks):
        q_ = p + f"encoders.{i}."
        a_ = q_ + "self_attn."
        x1 = layer_norm_affine(x, st[q_ + "norm_mha.weight"], st[q_ + "norm_mha.bias"], cfg.ln_eps)
        q = (x1 @ st[a_ + "linear_q.weight"].T + st[a_ + "linear_q.bias"]).reshape(T2, h, dk).transpose(1, 0, 2)       # (h, T, dk)
        k = (x1 @ st[a_ + "linear_k.weight"].T + st[a_ + "linear_k.bias"]).reshape(T2, h, dk).transpose(1, 0, 2)
        v = (x1 @ st[a_ + "linear_v.weight"].T + st[a_ + "linear_v.bias"]).reshape(T2, h, dk).transpose(1, 0, 2)
        pp = (pos @ st[a_ + "linear_pos.weight"].T).reshape(T2, h, dk).transpose(1, 0, 2)
        ac = (q + st[a_ + "pos_bias_u"][:, None, :]) @ k.transpose(0, 2, 1)
        bd = rel_shift(((q + st[a_ + "pos_bias_v"][:, None, :]) @ pp.transpose(0, 2, 1)).astype(F32))
        o = softmax((ac + bd).astype(F32)) @ v                                      # (h, T, dk)
        o = o.transpose(1, 0, 2).reshape(T2, d) @ st[a_ + "linear_out.weight"].T + st[a_ + "linear_out.bias"]
        x = (x + o).astype(F32)
        res = x
        c_ = q_ + "conv_module."
        y = layer_norm_affine(x, st[q_ + "norm_conv.weight"], st[q_ + "norm_conv.bias"], cfg.ln_eps).T                # (d, T)
        y = conv1d_ncl(y, st[c_ + "pointwise_conv1.weight"], st[c_ + "pointwise_conv1.bias"])
        y = (y[:d] * (F32(1.0) / (F32(1.0) + np.exp(-y[d:])))).astype(F32)          # GLU over channels
        y = conv1d_ncl(y, st[c_ + "depthwise_conv.weight"], st[c_ + "depthwise_conv.bias"], pad=(cfg.enc_kernel - 1) // 2, groups=d)
        y = silu(layer_norm_affine(y.T, st[c_ + "norm.weight"], st[c_ + "norm.bias"], cfg.ln_eps)).T
        y = conv1d_ncl(y, st[c_ + "pointwise_conv2.weight"], st[c_ + "pointwise_conv2.bias"]).T
        x = (y + res).astype(F32)
        f = layer_norm_affine(x, st[q_ + "norm_ff.weight"], st[q_ + "norm_ff.bias"], cfg.ln_eps)
        f = silu(f @ st[q_ + "feed_forward.w_1.weight"].T + st[q_ + "feed_forward.w_1.bias"])
        x = (x + (f @ st[q_ + "feed_forward.w_2.weight"].T + st[q_ + "feed_forward.w_2.bias"])).astype(F32)
        x = layer_norm_affine(x, st[q_ + "norm_final.weight"], st[q_ + "norm_final.bias"], cfg.ln_eps)
    return layer_norm_affine(x, st[p + "after_norm.weight"], st[p + "after_norm.bias"], cfg.ln_eps)


def perceiver(cfg, st, x):
    """:166-176.  x (T2, enc_dim) -> conds_latent (latents, model_dim)."""
    p = "gpt.perceiver_encoder."
    H, dh, inner, ffi = cfg.perc_heads, cfg.perc_dim_head, cfg.perc_inner, cfg.perc_ff
    x = (x @ st[p + "proj_context.weight"].T + st[p + "proj_context.bias"]).astype(F32)
    lat = st[p + "latents"].astype(F32)
    for j in range(cfg.perc_depth):
        a_ = p + f"layers.{j}.0."
        L = lat.shape[0]
        q = (lat @ st[a_ + "to_q.weight"].T).reshape(L, H, dh).transpose(1, 0, 2)
        cx = np.concatenate([lat, x], axis=0)
        kv = cx @ st[a_ + "to_kv.weight"].T
        k = kv[:, :inner].reshape(-1, H, dh).transpose(1, 0, 2)
        v = kv[:, inner:].reshape(-1, H, dh).transpose(1, 0, 2)
        o = softmax((q @ k.transpose(0, 2, 1)).astype(F32)) @ v
        lat = (o.transpose(1, 0, 2).reshape(L, inner) @ st[a_ + "to_out.weight"].T + lat).astype(F32)
        f_ = p + f"layers.{j}.1."
        hgl = lat @ st[f_ + "0.weight"].T + st[f_ + "0.bias"]                        # Linear -> GEGLU: x, gate = chunk(2); gelu(gate) * x
        hgl = (gelu_erf(hgl[:, ffi:].astype(F32)) * hgl[:, :ffi]).astype(F32)
        lat = (hgl @ st[f_ + "2.weight"].T + st[f_ + "2.bias"] + lat).astype(F32)
    nrm = np.maximum(np.sqrt((lat.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)), 1e-12)      # RMSNorm: F.normalize * sqrt(dim) * gamma
    return ((lat / nrm).astype(F32) * F32(math.sqrt(cfg.model_dim)) * st[p + "norm.gamma"]).astype(F32)


def speaker(cfg, st, mel):
    """:178-199.  mel (n_mels, T) -> (list of cond vectors (C_i,), cond_layer vector (voc_initial,))."""
    e = "bigvgan.speaker_encoder."
    T = mel.shape[1]
    x = mel
    feats = []
    nb = len(cfg.spk_channels) - 1
    for i in range(nb):
        if i == 0:
            x = tdnn(cfg, st, e + "blocks.0.", x, cfg.spk_kernels[0], cfg.spk_dilations[0])
        else:
            x = se_res2net(cfg, st, e + f"blocks.{i}.", x, cfg.spk_kernels[i], cfg.spk_dilations[i])
            feats.append(x)
    x = tdnn(cfg, st, e + "mfa.", np.concatenate(feats, axis=0), cfg.spk_kernels[-1], cfg.spk_dilations[-1])
    mean, std = compute_statistics(x, F32(1.0 / T))
    att = np.concatenate([x, np.repeat(mean, T, axis=1), np.repeat(std, T, axis=1)], axis=0)
    att = np.tanh(tdnn(cfg, st, e + "asp.tdnn.", att, 1, 1))
    att = softmax(conv1d_ncl(att, st[e + "asp.conv.conv.weight"], st[e + "asp.conv.conv.bias"]), axis=1)
    mean, std = compute_statistics(x, att)
    emb = np.concatenate([mean, std], axis=0)
    emb = batch_norm(emb, st, e + "asp_bn.norm.", cfg.bn_eps)
    emb = conv1d_ncl(emb, st[e + "fc.conv.weight"], st[e + "fc.conv.bias"])          # (spk_embed, 1)
    cond0 = conv1d_ncl(emb, st["bigvgan.cond_layer.weight"], st["bigvgan.cond_layer.bias"])[:, 0]
    conds = [conv1d_ncl(emb, st[f"bigvgan.conds.{i}.weight"], st[f"bigvgan.conds.{i}.bias"])[:, 0] for i in range(len(cfg.voc_channels))]
    return conds, cond0


def graph_a(cfg, st, audio_i16):
    """IndexTTS_A.forward: audio (L,) int16 -> (save_bigvgan_conds_0..n-1, bigvgan_cond_layer_speaker_embedding, conds_latent)."""
    mel = mel_front_end(cfg, st, np.asarray(audio_i16).reshape(-1))
    conds_latent = perceiver(cfg, st, conformer(cfg, st, mel))
    conds, cond0 = speaker(cfg, st, mel)
    return conds, cond0, conds_latent, mel
