"""Golden vectors for IndexTTS graph A.  Build-container only.

Runs the reference wrapper class IndexTTS_A (exec'd from /root/reference IndexTTS/Export_IndexTTS.py:60-200 where it lies: the
weight folds of __init__, the mel front end, the Conformer layer loop with the rel-pos attention spelled out, rel_shift, the
Perceiver loop, the ECAPA attentive statistics pooling and the conditioning 1x1 convs) over STAND-IN sub-modules for the
un-vendored ``indextts`` package, loaded with the seeded synthetic weights of ``mi355tts.weights.cond_spec`` by strict
``load_state_dict`` (so the key names / shapes of the spec are what these modules define).  The stand-ins restate the published
module definitions (wenet Conv2dSubsampling2 / ConvolutionModule / PositionwiseFeedForward / RelPositionalEncoding; lucidrains
PerceiverResampler Attention holders / FeedForward(GEGLU) / RMSNorm; speechbrain TDNNBlock / Res2NetBlock / SEBlock /
SERes2NetBlock / BatchNorm1d): "parity unpinned" for those definitions, pinned for every line of the wrapper.

Two textual patches make the wrapper runnable outside ``torch.onnx.export`` tracing, where ``tensor.shape[i]`` is a tensor:
``x.shape[2].unsqueeze(0)`` -> ``x.shape[2]`` and ``mel_signal.shape[-1].unsqueeze(0)`` -> ``mel_signal.shape[-1]``.

    python tests/golden/make_golden_indextts_a.py      # writes tests/golden/indextts_a.npz
"""
from __future__ import annotations

import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as Fn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import _ref_import as R                       # noqa: E402
from mi355tts import weights as W             # noqa: E402
from mi355tts.config import IndexCondConfig   # noqa: E402

SEED = 9527


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


# ---- wenet-style Conformer pieces (published definitions) -------------------------------------------------------------
class RelPositionalEncoding(nn.Module):
    def __init__(self, d, max_len):
        super().__init__()
        self.xscale = math.sqrt(d)
        pe = torch.zeros(max_len, d)
        pos = torch.arange(0, max_len, dtype=torch.float32).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2, dtype=torch.float32) * -(math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.pe = pe.unsqueeze(0)


class Conv2dSubsampling2(nn.Module):
    def __init__(self, idim, odim, max_len):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(1, odim, 3, 2), nn.ReLU())
        self.out = nn.Sequential(nn.Linear(odim * ((idim - 1) // 2), odim))
        self.pos_enc = RelPositionalEncoding(odim, max_len)


class RelAttnHolder(nn.Module):
    def __init__(self, h, d):
        super().__init__()
        self.h, self.d_k = h, d // h
        self.linear_q, self.linear_k, self.linear_v, self.linear_out = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.linear_pos = nn.Linear(d, d, bias=False)
        self.pos_bias_u = nn.Parameter(torch.zeros(h, d // h))
        self.pos_bias_v = nn.Parameter(torch.zeros(h, d // h))


class ConvModule(nn.Module):
    def __init__(self, d, k):
        super().__init__()
        self.pointwise_conv1 = nn.Conv1d(d, 2 * d, 1)
        self.depthwise_conv = nn.Conv1d(d, d, k, padding=(k - 1) // 2, groups=d)
        self.norm = nn.LayerNorm(d, eps=1e-5)
        self.activation = nn.SiLU()
        self.pointwise_conv2 = nn.Conv1d(d, d, 1)


class FeedForward(nn.Module):
    def __init__(self, d, lin):
        super().__init__()
        self.w_1, self.w_2, self.activation = nn.Linear(d, lin), nn.Linear(lin, d), nn.SiLU()

    def forward(self, x):
        return self.w_2(self.activation(self.w_1(x)))


class EncoderLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        d = cfg.enc_dim
        self.self_attn = RelAttnHolder(cfg.enc_heads, d)
        self.conv_module = ConvModule(d, cfg.enc_kernel)
        self.feed_forward = FeedForward(d, cfg.enc_linear)
        self.norm_mha, self.norm_conv, self.norm_ff, self.norm_final = (nn.LayerNorm(d, eps=cfg.ln_eps) for _ in range(4))


class ConditioningEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.embed = Conv2dSubsampling2(cfg.n_mels, cfg.enc_dim, 5000)
        self.encoders = nn.ModuleList([EncoderLayer(cfg) for _ in range(cfg.enc_blocks)])
        self.after_norm = nn.LayerNorm(cfg.enc_dim, eps=cfg.ln_eps)


# ---- lucidrains-style Perceiver pieces --------------------------------------------------------------------------------
class PercAttnHolder(nn.Module):
    def __init__(self, dim, dim_head, heads):
        super().__init__()
        self.heads = heads
        inner = dim_head * heads
        self.to_q, self.to_kv, self.to_out = nn.Linear(dim, inner, bias=False), nn.Linear(dim, 2 * inner, bias=False), nn.Linear(inner, dim, bias=False)


class GEGLU(nn.Module):
    def forward(self, x):
        x, gate = x.chunk(2, dim=-1)
        return Fn.gelu(gate) * x


class RMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return Fn.normalize(x, dim=-1) * self.scale * self.gamma


class Perceiver(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, ffi = cfg.model_dim, cfg.perc_ff
        self.proj_context = nn.Linear(cfg.enc_dim, D)
        self.latents = nn.Parameter(torch.zeros(cfg.latents, D))
        self.layers = nn.ModuleList([nn.ModuleList([PercAttnHolder(D, cfg.perc_dim_head, cfg.perc_heads),
                                                    nn.Sequential(nn.Linear(D, 2 * ffi), GEGLU(), nn.Linear(ffi, D))])
                                     for _ in range(cfg.perc_depth)])
        self.norm = RMSNorm(D)


# ---- speechbrain-style ECAPA-TDNN pieces ------------------------------------------------------------------------------
class SBConv1d(nn.Module):
    """speechbrain Conv1d wrapper: 'same' padding done by hand with reflect mode, inner module named ``conv``."""
    def __init__(self, cin, cout, k, d=1):
        super().__init__()
        self.k, self.d = k, d
        self.conv = nn.Conv1d(cin, cout, k, dilation=d)

    def forward(self, x):
        pad = self.d * (self.k - 1) // 2
        if pad:
            x = Fn.pad(x, (pad, pad), mode="reflect")
        return self.conv(x)


class SBBatchNorm1d(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.norm = nn.BatchNorm1d(c)

    def forward(self, x):
        return self.norm(x)


class TDNNBlock(nn.Module):
    def __init__(self, cin, cout, k, d):
        super().__init__()
        self.conv, self.activation, self.norm = SBConv1d(cin, cout, k, d), nn.ReLU(), SBBatchNorm1d(cout)

    def forward(self, x):
        return self.norm(self.activation(self.conv(x)))


class Res2NetBlock(nn.Module):
    def __init__(self, c, scale, k, d):
        super().__init__()
        self.scale = scale
        self.blocks = nn.ModuleList([TDNNBlock(c // scale, c // scale, k, d) for _ in range(scale - 1)])

    def forward(self, x):
        y = []
        for i, xi in enumerate(torch.chunk(x, self.scale, dim=1)):
            if i == 0:
                yi = xi
            elif i == 1:
                yi = self.blocks[i - 1](xi)
            else:
                yi = self.blocks[i - 1](xi + yi)
            y.append(yi)
        return torch.cat(y, dim=1)


class SEBlock(nn.Module):
    def __init__(self, c, se):
        super().__init__()
        self.conv1, self.conv2 = SBConv1d(c, se, 1), SBConv1d(se, c, 1)

    def forward(self, x):
        s = x.mean(dim=2, keepdim=True)
        return torch.sigmoid(self.conv2(torch.relu(self.conv1(s)))) * x


class SERes2NetBlock(nn.Module):
    def __init__(self, cin, cout, scale, se, k, d):
        super().__init__()
        self.tdnn1, self.res2net_block = TDNNBlock(cin, cout, 1, 1), Res2NetBlock(cout, scale, k, d)
        self.tdnn2, self.se_block = TDNNBlock(cout, cout, 1, 1), SEBlock(cout, se)

    def forward(self, x):
        return self.se_block(self.tdnn2(self.res2net_block(self.tdnn1(x)))) + x


class ASP(nn.Module):
    def __init__(self, c, att):
        super().__init__()
        self.tdnn, self.tanh, self.conv = TDNNBlock(3 * c, att, 1, 1), nn.Tanh(), SBConv1d(att, c, 1)


class ECAPA(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        ch, ks, ds = cfg.spk_channels, cfg.spk_kernels, cfg.spk_dilations
        self.blocks = nn.ModuleList([TDNNBlock(cfg.n_mels, ch[0], ks[0], ds[0])] +
                                    [SERes2NetBlock(ch[i - 1], ch[i], cfg.spk_res2net_scale, cfg.spk_se, ks[i], ds[i]) for i in range(1, len(ch) - 1)])
        self.mfa = TDNNBlock(ch[-1], ch[-1], ks[-1], ds[-1])
        self.asp = ASP(ch[-1], cfg.spk_att)
        self.asp_bn = SBBatchNorm1d(2 * ch[-1])
        self.fc = SBConv1d(2 * ch[-1], cfg.spk_embed, 1)


class VocoderConds(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.speaker_encoder = ECAPA(cfg)
        self.cond_layer = nn.Conv1d(cfg.spk_embed, cfg.voc_initial, 1)
        self.conds = nn.ModuleList([nn.Conv1d(cfg.spk_embed, c, 1) for c in cfg.voc_channels])
        self.num_upsamples = len(cfg.voc_channels)


class GPTConds(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.conditioning_encoder = ConditioningEncoder(cfg)
        self.perceiver_encoder = Perceiver(cfg)


def build(cfg, state):
    gpt, voc = GPTConds(cfg).eval().float(), VocoderConds(cfg).eval().float()
    gs = {k[4:]: t(v) for k, v in state.items() if k.startswith("gpt.")}
    vs = {k[8:]: t(v) for k, v in state.items() if k.startswith("bigvgan.")}
    gpt.load_state_dict(gs, strict=True)
    missing, unexpected = voc.load_state_dict(vs, strict=False)
    assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
    return types.SimpleNamespace(gpt=gpt, bigvgan=voc)


def gen(cfg, audio_len, tag, out):
    state = W.synth_state(W.cond_spec(cfg), SEED)
    ns = {"torch": torch, "torchaudio": types.SimpleNamespace(functional=types.SimpleNamespace(melscale_fbanks=R.melscale_fbanks))}
    patches = (("enc_len = x.shape[2].unsqueeze(0)", "enc_len = x.shape[2]"),
               ("ref_signal_len = mel_signal.shape[-1].unsqueeze(0)", "ref_signal_len = mel_signal.shape[-1]"))
    R.exec_lines(R.REF + "/IndexTTS/Export_IndexTTS.py", 60, 200, ns, replace=patches)
    stft_mod = R._load("indextts_stft_process", R.REF + "/IndexTTS/STFT_Process.py") if "onnxruntime" in sys.modules else None
    if stft_mod is None:
        sys.modules["onnxruntime"] = types.ModuleType("onnxruntime")
        stft_mod = R._load("indextts_stft_process", R.REF + "/IndexTTS/STFT_Process.py")
    with torch.no_grad():
        idx = build(cfg, state)
        custom_stft = stft_mod.STFT_Process(model_type="stft_B", n_fft=cfg.n_fft, hop_len=cfg.hop, win_length=cfg.n_fft, max_frames=0,
                                            window_type="hann").eval()
        part_a = ns["IndexTTS_A"](idx, custom_stft, cfg.n_fft, cfg.n_mels, cfg.sample_rate, cfg.max_signal_len)
        part_a.audio_pad = t(state["audio_pad"]).reshape(1, 1, -1)               # the wrapper draws it with torch.randn (:94): pinned here
        rng = np.random.default_rng(SEED)
        tt = np.arange(audio_len) / cfg.sample_rate
        audio = np.clip(0.2 * 32767 * np.sin(2 * np.pi * 180.0 * tt) * (1 + 0.5 * np.sin(2 * np.pi * 3.0 * tt)) + rng.normal(0, 900, audio_len),
                        -32768, 32767).astype(np.int16)
        res = part_a(t(audio).reshape(1, 1, -1))
        n = len(cfg.voc_channels)
        out[tag + "audio"] = audio
        for i in range(n):
            out[tag + f"cond_{i}"] = res[i].numpy().reshape(-1)
        out[tag + "cond_layer"] = res[n].numpy().reshape(-1)
        out[tag + "conds_latent"] = res[n + 1].numpy()[0]
    print(tag, {k: v.shape for k, v in out.items() if k.startswith(tag)})


if __name__ == "__main__":
    out = {}
    gen(IndexCondConfig.small(), 5000, "s_", out)
    gen(IndexCondConfig.small(), 12345, "r_", out)            # a second, ragged length
    np.savez_compressed(os.path.join(HERE, "indextts_a.npz"), **out)
    print("indextts_a.npz written")
