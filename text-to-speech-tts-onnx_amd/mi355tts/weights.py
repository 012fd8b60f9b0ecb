"""Weight specs, deterministic synthetic weights and the blob packer.

There are no checkpoints in the reference tree and no network, so
parity and benchmarks run on *synthetic seeded weights*.  Tensor names and shapes follow the
upstream ``state_dict`` names after ``remove_weight_norm`` (BigVGAN/Export_BigVGAN.py:54) so
the same dict loads into the reference modules (tests/golden/make_golden.py) and into this
engine (``pack_bigvgan`` / ``pack_f5`` -> flat fp32 blob in the canonical order documented in
include/mi355tts.h).

The packer also applies the reference's export-time folds:
  * q/k weight+bias * head_dim**-0.25          (F5_TTS/Export_F5.py:321-333)
  * Vocos LayerNorm weight * sqrt(C)            (F5_TTS/Export_F5.py:390-398)
  * Vocos layer-scale gamma folded into pwconv2 (F5_TTS/Export_F5.py:401-402)
"""
from __future__ import annotations

import hashlib
import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

from .config import IndexCondConfig, BigVGANConfig, F5Config, IndexGPTConfig

Spec = List[Tuple[str, Tuple[int, ...], str]]   # (name, shape, kind)


# --------------------------------------------------------------------------------------
# deterministic, platform-independent normal generator (counter-based: splitmix64 + Box-Muller)
# --------------------------------------------------------------------------------------
def _splitmix64(x: np.ndarray) -> np.ndarray:
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & np.uint64(0xFFFFFFFFFFFFFFFF)
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """N(mean, std^2) float32 tensor that depends only on (seed, name, shape)."""
    n = int(np.prod(shape)) if len(shape) else 1
    h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) * np.uint64(2) + np.uint64(h)
        a = _splitmix64(ctr)
        b = _splitmix64(ctr + np.uint64(1))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) / 9007199254740993.0   # (0,1)
    u2 = (b >> np.uint64(11)).astype(np.float64) / 9007199254740992.0           # [0,1)
    z = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)
    return (mean + std * z).astype(np.float32).reshape(shape)


def synth_normal_fast(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    """Same contract as synth_normal but drawn from numpy's PCG64 stream (~100x faster): for the full-size
    benchmark models only — golden fixtures and parity tests use the counter-based generator above."""
    h = int.from_bytes(hashlib.sha256(f"{seed}:{name}".encode()).digest()[:8], "little")
    rng = np.random.default_rng([seed, h])
    n = int(np.prod(shape)) if len(shape) else 1
    z = rng.standard_normal(n, dtype=np.float32)
    z *= np.float32(std)
    if mean:
        z += np.float32(mean)
    return z.reshape(shape)


# --------------------------------------------------------------------------------------
# BigVGAN
# --------------------------------------------------------------------------------------
def bigvgan_spec(cfg: BigVGANConfig) -> Spec:
    """Canonical tensor order == order of the fp32 blob given to mi_bigvgan_create.

    Layouts are PyTorch-native: Conv1d (Cout, Cin, k); ConvTranspose1d (Cin, Cout, k).
    """
    s: Spec = []
    c0 = cfg.upsample_initial_channel
    if cfg.pre_layernorm:        # IndexTTS graph F: gpt.final_norm in front of the vocoder
        s.append(("final_norm.weight", (cfg.num_mels,), "norm_w"))
        s.append(("final_norm.bias", (cfg.num_mels,), "bias"))
    s.append(("conv_pre.weight", (c0, cfg.num_mels, 7), "conv_pre"))
    s.append(("conv_pre.bias", (c0,), "bias"))
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        s.append((f"ups.{i}.0.weight", (cin, cout, k), "convt"))
        s.append((f"ups.{i}.0.bias", (cout,), "bias"))
        for j, (rk, dil) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
            n = i * cfg.num_kernels + j
            for l in range(len(dil)):
                s.append((f"resblocks.{n}.convs1.{l}.weight", (cout, cout, rk), "conv"))
                s.append((f"resblocks.{n}.convs1.{l}.bias", (cout,), "bias"))
                s.append((f"resblocks.{n}.convs2.{l}.weight", (cout, cout, rk), "conv_res"))
                s.append((f"resblocks.{n}.convs2.{l}.bias", (cout,), "bias"))
            for m in range(2 * len(dil)):
                s.append((f"resblocks.{n}.activations.{m}.act.alpha", (cout,), "snake"))
                s.append((f"resblocks.{n}.activations.{m}.act.beta", (cout,), "snake"))
    cl = cfg.stage_channels(cfg.num_upsamples - 1)
    s.append(("activation_post.act.alpha", (cl,), "snake"))
    s.append(("activation_post.act.beta", (cl,), "snake"))
    s.append(("conv_post.weight", (1, cl, 7), "conv_post"))
    if cfg.use_bias_at_final:
        s.append(("conv_post.bias", (1,), "bias"))
    return s


# gains keep activations O(1) through the stack and the pre-tanh signal well inside (-1, 1)
_GAIN = {"conv": 1.0, "convt": 1.0, "linear": 1.0, "conv_pre": 0.35, "conv_res": 0.4, "conv_post": 0.3}


def _fan_in(shape, kind: str) -> int:
    if kind in ("conv", "conv_pre", "conv_res", "conv_post"):
        return shape[1] * shape[2]
    if kind == "convt":      # each output sample sees Cin * k/stride taps (k = 2*stride, or k = stride in IndexTTS)
        return shape[0] * 2
    if kind in ("linear",):
        return shape[1]
    return 1


def synth_tensor(seed: int, name: str, shape, kind: str, fast: bool = False) -> np.ndarray:
    """Fan-in scaled init so activations stay O(1) through the stack (so the int16
    output is neither silent nor saturated; see DESIGN.md 'synthetic weights')."""
    synth_normal = globals()["synth_normal_fast" if fast else "synth_normal"]
    if kind in _GAIN:
        return synth_normal(seed, name, shape, std=_GAIN[kind] / math.sqrt(_fan_in(shape, kind)))
    if kind == "bias":
        return synth_normal(seed, name, shape, std=0.02)
    if kind == "snake":
        return synth_normal(seed, name, shape, std=0.3)
    if kind == "embed":
        return synth_normal(seed, name, shape, std=1.0)
    if kind == "embed_s":    # GPT token / position tables (upstream init std 0.02; larger so the tables matter)
        return synth_normal(seed, name, shape, std=0.5)
    if kind == "linear_t":   # HF Conv1D: (in, out)
        return synth_normal(seed, name, shape, std=1.0 / math.sqrt(shape[0]))
    if kind == "linear_t_res":
        return synth_normal(seed, name, shape, std=0.5 / math.sqrt(shape[0]))
    if kind == "norm_w":
        return synth_normal(seed, name, shape, std=0.1, mean=1.0)
    if kind == "mod":        # AdaLN modulation linears (upstream zero-inits these: dit.py:156-166)
        return synth_normal(seed, name, shape, std=0.5 / math.sqrt(shape[1]))
    if kind == "conv2d":     # (out, in, kh, kw)
        return synth_normal(seed, name, shape, std=1.0 / math.sqrt(shape[1] * shape[2] * shape[3]))
    if kind == "linear_res":  # last linear of a residual branch
        return synth_normal(seed, name, shape, std=0.5 / math.sqrt(shape[1]))
    if kind == "bias_big":   # position biases / BatchNorm running means: large enough to matter
        return synth_normal(seed, name, shape, std=0.3)
    if kind == "var":        # BatchNorm running variance: positive, around one
        return (0.5 + np.abs(synth_normal(seed, name, shape, std=0.5))).astype(np.float32)
    if kind == "gamma":      # layer-scale / GRN gamma
        return synth_normal(seed, name, shape, std=0.1, mean=0.5)
    raise ValueError(kind)


def synth_state(spec: Spec, seed: int = 9527, fast: bool = False) -> "OrderedDict[str, np.ndarray]":
    return OrderedDict((name, synth_tensor(seed, name, shape, kind, fast)) for name, shape, kind in spec)


def pack_state(spec: Spec, state: Dict[str, np.ndarray]) -> np.ndarray:
    parts = []
    for name, shape, _ in spec:
        t = np.asarray(state[name], dtype=np.float32)
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {t.shape} != spec {shape}")
        parts.append(t.reshape(-1))
    return np.ascontiguousarray(np.concatenate(parts))


def pack_bigvgan(cfg: BigVGANConfig, state: Dict[str, np.ndarray]) -> np.ndarray:
    """state_dict (weight-norm already removed) -> canonical fp32 blob."""
    return pack_state(bigvgan_spec(cfg), state)


def remove_weight_norm_state(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """weight_g / weight_v pairs -> plain weights (what ``remove_weight_norm`` does,
    BigVGAN/modeling_modified/bigvgan.py:412-424; dim=0 norm over the remaining axes)."""
    out = {}
    for k, v in state.items():
        if k.endswith(".weight_g"):
            base = k[: -len("_g")]
            vv = np.asarray(state[base + "_v"], dtype=np.float64)
            g = np.asarray(v, dtype=np.float64)
            nrm = np.sqrt((vv ** 2).sum(axis=tuple(range(1, vv.ndim)), keepdims=True))
            out[base] = (vv * (g / nrm)).astype(np.float32)
        elif k.endswith(".weight_v"):
            continue
        else:
            out[k] = np.asarray(v)
    return out


# --------------------------------------------------------------------------------------
# F5 (DiT + text embedding + Vocos)
# --------------------------------------------------------------------------------------
def f5_spec(cfg: F5Config) -> Spec:
    """Canonical order of the F5 blob (names = upstream f5_tts / vocos state_dict keys)."""
    d, td, ff = cfg.dim, cfg.text_dim, cfg.ff_dim
    s: Spec = []
    # --- time embedding (F5_TTS/modeling_modified/F5/modules.py:687-698)
    s += [("transformer.time_embed.time_mlp.0.weight", (d, cfg.freq_embed_dim), "linear"),
          ("transformer.time_embed.time_mlp.0.bias", (d,), "bias"),
          ("transformer.time_embed.time_mlp.2.weight", (d, d), "linear"),
          ("transformer.time_embed.time_mlp.2.bias", (d,), "bias")]
    # --- text embedding (dit.py:33-73)
    s.append(("transformer.text_embed.text_embed.weight", (cfg.text_num_embeds + 1, td), "embed"))
    for l in range(cfg.conv_layers):
        p = f"transformer.text_embed.text_blocks.{l}."
        s += [(p + "dwconv.weight", (td, 1, 7), "conv"), (p + "dwconv.bias", (td,), "bias"),
              (p + "norm.weight", (td,), "norm_w"), (p + "norm.bias", (td,), "bias"),
              (p + "pwconv1.weight", (td * cfg.conv_mult, td), "linear"),
              (p + "pwconv1.bias", (td * cfg.conv_mult,), "bias"),
              (p + "grn.gamma", (1, 1, td * cfg.conv_mult), "gamma"),
              (p + "grn.beta", (1, 1, td * cfg.conv_mult), "bias"),
              (p + "pwconv2.weight", (td, td * cfg.conv_mult), "linear"),
              (p + "pwconv2.bias", (td,), "bias")]
    # --- input embedding (dit.py:79-87, modules.py:167-190)
    cin = cfg.mel_dim * 2 + td
    g = cfg.pos_conv_groups
    s += [("transformer.input_embed.proj.weight", (d, cin), "linear"),
          ("transformer.input_embed.proj.bias", (d,), "bias"),
          ("transformer.input_embed.conv_pos_embed.conv1d.0.weight", (d, d // g, cfg.pos_conv_kernel), "conv"),
          ("transformer.input_embed.conv_pos_embed.conv1d.0.bias", (d,), "bias"),
          ("transformer.input_embed.conv_pos_embed.conv1d.2.weight", (d, d // g, cfg.pos_conv_kernel), "conv"),
          ("transformer.input_embed.conv_pos_embed.conv1d.2.bias", (d,), "bias")]
    # --- DiT blocks (modules.py:575-613)
    for i in range(cfg.depth):
        p = f"transformer.transformer_blocks.{i}."
        s += [(p + "attn_norm.linear.weight", (6 * d, d), "mod"), (p + "attn_norm.linear.bias", (6 * d,), "bias"),
              (p + "attn.to_q.weight", (d, d), "linear"), (p + "attn.to_q.bias", (d,), "bias"),
              (p + "attn.to_k.weight", (d, d), "linear"), (p + "attn.to_k.bias", (d,), "bias"),
              (p + "attn.to_v.weight", (d, d), "linear"), (p + "attn.to_v.bias", (d,), "bias"),
              (p + "attn.to_out.0.weight", (d, d), "linear"), (p + "attn.to_out.0.bias", (d,), "bias"),
              (p + "ff.ff.0.0.weight", (ff, d), "linear"), (p + "ff.ff.0.0.bias", (ff,), "bias"),
              (p + "ff.ff.2.weight", (d, ff), "linear"), (p + "ff.ff.2.bias", (d,), "bias")]
    s += [("transformer.norm_out.linear.weight", (2 * d, d), "mod"),
          ("transformer.norm_out.linear.bias", (2 * d,), "bias"),
          ("transformer.proj_out.weight", (cfg.mel_dim, d), "linear"),
          ("transformer.proj_out.bias", (cfg.mel_dim,), "bias")]
    # --- Vocos (vocos/models.py:44-83, modules.py:20-51, heads.py:39-59)
    vd, vi = cfg.vocos_dim, cfg.vocos_intermediate
    s += [("vocos.backbone.embed.weight", (vd, cfg.mel_dim, 7), "conv"), ("vocos.backbone.embed.bias", (vd,), "bias"),
          ("vocos.backbone.norm.weight", (vd,), "norm_w"), ("vocos.backbone.norm.bias", (vd,), "bias")]
    for l in range(cfg.vocos_layers):
        p = f"vocos.backbone.convnext.{l}."
        s += [(p + "dwconv.weight", (vd, 1, 7), "conv"), (p + "dwconv.bias", (vd,), "bias"),
              (p + "norm.weight", (vd,), "norm_w"), (p + "norm.bias", (vd,), "bias"),
              (p + "pwconv1.weight", (vi, vd), "linear"), (p + "pwconv1.bias", (vi,), "bias"),
              (p + "pwconv2.weight", (vd, vi), "linear"), (p + "pwconv2.bias", (vd,), "bias"),
              (p + "gamma", (vd,), "gamma")]
    s += [("vocos.backbone.final_layer_norm.weight", (vd,), "norm_w"),
          ("vocos.backbone.final_layer_norm.bias", (vd,), "bias"),
          ("vocos.head.out.weight", (cfg.n_fft + 2, vd), "linear"),
          ("vocos.head.out.bias", (cfg.n_fft + 2,), "bias")]
    return s


def f5_packed_spec(cfg: F5Config) -> Spec:
    """Blob order after folding: vocos gammas are folded away."""
    return [e for e in f5_spec(cfg) if not (e[0].startswith("vocos.") and e[0].endswith(".gamma"))]


def fold_f5(cfg: F5Config, state: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Apply the export-time folds (see module docstring) and return a new dict whose
    keys are those of ``f5_packed_spec``."""
    st = OrderedDict((k, np.array(v, dtype=np.float32, copy=True)) for k, v in state.items())
    sf = math.pow(cfg.dim_head, -0.25)
    if getattr(cfg, "ref_fp16_attn", False):
        sf *= 0.1                                 # Export_F5.py:322-324 ("To avoid overflow in float16 format")
    sf = np.float32(sf)
    for i in range(cfg.depth):
        p = f"transformer.transformer_blocks.{i}.attn."
        for nm in ("to_q", "to_k"):
            st[p + nm + ".weight"] = st[p + nm + ".weight"] * sf
            st[p + nm + ".bias"] = st[p + nm + ".bias"] * sf
    rt = np.float32(math.sqrt(cfg.vocos_dim))
    st["vocos.backbone.norm.weight"] = st["vocos.backbone.norm.weight"] * rt
    st["vocos.backbone.final_layer_norm.weight"] = st["vocos.backbone.final_layer_norm.weight"] * rt
    for l in range(cfg.vocos_layers):
        p = f"vocos.backbone.convnext.{l}."
        st[p + "norm.weight"] = st[p + "norm.weight"] * rt
        gamma = st.pop(p + "gamma")
        st[p + "pwconv2.weight"] = gamma[:, None] * st[p + "pwconv2.weight"]
        st[p + "pwconv2.bias"] = gamma * st[p + "pwconv2.bias"]
    return st


def pack_f5(cfg: F5Config, state: Dict[str, np.ndarray]) -> np.ndarray:
    """Unfolded upstream-style state dict -> folded canonical fp32 blob."""
    return pack_state(f5_packed_spec(cfg), fold_f5(cfg, state))


# --------------------------------------------------------------------------------------
# IndexTTS acoustic GPT-2 (graphs B..E)
# --------------------------------------------------------------------------------------
def gpt_spec(cfg: IndexGPTConfig) -> Spec:
    """Upstream-named state of ``indexTTS.gpt`` as the export wrappers read it (IndexTTS/Export_IndexTTS.py:203-289).
    ``c_attn / c_proj / c_fc`` are HF ``Conv1D`` modules: weight is (in, out)."""
    h, n = cfg.hidden, cfg.inner
    s: Spec = [("text_embedding.weight", (cfg.text_tokens, h), "embed_s"),
               ("text_pos_embedding.emb.weight", (cfg.max_text_pos, h), "embed_s"),
               ("inference_model.embeddings.weight", (cfg.mel_codes, h), "embed_s"),
               ("inference_model.text_pos_embedding.emb.weight", (cfg.max_mel_pos, h), "embed_s")]
    for i in range(cfg.layers):
        p = f"inference_model.transformer.h.{i}."
        s += [(p + "ln_1.weight", (h,), "norm_w"), (p + "ln_1.bias", (h,), "bias"),
              (p + "attn.c_attn.weight", (h, 3 * h), "linear_t"), (p + "attn.c_attn.bias", (3 * h,), "bias"),
              (p + "attn.c_proj.weight", (h, h), "linear_t_res"), (p + "attn.c_proj.bias", (h,), "bias"),
              (p + "ln_2.weight", (h,), "norm_w"), (p + "ln_2.bias", (h,), "bias"),
              (p + "mlp.c_fc.weight", (h, n), "linear_t"), (p + "mlp.c_fc.bias", (n,), "bias"),
              (p + "mlp.c_proj.weight", (n, h), "linear_t_res"), (p + "mlp.c_proj.bias", (h,), "bias")]
    s += [("inference_model.transformer.ln_f.weight", (h,), "norm_w"),
          ("inference_model.transformer.ln_f.bias", (h,), "bias"),
          ("inference_model.lm_head.0.weight", (h,), "norm_w"), ("inference_model.lm_head.0.bias", (h,), "bias"),
          ("inference_model.lm_head.1.weight", (cfg.mel_codes, h), "linear"),
          ("inference_model.lm_head.1.bias", (cfg.mel_codes,), "bias")]
    return s


def fold_gpt(cfg: IndexGPTConfig, state: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Export-time folds of IndexTTS_E.__init__ (Export_IndexTTS.py:252-268): Conv1D weights transposed to
    (out, in) rows, q and k rows (weight and bias) scaled by head_dim**-0.25.  Keys/order stay those of gpt_spec;
    the transposed tensors keep their names."""
    st = OrderedDict((k, np.array(v, dtype=np.float32, copy=True)) for k, v in state.items())
    sc = np.float32(float(cfg.head_dim) ** -0.25)
    h = cfg.hidden
    for i in range(cfg.layers):
        p = f"inference_model.transformer.h.{i}."
        w = np.ascontiguousarray(st[p + "attn.c_attn.weight"].T)
        b = st[p + "attn.c_attn.bias"]
        w[: 2 * h] *= sc
        b[: 2 * h] *= sc
        st[p + "attn.c_attn.weight"] = w
        for nm in ("attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight"):
            st[p + nm] = np.ascontiguousarray(st[p + nm].T)
    return st


def pack_gpt(cfg: IndexGPTConfig, state: Dict[str, np.ndarray]) -> np.ndarray:
    st = fold_gpt(cfg, state)
    return np.ascontiguousarray(np.concatenate([st[name].reshape(-1) for name, _, _ in gpt_spec(cfg)]))


# ---------------------------------------------------------------------------------------------------------------------
# IndexTTS graph A: conditioning encoder + perceiver (indexTTS.gpt.*) and ECAPA speaker encoder + cond layers (indexTTS.bigvgan.*)
# ---------------------------------------------------------------------------------------------------------------------
def cond_spec(cfg: IndexCondConfig) -> Spec:
    """Upstream-named state the export wrapper IndexTTS_A reads (IndexTTS/Export_IndexTTS.py:74-200): ``gpt.conditioning_encoder``
    (wenet-style Conformer: Conv2dSubsampling2 + rel-pos blocks with a conv module), ``gpt.perceiver_encoder`` (lucidrains-style
    PerceiverResampler) and ``bigvgan.speaker_encoder`` (speechbrain-style ECAPA-TDNN: wrapped Conv1d / BatchNorm1d modules,
    hence the doubled ``conv.conv`` / ``norm.norm`` names) + ``bigvgan.cond_layer`` / ``bigvgan.conds``.  The sub-module
    definitions are un-vendored (no source in the reference tree): names and shapes follow the published packages."""
    d, h, lin, k = cfg.enc_dim, cfg.enc_heads, cfg.enc_linear, cfg.enc_kernel
    s: Spec = [("audio_pad", (cfg.audio_pad,), "embed"),                    # the wrapper's torch.randn(0.1 s) constant (:94)
               ("gpt.conditioning_encoder.embed.conv.0.weight", (d, 1, 3, 3), "conv2d"),
               ("gpt.conditioning_encoder.embed.conv.0.bias", (d,), "bias"),
               ("gpt.conditioning_encoder.embed.out.0.weight", (d, d * cfg.sub_freq), "linear"),
               ("gpt.conditioning_encoder.embed.out.0.bias", (d,), "bias")]
    for i in range(cfg.enc_blocks):
        p = f"gpt.conditioning_encoder.encoders.{i}."
        for n in ("norm_mha", "norm_conv", "norm_ff", "norm_final", "conv_module.norm"):
            s += [(p + n + ".weight", (d,), "norm_w"), (p + n + ".bias", (d,), "bias")]
        for n in ("linear_q", "linear_k", "linear_v", "linear_out"):
            s += [(p + f"self_attn.{n}.weight", (d, d), "linear_res" if n == "linear_out" else "linear"), (p + f"self_attn.{n}.bias", (d,), "bias")]
        s += [(p + "self_attn.linear_pos.weight", (d, d), "linear"),
              (p + "self_attn.pos_bias_u", (h, cfg.enc_dk), "bias_big"), (p + "self_attn.pos_bias_v", (h, cfg.enc_dk), "bias_big"),
              (p + "conv_module.pointwise_conv1.weight", (2 * d, d, 1), "conv"), (p + "conv_module.pointwise_conv1.bias", (2 * d,), "bias"),
              (p + "conv_module.depthwise_conv.weight", (d, 1, k), "conv"), (p + "conv_module.depthwise_conv.bias", (d,), "bias"),
              (p + "conv_module.pointwise_conv2.weight", (d, d, 1), "conv_res"), (p + "conv_module.pointwise_conv2.bias", (d,), "bias"),
              (p + "feed_forward.w_1.weight", (lin, d), "linear"), (p + "feed_forward.w_1.bias", (lin,), "bias"),
              (p + "feed_forward.w_2.weight", (d, lin), "linear_res"), (p + "feed_forward.w_2.bias", (d,), "bias")]
    s += [("gpt.conditioning_encoder.after_norm.weight", (d,), "norm_w"), ("gpt.conditioning_encoder.after_norm.bias", (d,), "bias")]
    D, inner, ffi = cfg.model_dim, cfg.perc_inner, cfg.perc_ff
    s += [("gpt.perceiver_encoder.proj_context.weight", (D, d), "linear"), ("gpt.perceiver_encoder.proj_context.bias", (D,), "bias"),
          ("gpt.perceiver_encoder.latents", (cfg.latents, D), "embed")]
    for j in range(cfg.perc_depth):
        p = f"gpt.perceiver_encoder.layers.{j}."
        s += [(p + "0.to_q.weight", (inner, D), "linear"), (p + "0.to_kv.weight", (2 * inner, D), "linear"),
              (p + "0.to_out.weight", (D, inner), "linear_res"),
              (p + "1.0.weight", (2 * ffi, D), "linear"), (p + "1.0.bias", (2 * ffi,), "bias"),
              (p + "1.2.weight", (D, ffi), "linear_res"), (p + "1.2.bias", (D,), "bias")]
    s += [("gpt.perceiver_encoder.norm.gamma", (D,), "norm_w")]

    def tdnn(prefix, cin, cout, kk):
        return [(prefix + "conv.conv.weight", (cout, cin, kk), "conv"), (prefix + "conv.conv.bias", (cout,), "bias"),
                (prefix + "norm.norm.weight", (cout,), "norm_w"), (prefix + "norm.norm.bias", (cout,), "bias"),
                (prefix + "norm.norm.running_mean", (cout,), "bias_big"), (prefix + "norm.norm.running_var", (cout,), "var")]
    ch, ks = cfg.spk_channels, cfg.spk_kernels
    e = "bigvgan.speaker_encoder."
    s += tdnn(e + "blocks.0.", cfg.n_mels, ch[0], ks[0])
    for i in range(1, len(ch) - 1):
        p = e + f"blocks.{i}."
        c, sc = ch[i], cfg.spk_res2net_scale
        s += tdnn(p + "tdnn1.", ch[i - 1], c, 1)
        for r in range(sc - 1):
            s += tdnn(p + f"res2net_block.blocks.{r}.", c // sc, c // sc, ks[i])
        s += tdnn(p + "tdnn2.", c, c, 1)
        s += [(p + "se_block.conv1.conv.weight", (cfg.spk_se, c, 1), "conv"), (p + "se_block.conv1.conv.bias", (cfg.spk_se,), "bias"),
              (p + "se_block.conv2.conv.weight", (c, cfg.spk_se, 1), "conv"), (p + "se_block.conv2.conv.bias", (c,), "bias")]
    cm = ch[-1]
    s += tdnn(e + "mfa.", cm, cm, ks[-1])
    s += tdnn(e + "asp.tdnn.", 3 * cm, cfg.spk_att, 1)
    s += [(e + "asp.conv.conv.weight", (cm, cfg.spk_att, 1), "conv"), (e + "asp.conv.conv.bias", (cm,), "bias"),
          (e + "asp_bn.norm.weight", (2 * cm,), "norm_w"), (e + "asp_bn.norm.bias", (2 * cm,), "bias"),
          (e + "asp_bn.norm.running_mean", (2 * cm,), "bias_big"), (e + "asp_bn.norm.running_var", (2 * cm,), "var"),
          (e + "fc.conv.weight", (cfg.spk_embed, 2 * cm, 1), "conv"), (e + "fc.conv.bias", (cfg.spk_embed,), "bias"),
          ("bigvgan.cond_layer.weight", (cfg.voc_initial, cfg.spk_embed, 1), "conv"), ("bigvgan.cond_layer.bias", (cfg.voc_initial,), "bias")]
    for i, c in enumerate(cfg.voc_channels):
        s += [(f"bigvgan.conds.{i}.weight", (c, cfg.spk_embed, 1), "conv"), (f"bigvgan.conds.{i}.bias", (c,), "bias")]
    return s


def fold_cond(cfg: IndexCondConfig, state: Dict[str, np.ndarray]) -> "OrderedDict[str, np.ndarray]":
    """Export-time folds of IndexTTS_A.__init__ (Export_IndexTTS.py:88-129) that change VALUES: embed.out weight and bias times
    xscale = sqrt(enc_dim) (:88-89); per encoder layer linear_q / linear_k (weight, bias), linear_pos (weight) and
    pos_bias_u / pos_bias_v times d_k**-0.25 (:97-104); perceiver to_q and the K half of to_kv times dim_head**-0.25 (:121-125).
    The wrapper's per-head weight views (:106-129) only re-index the same numbers and are not applied here: the oracle and
    the engine contract over the same (head, d) axes of the unsplit matrices."""
    st = OrderedDict((k, np.array(v, dtype=np.float32, copy=True)) for k, v in state.items())
    xs = np.float32(math.sqrt(cfg.enc_dim))
    st["gpt.conditioning_encoder.embed.out.0.weight"] *= xs
    st["gpt.conditioning_encoder.embed.out.0.bias"] *= xs
    sc = np.float32(float(cfg.enc_dk) ** -0.25)
    for i in range(cfg.enc_blocks):
        p = f"gpt.conditioning_encoder.encoders.{i}.self_attn."
        for n in ("linear_q.weight", "linear_q.bias", "linear_k.weight", "linear_k.bias", "linear_pos.weight", "pos_bias_u", "pos_bias_v"):
            st[p + n] *= sc
    ps = np.float32(float(cfg.perc_dim_head) ** -0.25)
    for j in range(cfg.perc_depth):
        p = f"gpt.perceiver_encoder.layers.{j}.0."
        st[p + "to_q.weight"] *= ps
        st[p + "to_kv.weight"][: cfg.perc_inner] *= ps
    return st


def pack_cond(cfg: IndexCondConfig, state: Dict[str, np.ndarray]) -> np.ndarray:
    """cond_spec order, folded values: the blob mi_indextts_cond_create takes."""
    return pack_state(cond_spec(cfg), fold_cond(cfg, state))


def synth_vocab(n: int = 2545) -> Dict[str, int]:
    """Synthetic vocab.txt stand-in: line i holds one character (F5-TTS-ONNX-Inference.py:88-92
    maps ``char[:-1] -> line index``).  Index 0 is the space, like upstream's vocab."""
    chars = [" "] + [chr(c) for c in range(33, 127)]
    i = 0x4E00
    while len(chars) < n:
        chars.append(chr(i))
        i += 1
    return {c: k for k, c in enumerate(chars[:n])}


# ---------------------------------------------------------------------------------------------
# Synthetic workload inputs shared by bench.py, the -m gpu tests and the golden generators
# ---------------------------------------------------------------------------------------------
F5_BENCH_REF_TEXT = "Some call me nature, others call me mother nature, I am the breeze and rain. "
F5_BENCH_GEN_TEXT = "The quick brown fox jumps over the lazy dog while seven wizards brew a potion"


def f5_synthetic_inputs(cfg: F5Config, U: int, rank: int = 0, L: int = 144000, first: int = None):
    """BASELINE configs[2]/[3]: 6.0 s reference audio (144000 samples -> 563 frames), equal-length ~15-word ASCII
    ref/gen texts (-> N = 1126 by the duration formula of F5-TTS-ONNX-Inference.py:227-231), char-level ids against
    the synthetic vocab, injected noise.  Returns (audio (U,L) i16, ids (U,T) i32, N, noise (U,N,mel) f32).
    (`L` other than 144000 is for reduced-size plumbing tests only.)  Utterance u of the call is utterance `first + u` of the
    job's list (seed 9527 + first + u: configs[3] is seeds 9527 .. 9590 partitioned over the ranks with shard.shard_range);
    without `first` it is 64 * rank + u, the numbering the fixtures were generated with."""
    base = 64 * rank if first is None else int(first)
    ref_text = F5_BENCH_REF_TEXT
    gen_text = (F5_BENCH_GEN_TEXT + " " * len(ref_text))[:len(ref_text)]
    vocab = synth_vocab(cfg.text_num_embeds)
    ids = np.asarray([vocab.get(c, 0) for c in (ref_text + gen_text)], dtype=np.int32)
    ref_frames = L // cfg.hop_length + 1
    N = ref_frames + int(ref_frames / len(ref_text.encode()) * len(gen_text.encode()) / 1.0)
    audio = np.empty((U, L), np.int16)
    tt = np.arange(L) / cfg.sample_rate
    for u in range(U):
        a = 0.1 * 32767 * np.sin(2 * np.pi * 220 * tt) + synth_normal(9527 + base + u, "audio", (L,), std=500.0)
        audio[u] = np.clip(np.round(a), -32768, 32767).astype(np.int16)
    noise = np.stack([synth_normal(9527 + base + u, "noise", (N, cfg.mel_dim)) for u in range(U)])
    return audio, np.tile(ids[None], (U, 1)), N, noise


def bigvgan_synthetic_mel(cfg: BigVGANConfig, B: int, F: int, rank: int = 0) -> np.ndarray:
    """BASELINE configs[0]/[1]: a log-mel shaped (B, num_mels, F) input, N(-2, 2) clipped to [-11.5, 2.5]."""
    return synth_normal(100 + rank, "mel", (B, cfg.num_mels, F), std=2.0, mean=-2.0).clip(-11.5, 2.5)
