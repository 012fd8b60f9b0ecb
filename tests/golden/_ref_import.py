"""Import the reference's PyTorch module code from /root/reference with ``sys.modules`` stubs.

Used ONLY by tests/golden/make_golden.py in the build container (where /root/reference is
mounted) to produce the committed fixtures.  Nothing here is copied from the reference: the
reference files are imported where they lie.  Pieces the reference imports from un-vendored
packages (NVIDIA BigVGAN ``activations``/``utils``/``env``; torchaudio; x_transformers;
librosa; vocos.spectral_ops) are stubbed or restated from their published definitions
(SURVEY.md §8c).
"""
from __future__ import annotations

import importlib.util
import math
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def _load(name: str, path: str):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def _pkg(name: str):
    if name not in sys.modules:
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
    return sys.modules[name]


# ------------------------------------------------------------------------------------------
# BigVGAN
# ------------------------------------------------------------------------------------------
class _SnakeBeta(nn.Module):
    """NVIDIA/BigVGAN activations.SnakeBeta (published definition; un-vendored in the reference)."""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.alpha_logscale = alpha_logscale
        if alpha_logscale:
            self.alpha = nn.Parameter(torch.zeros(in_features) * alpha)
            self.beta = nn.Parameter(torch.zeros(in_features) * alpha)
        else:
            self.alpha = nn.Parameter(torch.ones(in_features) * alpha)
            self.beta = nn.Parameter(torch.ones(in_features) * alpha)
        self.no_div_by_zero = 0.000000001

    def forward(self, x):
        alpha = self.alpha.unsqueeze(0).unsqueeze(-1)
        beta = self.beta.unsqueeze(0).unsqueeze(-1)
        if self.alpha_logscale:
            alpha = torch.exp(alpha)
            beta = torch.exp(beta)
        return x + (1.0 / (beta + self.no_div_by_zero)) * torch.pow(torch.sin(x * alpha), 2)


class _AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


def load_bigvgan_ref():
    """Returns the reference ``bigvgan`` module (BigVGAN/modeling_modified/bigvgan.py)."""
    act = types.ModuleType("activations")
    act.SnakeBeta = _SnakeBeta
    act.Snake = _SnakeBeta
    sys.modules["activations"] = act
    utils = types.ModuleType("utils")
    utils.init_weights = lambda m, mean=0.0, std=0.01: None
    utils.get_padding = lambda k, d=1: int((k * d - d) / 2)
    sys.modules["utils"] = utils
    env = types.ModuleType("env")
    env.AttrDict = _AttrDict
    sys.modules["env"] = env
    _pkg("alias_free_activation")
    _pkg("alias_free_activation.torch")
    d = REF + "/BigVGAN/modeling_modified/"
    _load("alias_free_activation.torch.filter", d + "filter.py")
    _load("alias_free_activation.torch.resample", d + "resample.py")
    _load("alias_free_activation.torch.act", d + "act.py")
    return _load("bigvgan", d + "bigvgan.py")


def bigvgan_hparams(cfg) -> _AttrDict:
    return _AttrDict(
        num_mels=cfg.num_mels, upsample_initial_channel=cfg.upsample_initial_channel,
        upsample_rates=list(cfg.upsample_rates), upsample_kernel_sizes=list(cfg.upsample_kernel_sizes),
        resblock="1", resblock_kernel_sizes=list(cfg.resblock_kernel_sizes),
        resblock_dilation_sizes=[list(d) for d in cfg.resblock_dilation_sizes],
        activation="snakebeta", snake_logscale=cfg.snake_logscale,
        use_bias_at_final=cfg.use_bias_at_final, use_tanh_at_final=cfg.use_tanh_at_final)


def exec_lines(path: str, start: int, end: int, ns: dict, replace=()):
    """exec source lines [start, end] (1-based, inclusive) of a reference script that cannot be
    imported as a module (import-time side effects: Export_*.py)."""
    import textwrap
    with open(path, "r", encoding="utf-8") as f:
        src = textwrap.dedent("".join(f.readlines()[start - 1:end]))
    for a, b in replace:
        assert a in src, a
        src = src.replace(a, b)
    exec(compile(src, f"{path}:{start}-{end}", "exec"), ns)
    return ns


# ------------------------------------------------------------------------------------------
# F5 (DiT, Vocos, STFT_Process)
# ------------------------------------------------------------------------------------------
def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate, norm=None, mel_scale="htk"):
    """torchaudio.functional.melscale_fbanks restated from its published definition (HTK, norm=None);
    un-vendored in the reference (Export_F5.py:113)."""
    assert norm is None and mel_scale == "htk"
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + (f_min / 700.0))
    m_max = 2595.0 * math.log10(1.0 + (f_max / 700.0))
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    zero = torch.zeros(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    return torch.max(zero, torch.min(down_slopes, up_slopes))


def slaney_mel_basis(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kw):
    """librosa.filters.mel (htk=False, norm='slaney') restated from its published definition; librosa is not installed and the
    reference's bigvgan-type mel front end calls it (modeling_modified/F5/modules.py:18,45).  Cross-checked in
    make_golden_bigvgan_mel.py against transformers.audio_utils.mel_filter_bank(norm='slaney', mel_scale='slaney')."""
    import numpy as np
    assert not kw.get("htk", False) and kw.get("norm", "slaney") == "slaney"
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp

    def hz_to_mel(f):
        f = np.asanyarray(f, dtype=np.float64)
        lin = f / f_sp
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, lin)

    def mel_to_hz(m):
        m = np.asanyarray(m, dtype=np.float64)
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)

    fftfreqs = np.linspace(0, float(sr) / 2, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights.astype(np.float32)


def load_f5_ref(fp16: bool = False):
    """Returns (modules, dit, vocos_models, vocos_heads, STFT_Process) from the reference files.  ``fp16``: the
    reference's fp16-transformer variant of modules.py (F5/fp16/modules.py, which Export_F5.py:88-89 installs over
    f5_tts/model/modules.py when use_fp16_transformer is set)."""
    ort = types.ModuleType("onnxruntime")
    sys.modules["onnxruntime"] = ort
    ta = _pkg("torchaudio")
    taf = _pkg("torchaudio.functional")
    taff = _pkg("torchaudio.functional.functional")
    taff._hz_to_mel = lambda *a, **k: 0.0
    taff._mel_to_hz = lambda *a, **k: 0.0
    taf.melscale_fbanks = melscale_fbanks
    taf.functional = taff
    ta.functional = taf
    _pkg("librosa")
    lf = _pkg("librosa.filters")
    lf.mel = slaney_mel_basis          # (the bigvgan-type mel front end of modules.py:30-72 calls it; nothing else does)
    _pkg("x_transformers")
    xt = _pkg("x_transformers.x_transformers")
    xt.apply_rotary_pos_emb = lambda *a, **k: None

    class RotaryEmbedding(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    xt.RotaryEmbedding = RotaryEmbedding
    _pkg("f5_tts")
    _pkg("f5_tts.model")
    _pkg("f5_tts.model.backbones")
    d = REF + "/F5_TTS/modeling_modified/"
    modules = _load("f5_tts.model.modules", d + ("F5/fp16/modules.py" if fp16 else "F5/modules.py"))
    dit = _load("f5_tts.model.backbones.dit", d + "F5/dit.py")
    _pkg("vocos")
    so = _pkg("vocos.spectral_ops")

    class _Nop(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    so.ISTFT = _Nop
    so.IMDCT = _Nop
    vmod = _load("vocos.modules", d + "vocos/modules.py")
    vheads = _load("vocos.heads", d + "vocos/heads.py")
    vmodels = _load("vocos.models", d + "vocos/models.py")
    stft = _load("STFT_Process", REF + "/F5_TTS/STFT_Process.py")
    return modules, dit, vmodels, vheads, stft


# ------------------------------------------------------------------------------------------
# IndexTTS vocoder (graph F)
# ------------------------------------------------------------------------------------------
def load_indextts_bigvgan_ref():
    """Returns the reference IndexTTS ``models`` module (IndexTTS/modeling_modified/models.py) with its alias-free
    activation modules; un-vendored ``indextts.BigVGAN.{activations, ECAPA_TDNN, utils}`` are stubbed / restated."""
    _pkg("indextts")
    bvp = _pkg("indextts.BigVGAN")
    act = types.ModuleType("indextts.BigVGAN.activations")
    act.SnakeBeta = _SnakeBeta
    act.Snake = _SnakeBeta
    sys.modules["indextts.BigVGAN.activations"] = act
    bvp.activations = act
    ec = types.ModuleType("indextts.BigVGAN.ECAPA_TDNN")

    class ECAPA_TDNN(nn.Module):          # speaker encoder: belongs to graph A, not exercised by graph F
        def __init__(self, *a, **k):
            super().__init__()
    ec.ECAPA_TDNN = ECAPA_TDNN
    sys.modules["indextts.BigVGAN.ECAPA_TDNN"] = ec
    ut = types.ModuleType("indextts.BigVGAN.utils")
    ut.init_weights = lambda m, mean=0.0, std=0.01: None
    ut.get_padding = lambda k, d=1: int((k * d - d) / 2)
    sys.modules["indextts.BigVGAN.utils"] = ut
    d = REF + "/IndexTTS/modeling_modified/"
    _pkg("indextts.BigVGAN.alias_free_torch")
    _load("indextts.BigVGAN.alias_free_torch.filter", d + "filter.py")
    _load("indextts.BigVGAN.alias_free_torch.resample", d + "resample.py")
    aft = _load("indextts.BigVGAN.alias_free_torch.act", d + "act.py")
    sys.modules["indextts.BigVGAN.alias_free_torch"].Activation1d = aft.Activation1d
    return _load("indextts.BigVGAN.models", d + "models.py")
