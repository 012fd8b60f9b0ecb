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
    with open(path, "r", encoding="utf-8") as f:
        src = "".join(f.readlines()[start - 1:end])
    for a, b in replace:
        assert a in src, a
        src = src.replace(a, b)
    exec(compile(src, f"{path}:{start}-{end}", "exec"), ns)
    return ns
