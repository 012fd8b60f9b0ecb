"""Checkpoint readers: the on-disk formats on the input side of the path -> the engine's packed fp32 blobs.

What the reference's export scripts load (all un-vendored upstream loaders; the key layouts are what they consume):

  F5-TTS   ``model_1250000.safetensors`` through f5_tts ``load_checkpoint(..., use_ema=True)``
           (F5_TTS/Export_F5.py:207-221): keys ``ema_model.transformer.*`` (+ ``ema_model.initted`` / ``ema_model.step`` and
           ``ema_model.mel_spec.*`` buffers, which the loader drops); a ``.pt`` holds the same under ``ema_model_state_dict``.
  Vocos    ``pytorch_model.bin`` through ``Vocos.from_pretrained`` (F5_TTS/modeling_modified/vocos/pretrained.py:62-79):
           a plain state dict with ``backbone.*``, ``head.*`` and ``feature_extractor.*`` (mel buffers, unused here).
  BigVGAN  ``bigvgan_generator.pt`` through ``BigVGAN._from_pretrained`` (BigVGAN/modeling_modified/bigvgan.py:505-514):
           ``{"generator": state_dict}``, convolutions either plain or weight-normed (``weight_g`` / ``weight_v`` pairs, or
           torch >= 2.1 ``parametrizations.weight.original0/1``) — folded like ``remove_weight_norm`` (:412-424).

The result is the upstream-named state dict that ``weights.pack_f5`` / ``weights.pack_bigvgan`` take (they apply the
export-time folds).  safetensors is parsed here (8-byte little-endian header length, JSON header, raw little-endian data);
``.pt`` / ``.bin`` go through ``torch.load(weights_only=True)``.
"""
from __future__ import annotations

import json
import struct
from typing import Dict, Optional

import numpy as np

from .config import BigVGANConfig, F5Config
from . import weights as W

_ST_DTYPES = {"F32": np.float32, "F16": np.float16, "F64": np.float64, "I64": np.int64, "I32": np.int32, "I16": np.int16,
              "I8": np.int8, "U8": np.uint8, "BOOL": np.bool_}


def read_safetensors(path: str) -> Dict[str, np.ndarray]:
    """All tensors of a .safetensors file as numpy arrays (BF16 widened to float32).  The header is validated before
    anything is allocated from it: a truncated or malformed file raises a ValueError that names the tensor."""
    import os
    out = {}
    fsize = os.path.getsize(path)
    with open(path, "rb") as f:
        h8 = f.read(8)
        if len(h8) != 8:
            raise ValueError(f"{path}: not a safetensors file (shorter than its 8-byte header length)")
        (hlen,) = struct.unpack("<Q", h8)
        if hlen > fsize - 8:
            raise ValueError(f"{path}: header length {hlen} exceeds the file ({fsize} bytes)")
        hraw = f.read(hlen)
        if len(hraw) != hlen:
            raise ValueError(f"{path}: short read of the header")
        try:
            header = json.loads(hraw.decode("utf-8"))
        except (UnicodeDecodeError, json.JSONDecodeError) as e:
            raise ValueError(f"{path}: header is not JSON ({e})") from None
        if not isinstance(header, dict):
            raise ValueError(f"{path}: header is not a JSON object")
        base = 8 + hlen
        dsize = fsize - base
        plan = []
        for name, meta in header.items():          # validate every entry first: no partial work on a bad file
            if name == "__metadata__":
                continue
            try:
                dt, shape = meta["dtype"], tuple(int(d) for d in meta["shape"])
                lo, hi = (int(v) for v in meta["data_offsets"])
            except (KeyError, TypeError, ValueError):
                raise ValueError(f"{path}: tensor {name}: malformed header entry {meta!r}") from None
            if dt == "BF16":
                isz = 2
            elif dt in _ST_DTYPES:
                isz = np.dtype(_ST_DTYPES[dt]).itemsize
            else:
                raise ValueError(f"{path}: tensor {name} has unsupported dtype {dt}")
            if any(d < 0 for d in shape):
                raise ValueError(f"{path}: tensor {name}: negative dimension in shape {shape}")
            if not (0 <= lo <= hi <= dsize):
                raise ValueError(f"{path}: tensor {name}: data_offsets [{lo}, {hi}) outside the {dsize}-byte data section")
            want = isz                      # python integers: a hostile shape cannot wrap the product
            for d in shape:
                want *= d
                if want > (1 << 56):        # absurd long before it could overflow anything downstream
                    raise ValueError(f"{path}: tensor {name}: shape {shape} x {dt} is larger than any file")
            if hi - lo != want:
                raise ValueError(f"{path}: tensor {name}: {hi - lo} bytes stored, shape {shape} x {dt} needs {want}")
            plan.append((name, dt, shape, lo, hi))
        for name, dt, shape, lo, hi in plan:
            f.seek(base + lo)
            raw = f.read(hi - lo)
            if len(raw) != hi - lo:
                raise ValueError(f"{path}: tensor {name}: short read ({len(raw)} of {hi - lo} bytes)")
            if dt == "BF16":
                a = (np.frombuffer(raw, dtype="<u2").astype(np.uint32) << 16).view(np.float32)
            else:
                a = np.frombuffer(raw, dtype=np.dtype(_ST_DTYPES[dt]).newbyteorder("<"))
            out[name] = np.array(a, copy=True).reshape(shape)
    return out


def write_safetensors(path: str, tensors: Dict[str, np.ndarray], metadata: Optional[Dict[str, str]] = None) -> None:
    """Minimal writer (tests, and re-exporting a folded state): float32 / float16 / integer numpy arrays."""
    inv = {np.dtype(v): k for k, v in _ST_DTYPES.items()}
    header, blobs, off = {}, [], 0
    if metadata:
        header["__metadata__"] = metadata
    for name, a in tensors.items():
        a = np.ascontiguousarray(a)
        if a.dtype not in inv:
            raise ValueError(f"{name}: dtype {a.dtype} not supported")
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        header[name] = {"dtype": inv[a.dtype], "shape": list(a.shape), "data_offsets": [off, off + len(raw)]}
        blobs.append(raw)
        off += len(raw)
    h = json.dumps(header, separators=(",", ":")).encode("utf-8")
    h += b" " * ((8 - len(h) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)))
        f.write(h)
        for raw in blobs:
            f.write(raw)


def read_torch(path: str) -> dict:
    """torch.load of a .pt / .pth / .bin checkpoint (tensors only) with every tensor converted to numpy."""
    import torch
    obj = torch.load(path, map_location="cpu", weights_only=True)

    def conv(o):
        if isinstance(o, torch.Tensor):
            return o.detach().to(torch.float32).numpy() if o.is_floating_point() else o.detach().numpy()
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        return o
    return conv(obj)


def _read_any(path: str) -> dict:
    return read_safetensors(path) if str(path).endswith(".safetensors") else read_torch(path)


# ---------------------------------------------------------------------------------------------------------------------
# F5-TTS + Vocos
# ---------------------------------------------------------------------------------------------------------------------
def f5_transformer_state(ckpt: dict, use_ema: bool = True) -> Dict[str, np.ndarray]:
    """What f5_tts ``load_checkpoint(model, path, device, use_ema=True)`` hands to ``model.load_state_dict``: the
    ``ema_model.``-prefixed tensors (a .pt nests them under ``ema_model_state_dict``), without the EMA bookkeeping
    (``initted``, ``step``) and the mel-spectrogram buffers; keys come out as ``transformer.*``."""
    if use_ema:
        sd = ckpt.get("ema_model_state_dict", ckpt)
        sd = {k[len("ema_model."):]: v for k, v in sd.items() if k.startswith("ema_model.") and k not in ("ema_model.initted", "ema_model.step")}
    else:
        sd = ckpt.get("model_state_dict", ckpt)
    out = {k: np.asarray(v, dtype=np.float32) for k, v in sd.items() if k.startswith("transformer.")}
    if not out:
        raise ValueError("no transformer.* tensors found (is this an F5-TTS checkpoint, and is use_ema right?)")
    return out


def vocos_state(ckpt: dict) -> Dict[str, np.ndarray]:
    """``pytorch_model.bin`` of Vocos: backbone.* / head.* (feature_extractor.* are mel buffers the path does not use)."""
    out = {"vocos." + k: np.asarray(v, dtype=np.float32) for k, v in ckpt.items() if k.startswith(("backbone.", "head."))}
    if not out:
        raise ValueError("no backbone.* / head.* tensors found (is this a Vocos checkpoint?)")
    return out


def load_f5_state(cfg: F5Config, f5_ckpt: str, vocos_ckpt: str, use_ema: bool = True) -> Dict[str, np.ndarray]:
    """Both checkpoints -> the state dict of ``weights.f5_spec(cfg)`` (missing / mis-shaped tensors raise)."""
    st = {}
    st.update(f5_transformer_state(_read_any(f5_ckpt), use_ema))
    st.update(vocos_state(_read_any(vocos_ckpt)))
    return _check(W.f5_spec(cfg), st, f"{f5_ckpt} + {vocos_ckpt}")


def pack_f5_from_files(cfg: F5Config, f5_ckpt: str, vocos_ckpt: str, use_ema: bool = True) -> np.ndarray:
    return W.pack_f5(cfg, load_f5_state(cfg, f5_ckpt, vocos_ckpt, use_ema))


# ---------------------------------------------------------------------------------------------------------------------
# BigVGAN
# ---------------------------------------------------------------------------------------------------------------------
def fold_weight_norm(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """weight_g / weight_v (torch.nn.utils.weight_norm) and parametrizations.weight.original0 / original1
    (torch.nn.utils.parametrizations.weight_norm) -> plain ``weight``, as ``remove_weight_norm`` leaves them."""
    st = {}
    for k, v in state.items():
        if k.endswith(".parametrizations.weight.original0"):
            st[k[: -len(".parametrizations.weight.original0")] + ".weight_g"] = v
        elif k.endswith(".parametrizations.weight.original1"):
            st[k[: -len(".parametrizations.weight.original1")] + ".weight_v"] = v
        else:
            st[k] = v
    return W.remove_weight_norm_state(st)


def load_bigvgan_state(cfg: BigVGANConfig, ckpt: str) -> Dict[str, np.ndarray]:
    """``bigvgan_generator.pt`` (``{"generator": ...}``; a bare state dict is accepted too) -> state of ``weights.bigvgan_spec``."""
    obj = _read_any(ckpt)
    sd = obj["generator"] if isinstance(obj.get("generator", None), dict) else obj
    sd = fold_weight_norm({k: np.asarray(v) for k, v in sd.items()})
    sd = {k: v.astype(np.float32) for k, v in sd.items() if ".filter" not in k}       # registered FIR buffers are rebuilt
    return _check(W.bigvgan_spec(cfg), sd, ckpt)


def pack_bigvgan_from_file(cfg: BigVGANConfig, ckpt: str) -> np.ndarray:
    return W.pack_bigvgan(cfg, load_bigvgan_state(cfg, ckpt))


def _check(spec, st: Dict[str, np.ndarray], what: str) -> Dict[str, np.ndarray]:
    missing = [n for n, _, _ in spec if n not in st]
    if missing:
        raise KeyError(f"{what}: {len(missing)} tensors missing, e.g. {missing[:4]}")
    bad = [(n, st[n].shape, tuple(sh)) for n, sh, _ in spec if tuple(st[n].shape) != tuple(sh)]
    if bad:
        raise ValueError(f"{what}: shape mismatch (does the config match the checkpoint?), e.g. {bad[:3]}")
    return {n: st[n] for n, _, _ in spec}
