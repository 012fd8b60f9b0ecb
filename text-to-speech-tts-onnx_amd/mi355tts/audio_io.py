"""Audio file I/O either side of the hot path (host only, numpy).

The reference drivers read the prompt with pydub and write the result with soundfile:

    audio = np.array(AudioSegment.from_file(reference_audio).set_channels(1).set_frame_rate(SAMPLE_RATE)
                     .get_array_of_samples(), dtype=np.int16)            F5_TTS/F5-TTS-ONNX-Inference.py:223
    sf.write(generated_audio, audio_out.reshape(-1), SAMPLE_RATE, format='WAVEX')                   :315

Neither package is installed here and both are thin: for RIFF/WAVE input pydub parses the file itself, mixes to mono
with ``audioop.tomono(data, width, 0.5, 0.5)`` and resamples with ``audioop.ratecv`` (a linear-interpolation DDA);
soundfile's 'WAVEX' is a WAVE_FORMAT_EXTENSIBLE header.  Both are restated below in numpy.  tests/test_audio_io.py
pins ``ratecv`` / ``tomono`` against CPython's own ``audioop`` module (bit-exact) and round-trips the writer.

Deviation: a prompt that is not 16-bit PCM is converted to 16 bits first (pydub would hand 8/24/32-bit samples to
``np.array(..., dtype=np.int16)`` unconverted, which wraps).
"""
from __future__ import annotations

import math
import struct
from typing import Tuple

import numpy as np

_PCM_GUID = bytes.fromhex("0100000000001000800000aa00389b71")      # KSDATAFORMAT_SUBTYPE_PCM


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """RIFF/WAVE PCM (8/16/24/32-bit, WAVE_FORMAT_PCM or EXTENSIBLE/PCM) -> (int16 array (frames, channels), rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, rate, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 40:
                if body[24:40] != _PCM_GUID:
                    raise ValueError(f"{path}: WAVE_FORMAT_EXTENSIBLE sub-format is not PCM")
                tag = 1
            if tag != 1:
                raise ValueError(f"{path}: only integer PCM is supported (format tag {tag})")
            fmt = (ch, rate, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    ch, rate, bits = fmt
    width = bits // 8
    n = len(pcm) // (width * ch)
    raw = np.frombuffer(pcm[: n * width * ch], dtype=np.uint8).reshape(n * ch, width)
    if width == 1:
        s = (raw[:, 0].astype(np.int16) - 128) << 8
    elif width == 2:
        s = raw.copy().view("<i2")[:, 0]
    elif width == 3:
        v = raw[:, 0].astype(np.int32) | (raw[:, 1].astype(np.int32) << 8) | (raw[:, 2].astype(np.int8).astype(np.int32) << 16)
        s = (v >> 8).astype(np.int16)
    elif width == 4:
        s = (raw.copy().view("<i4")[:, 0] >> 16).astype(np.int16)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    return s.astype(np.int16).reshape(n, ch), int(rate)


def tomono(x: np.ndarray) -> np.ndarray:
    """pydub ``set_channels(1)``: stereo -> ``audioop.tomono(data, 2, 0.5, 0.5)`` = floor(0.5 l + 0.5 r); more channels
    -> floor(mean) (pydub averages the split channels the same way)."""
    x = np.asarray(x, dtype=np.int16)
    if x.ndim == 1 or x.shape[1] == 1:
        return x.reshape(-1)
    return np.floor(x.astype(np.float64).sum(axis=1) / x.shape[1]).astype(np.int16)


def ratecv(x: np.ndarray, inrate: int, outrate: int) -> np.ndarray:
    """``audioop.ratecv(data, 2, 1, inrate, outrate, None)`` (CPython Modules/audioop.c): a DDA that emits output k
    once input j = ceil(k*inrate/outrate) has been read, out = trunc((x[j-1]*d + x[j]*(outrate-d)) / outrate) with
    d = j*outrate - k*inrate, on samples scaled to 32 bits; x[-1] = 0."""
    x = np.asarray(x, dtype=np.int16).reshape(-1)
    if inrate <= 0 or outrate <= 0:
        raise ValueError("sampling rate not > 0")
    g = math.gcd(int(inrate), int(outrate))
    inr, outr = int(inrate) // g, int(outrate) // g
    n = x.size
    if n == 0:
        return x.copy()
    kmax = ((n - 1) * outr) // inr                      # largest k with ceil(k*inr/outr) <= n-1
    k = np.arange(kmax + 1, dtype=np.int64)
    j = -((-k * inr) // outr)                           # ceil
    d = (j * outr - k * inr).astype(np.float64)
    x32 = x.astype(np.int64) * 65536
    cur = x32[j].astype(np.float64)
    prev = np.where(j > 0, x32[np.maximum(j - 1, 0)], 0).astype(np.float64)
    o = np.trunc((prev * d + cur * (outr - d)) / float(outr)).astype(np.int64)
    return (o >> 16).astype(np.int16)


def load_prompt(path: str, sample_rate: int = 24000) -> np.ndarray:
    """The reference's prompt loader for a WAV file: mono, `sample_rate`, int16, shape (1, 1, L)."""
    x, rate = read_wav(path)
    m = tomono(x)
    if rate != sample_rate:
        m = ratecv(m, rate, sample_rate)
    return np.ascontiguousarray(m, dtype=np.int16).reshape(1, 1, -1)


def write_wavex(path: str, audio: np.ndarray, sample_rate: int = 24000) -> None:
    """``soundfile.write(path, int16, rate, format='WAVEX')``: 16-bit PCM in a WAVE_FORMAT_EXTENSIBLE container
    (fmt chunk of 40 bytes, PCM sub-format GUID, 'fact' chunk with the frame count, as libsndfile lays it out)."""
    a = np.asarray(audio)
    if a.dtype != np.int16:
        raise ValueError("write_wavex expects int16 samples")
    if a.ndim == 2 and a.shape[1] in (1, 2) and a.shape[0] != 1:
        a = np.ascontiguousarray(a)                      # (frames, channels)
    else:
        a = np.ascontiguousarray(a.reshape(-1, 1))       # (L,), (1, L) or the engine's (1, 1, L): mono
    frames, ch = a.shape
    mask = {1: 0x4, 2: 0x3}.get(ch, 0)                   # front-centre / front-left+right
    payload = a.astype("<i2").tobytes()
    fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, sample_rate, sample_rate * ch * 2, ch * 2, 16, 22, 16, mask) + _PCM_GUID
    fact = struct.pack("<I", frames)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact \
        + b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)
