#!/usr/bin/env python
"""gemm_x3d.hip against float64 numpy through mi_conv1d (k = 1 convolution = linear layer, plain row epilogue), and bare-GEMM
timings of the four one-utterance DiT shapes with the exact-fit kernel on and off (run on the GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts import _lib, bigvgan
_lib.init(0)
rng = np.random.default_rng(7)
bad = 0
for T, Cin, Cout in [] if os.environ.get("X3D_TIMING_ONLY") == "1" else [(2252, 1024, 1024), (2252, 1024, 2048), (2252, 1024, 3072), (2252, 2048, 1024), (2100, 1024, 1024),
                     (2304, 1024, 3072), (4504, 1024, 2048), (9008, 1024, 1024), (1126, 1024, 3072)]:
    x = rng.standard_normal((1, Cin, T)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1)) / np.sqrt(Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = np.einsum("oc,ct->ot", w[:, :, 0].astype(np.float64), x[0].astype(np.float64)) + b[:, None].astype(np.float64)
    for on in (1, 0):
        _lib.set_option("gemm_x3d", on)
        y = bigvgan.conv1d(x, w, b, dtype="f32")[0].astype(np.float64)
        err = np.abs(y - ref).max() / np.abs(ref).max()
        rms = np.sqrt(((y - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
        flag = "" if err < 2e-6 else "   <-- BAD"
        bad += err >= 2e-6
        print(f"T{T} K{Cin} N{Cout} x3d={on}: max rel {err:.2e} rms rel {rms:.2e}{flag}", flush=True)
it = int(os.environ.get("ITERS", "40"))
for dbg in (0, 4):
    os.environ["MI355TTS_GEMM_DBG"] = str(dbg)
    for (K, N) in [(1024, 3072), (1024, 1024), (1024, 2048), (2048, 1024)]:
        r = []
        for on in (0, 1):
            _lib.set_option("gemm_x3d", on)
            r.append(_lib.bench_conv_gemm("f32", 2, 1126, K, N, 1, 1, iters=it) * 1e3)
        print(f"dbg{dbg} K{K} N{N}: stream-K {r[0]:6.1f} us   exact-fit {r[1]:6.1f} us", flush=True)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
