#!/usr/bin/env python
"""gemm_x1d.hip (exact-fit 16-bit linear layers) against float64 numpy and against the eight-phase kernel it replaces, through
mi_conv1d (k = 1), plus bare timings of the four eight-utterance DiT shapes with it on and off (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts import _lib, bigvgan
_lib.init(0)
rng = np.random.default_rng(11)
bad = 0
for dt, tol in (("bf16", 1.2e-2), ("f16", 1.5e-3)):
    for T, Cin, Cout in [(18016, 1024, 1024), (18016, 1024, 3072), (18016, 2048, 1024), (18000, 1024, 2048)]:
        x = rng.standard_normal((1, Cin, T)).astype(np.float32)
        w = (rng.standard_normal((Cout, Cin, 1)) / np.sqrt(Cin)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        ref = np.einsum("oc,ct->ot", w[:, :, 0].astype(np.float64), x[0].astype(np.float64)) + b[:, None].astype(np.float64)
        outs = {}
        for on in (1, 0):
            _lib.set_option("gemm_x1d", on)
            outs[on] = bigvgan.conv1d(x, w, b, dtype=dt)[0].astype(np.float64)
            err = np.abs(outs[on] - ref).max() / np.abs(ref).max()
            flag = "" if err < tol else "   <-- BAD"
            bad += err >= tol
            print(f"{dt} T{T} K{Cin} N{Cout} x1d={on}: max err / max |ref| {err:.2e}{flag}", flush=True)
        d = np.abs(outs[1] - outs[0]).max() / np.abs(ref).max()
        print(f"     x1d vs eight-phase kernel: {d:.2e}", flush=True)
        bad += d >= tol
it = int(os.environ.get("ITERS", "30"))
for (K, N) in [(1024, 3072), (1024, 1024), (1024, 2048), (2048, 1024)]:
    r = []
    for on in (0, 1):
        _lib.set_option("gemm_x1d", on)
        r.append(_lib.bench_conv_gemm("bf16", 16, 1126, K, N, 1, 1, iters=it) * 1e3)
    fl = 2.0 * 18016 * N * K
    print(f"bf16 M18016 K{K} N{N}: eight-phase (+ row split) {r[0]:6.1f} us {fl / r[0] / 1e6:6.0f} TF   exact-fit {r[1]:6.1f} us {fl / r[1] / 1e6:6.0f} TF", flush=True)
print("FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
