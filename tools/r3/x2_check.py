#!/usr/bin/env python
"""fp32 linear layer through the panel-plane kernel with 3 bf16 planes / 2 fp16 planes / the native fp32 MFMA: error against
float64 and time per launch (run on the GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts import _lib, weights as W
from mi355tts import bigvgan as BV
_lib.init(0)
rms = lambda a: float(np.sqrt(np.mean(np.square(a.astype(np.float64)))))
for Ci, Co, T, B, scale in [(1024, 3072, 1126, 2, 1.0), (2048, 1024, 1126, 2, 1.0), (1024, 1024, 1126, 2, 300.0), (1024, 1024, 1126, 2, 1e-4)]:
    x = (W.synth_normal(1, f"x{Ci}{T}", (B, Ci, T)) * scale).astype(np.float32)
    w = W.synth_normal(2, f"w{Ci}{Co}", (Co, Ci, 1), std=1.0 / np.sqrt(Ci))
    b = W.synth_normal(3, "b", (Co,), std=0.1)
    ref64 = np.einsum("oc,bct->bot", w[:, :, 0].astype(np.float64), x.astype(np.float64)) + b.astype(np.float64)[None, :, None]
    out = {}
    for name, opts in [("planes3", {"gemm_f32_x3": 1, "gemm_f32_x3p": 1, "gemm_f32_planes": 3}), ("planes2", {"gemm_f32_x3": 1, "gemm_f32_x3p": 1, "gemm_f32_planes": 2}),
                       ("native", {"gemm_f32_x3": 0, "gemm_f32_x3p": 0})]:
        for k, v in opts.items():
            _lib.set_option(k, v)
        y = BV.conv1d(x, w, b, dtype="f32")
        y2 = BV.conv1d(x, w, b, dtype="f32")
        out[name] = y
        print(f"Ci{Ci} Co{Co} M{B*T} scale{scale:g} {name}: rms err vs f64 {rms(y - ref64):.3e} (rel {rms(y - ref64) / rms(ref64):.3e}) max {np.abs(y - ref64).max():.3e} repeat-identical {np.array_equal(y, y2)}", flush=True)
    print("   planes2 - planes3 max", np.abs(out["planes2"] - out["planes3"]).max())
for pl in (3, 2):
    _lib.set_option("gemm_f32_x3", 1); _lib.set_option("gemm_f32_x3p", 1); _lib.set_option("gemm_f32_planes", pl)
    for name, B, T, Cin, N in [("qkv", 2, 1126, 1024, 3072), ("o", 2, 1126, 1024, 1024), ("ff1", 2, 1126, 1024, 2048), ("ff2", 2, 1126, 2048, 1024)]:
        ms = _lib.bench_conv_gemm("f32", B, T, Cin, N, 1, 1, iters=400)
        fl = 2.0 * B * T * N * Cin
        print(f"planes{pl} {name}: {ms*1e3:7.1f} us {fl/ms/1e9:7.1f} TFLOP/s (fp32-equivalent)", flush=True)
