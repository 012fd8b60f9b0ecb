#!/usr/bin/env python
"""Micro-benchmark of the implicit-GEMM conv kernel on the hot shapes (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "text-to-speech-tts-onnx_amd"))
from mi355tts import _lib
_lib.init(0)
shapes = [  # name, dtype, B, T, Cin, N, taps, dil
    ("dit_qkv", "bf16", 2, 1126, 1024, 3072, 1, 1), ("dit_o", "bf16", 2, 1126, 1024, 1024, 1, 1),
    ("dit_ff1", "bf16", 2, 1126, 1024, 2048, 1, 1), ("dit_ff2", "bf16", 2, 1126, 2048, 1024, 1, 1),
    ("dit_qkv_u8", "bf16", 16, 1126, 1024, 3072, 1, 1),
    ("gemm4k", "bf16", 1, 4096, 4096, 4096, 1, 1),
    ("bv_s0_k11", "f16", 8, 2048, 768, 768, 11, 5), ("bv_s0_k3", "f16", 8, 2048, 768, 768, 3, 1),
    ("bv_s1_k7", "f16", 8, 8192, 384, 384, 7, 3), ("bv_s2_k7", "f16", 8, 16384, 192, 192, 7, 1),
    ("dit_qkv_f32", "f32", 2, 1126, 1024, 3072, 1, 1),
]
sel = sys.argv[1:]
if sel and sel[0] == "custom":           # custom dtype B T Cin N taps dil [more 7-tuples...]
    a = sel[1:]
    shapes = [("custom", a[i], *map(int, a[i + 1:i + 7])) for i in range(0, len(a), 7)]
    sel = []
for name, dt, B, T, Cin, N, taps, dil in shapes:
    if sel and name not in sel:
        continue
    ms = _lib.bench_conv_gemm(dt, B, T, Cin, N, taps, dil, iters=int(os.environ.get("ITERS", "20")))
    fl = 2.0 * B * T * N * Cin * taps
    print(f"{name:12s} {dt:5s} B{B} T{T} Cin{Cin} N{N} k{taps} d{dil}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TFLOP/s", flush=True)
