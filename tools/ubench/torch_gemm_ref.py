#!/usr/bin/env python
"""Library GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the hot-path shapes, warm, random data: the yardstick the
hand-written kernels are compared with in DESIGN.md (tuning aid; nothing in the product path uses it)."""
import torch, time
dev = torch.device("cuda", 0)
shapes = [("8192^3", 8192, 8192, 8192), ("4096^3", 4096, 4096, 4096), ("dit_qkv_u8", 18016, 3072, 1024), ("dit_o_u8", 18016, 1024, 1024),
          ("dit_ff1_u8", 18016, 2048, 1024), ("dit_ff2_u8", 18016, 1024, 2048), ("dit_qkv_u1", 2252, 3072, 1024), ("dit_o_u1", 2252, 1024, 1024),
          ("dit_ff2_u1", 2252, 1024, 2048), ("bv_s0_k11 (as plain GEMM)", 16384, 768, 8448), ("bv_s1_k7", 65536, 384, 2688), ("bv_s2_k7", 131072, 192, 1344)]
for dt in (torch.bfloat16, torch.float32):
    for name, M, N, K in shapes:
        if dt == torch.float32 and M * N * K > 8192 * 8192 * 2048:
            continue
        a = torch.randn(M, K, device=dev, dtype=dt)
        b = (torch.randn(N, K, device=dev, dtype=dt) * 0.1)
        for _ in range(10):
            c = a @ b.t()
        torch.cuda.synchronize()
        iters = 200 if dt == torch.bfloat16 else 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{str(dt):16s} {name:28s} M{M} N{N} K{K}: {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
