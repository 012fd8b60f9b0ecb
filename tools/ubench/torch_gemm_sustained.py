import torch, time
dev = torch.device("cuda", 0)
for name, M, N, K in [("dit_qkv_u8", 18016, 3072, 1024), ("dit_ff2_u8", 18016, 1024, 2048), ("8192^3", 8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.1
    for iters in (200, 15000 if M > 10000 else 3000):
        for _ in range(10): c = a @ b.t()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): c = a @ b.t()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        print(f"{name:12s} iters {iters:6d}: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:8.1f} TFLOP/s", flush=True)
        time.sleep(2)
