import torch
dev = torch.device("cuda", 0)
for name, M, N, K in [("dit_qkv_u8", 18016, 3072, 1024), ("dit_ff2_u8", 18016, 1024, 2048), ("dit_o_u8", 18016, 1024, 1024), ("8192^3", 8192, 8192, 8192), ("dit_o_u1", 2252, 1024, 1024), ("bv_s0", 16384, 768, 8448)]:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16); b = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.1
    for _ in range(3): c = a @ b.t()
    torch.cuda.synchronize()
a = torch.randn(2252, 1024, device=dev); b = torch.randn(1024, 1024, device=dev)
for _ in range(3): c = a @ b.t()
a = torch.randn(18016, 1024, device=dev); b = torch.randn(3072, 1024, device=dev)
for _ in range(3): c = a @ b.t()
torch.cuda.synchronize()
