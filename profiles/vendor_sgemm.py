import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda'
for (M, N, K) in [(8192, 2048, 1024), (8192, 2048, 1536), (8192, 2048, 768), (16384, 2048, 1024), (8192, 8192, 8192), (51712, 2048, 512)]:
    a = torch.rand(M, K, device=dev) * 2 - 1
    b = torch.rand(N, K, device=dev) * 2 - 1
    for _ in range(5): c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): c = a @ b.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print('torch fp32 (rocBLAS/hipBLASLt) M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s' % (M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
