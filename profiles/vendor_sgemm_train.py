"""Calibration: the vendor library (torch fp32 matmul / addmm on this image = hipBLASLt) on the train step's plain whole-sequence
GEMM shapes -- input projections, data gradients, weight gradients (accumulating, beta = 1).  Not a dependency."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = 'cuda'


def bench(M, N, K, ta=False, tb=True, acc=False, iters=20):
    A = torch.randn((K, M) if ta else (M, K), device=dev)
    B = torch.randn((N, K) if tb else (K, N), device=dev)
    a = A.t() if ta else A
    b = B.t() if tb else B
    C = torch.zeros((M, N), device=dev)
    f = (lambda: C.addmm_(a, b)) if acc else (lambda: torch.matmul(a, b, out=C))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print('M=%6d N=%5d K=%6d ta=%d tb=%d beta=%d: %8.1f us  %6.1f TFLOP/s' % (M, N, K, ta, tb, acc, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)


TB = 51712
for K in (512, 1024, 1536):
    bench(TB, 2048, K)                                  # forward input projection  X[TB][K] . Wx[4W][K]^T
for N in (512, 1024, 1536):
    bench(2048, N, TB, ta=True, tb=False, acc=True)     # weight gradient   dW += Z[TB][4W]^T . X[TB][N]
    bench(TB, N, 2048, tb=False)                        # data gradient  Z[TB][4W] . Wx[4W][N]
    bench(TB, N, 2048, tb=False, acc=True)
