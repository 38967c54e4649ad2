"""Calibration: the vendor BLAS behind torch (hipBLASLt / rocBLAS) on the tower GEMM shapes, random data.
Not part of the product path; prints TFLOP/s so the hand-written kernels can be read against it."""
import os, sys
import torch

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 262144))
shapes = [("qkv", M, 2304, 768), ("oproj", M, 768, 768), ("fc1", M, 3072, 768), ("fc2", M, 768, 3072),
          ("sq4096", 4096, 4096, 4096), ("sq8192", 8192, 8192, 8192)]
torch.manual_seed(0)
for name, m, N, K in shapes:
    A = torch.randn(m, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev).to(torch.bfloat16)
    for label, fn in (("matmul", lambda: torch.matmul(A, W.t())), ("linear+bias", lambda: torch.nn.functional.linear(A, W, bias))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name:8s} {label:12s} M={m} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*m*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
