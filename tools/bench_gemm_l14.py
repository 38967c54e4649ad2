import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import _lib, ops
dev = torch.device("cuda:0")
M = 262144
shapes = [("qkv", 3072, 1024, _lib.EPI_BIAS_BF16), ("oproj", 1024, 1024, _lib.EPI_BIAS_RESID_F32),
          ("fc1", 4096, 1024, _lib.EPI_BIAS_GELU_BF16), ("fc2", 1024, 4096, _lib.EPI_BIAS_RESID_F32), ("patch", 1024, 640, _lib.EPI_F32)]
torch.manual_seed(0)
for name, N, K, epi in shapes:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_F32) else torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16(A, W, out, epi, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_bf16(A, W, out, epi, bias=bias)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name:8s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
    del A, W, out
