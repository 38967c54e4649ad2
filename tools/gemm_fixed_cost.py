"""Per-tile fixed cost of the 256x256 GEMM: main-loop-only time (SGCLIP_GEMM_DEBUG=1) against the reduction depth."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import _lib, ops
dev = torch.device("cuda:0")
M, N = 262144, 2304
for K in (256, 512, 768, 1536, 3072):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_BF16, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_bf16(A, W, out, _lib.EPI_BIAS_BF16, bias=bias)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles_per_cu = (M // 256) * (N // 256) / 256
    print(f"K={K}: {ms*1e3:8.1f} us  per tile {ms*1e3/tiles_per_cu:6.2f} us  stages {K//64}", flush=True)
