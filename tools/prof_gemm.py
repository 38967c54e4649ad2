"""One launch of each tower GEMM shape (M = 65536) for PMC collection."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import _lib, ops
dev = torch.device("cuda:0")
M = int(os.environ.get("M", 65536))
torch.manual_seed(0)
for name, N, K, epi in [("qkv", 2304, 768, _lib.EPI_BIAS_BF16), ("oproj", 768, 768, _lib.EPI_BIAS_RESID_F32),
                        ("fc1", 3072, 768, _lib.EPI_BIAS_GELU_BF16), ("fc2", 768, 3072, _lib.EPI_BIAS_RESID_F32)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == _lib.EPI_BIAS_RESID_F32 else torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16(A, W, out, epi, bias=bias)
    torch.cuda.synchronize()
