"""One weight-gradient GEMM shape and one NT GEMM of the same size, a few launches each, for counter collection."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_amd import _lib, grad_ops as G, ops

rows, out_f, in_f = 102400, 3072, 768
dy = torch.randn(rows, out_f, device="cuda").bfloat16()
x = torch.randn(rows, in_f, device="cuda").bfloat16()
dw = torch.zeros(out_f, in_f, device="cuda")
w = torch.randn(out_f, in_f, device="cuda").bfloat16()
y = torch.empty(rows, out_f, device="cuda", dtype=torch.bfloat16)
b = torch.zeros(out_f, device="cuda")
for _ in range(3):
    G.linear_wgrad(dy, x, dw)
    ops.gemm_bf16(x, w, y, _lib.EPI_BIAS_BF16, bias=b)
torch.cuda.synchronize()
