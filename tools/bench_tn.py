"""Times sgclip_gemm_tn_bf16 (weight gradients) on the shapes of a ViT-B/32 layer: rows x (out, in)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_amd import grad_ops as G

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
for out_f, in_f in ((3072, 768), (768, 3072), (2304, 768), (768, 768), (4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)):
    dys = [torch.randn(rows, out_f, device="cuda").bfloat16() for _ in range(3)]
    xs = [torch.randn(rows, in_f, device="cuda").bfloat16() for _ in range(3)]
    dw = torch.zeros(out_f, in_f, device="cuda")
    for i in range(3):
        G.linear_wgrad(dys[i], xs[i], dw)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 12
    a.record()
    for i in range(n):
        G.linear_wgrad(dys[i % 3], xs[i % 3], dw)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    print(f"rows {rows} dW [{out_f} x {in_f}]: {us:8.1f} us  {2.0 * rows * out_f * in_f / us / 1e6:7.1f} TFLOP/s")
