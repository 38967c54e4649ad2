"""Times sgclip_cast_colsum_f32_bf16 against the separate cast and column-sum kernels (M x N f32)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from laser_amd import grad_ops as G

M, N = int(sys.argv[1]) if len(sys.argv) > 1 else 102400, int(sys.argv[2]) if len(sys.argv) > 2 else 768
xs = [torch.randn(M, N, device="cuda") for _ in range(4)]   # rotate buffers: more than the 256 MiB cache
acc = torch.zeros(N, device="cuda")

def timed(fn, n=20):
    for i in range(3):
        fn(xs[i % 4])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(xs[i % 4])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("fused  us", timed(lambda x: G.cast_bf16_colsum(x, acc)))
print("cast   us", timed(lambda x: G.cast_bf16(x)))
print("colsum us", timed(lambda x: G.colsum(x, acc)))
print("bytes/fused GB", M * N * 6 / 1e9)
