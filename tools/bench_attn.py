"""Attention microbenchmark: sgclip_attention at the tower geometries (random data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, n_seq, T, heads, causal in (("l14-336", 454, 577, 16, False), ("b32", 5242, 50, 12, False), ("text", 285, 20, 8, True),
                                      ("text-l14", 285, 20, 12, True)):
    d = heads * 64
    qkv = torch.randn(n_seq * T, 3 * d, device=dev).to(torch.bfloat16)
    for _ in range(2):
        ops.attention(qkv, n_seq, T, heads, causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        ops.attention(qkv, n_seq, T, heads, causal)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:9s} n_seq={n_seq} T={T} heads={heads}: {ms*1e3:8.1f} us  {4.0*n_seq*heads*T*T*64/ms/1e9:7.1f} TFLOP/s", flush=True)
