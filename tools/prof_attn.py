"""Runs sgclip_attention on the ViT-L/14@336 shape (T = 577, 16 heads) a few times: a target for rocprofv3 counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import ops
dev = torch.device("cuda:0")
n_seq, T, heads = int(os.environ.get("NSEQ", 512)), 577, 16
qkv = torch.randn(n_seq * T, 3 * heads * 64, device=dev).to(torch.bfloat16)
for _ in range(3):
    ops.attention(qkv, n_seq, T, heads, False, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.attention(qkv, n_seq, T, heads, False, None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"attention n_seq={n_seq} T={T} heads={heads}: {ms:.3f} ms  {4.0*n_seq*heads*T*T*64/ms/1e9:.1f} TFLOP/s")
