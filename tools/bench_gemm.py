"""GEMM microbenchmark on the tower shapes (random data), interleaved rounds in one process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import _lib, ops

dev = torch.device("cuda:0")
M = int(os.environ.get("M", 65536))
shapes = [("qkv", 2304, 768, _lib.EPI_BIAS_BF16), ("oproj", 768, 768, _lib.EPI_BIAS_RESID_F32),
          ("fc1", 3072, 768, _lib.EPI_BIAS_GELU_BF16), ("fc1nogelu", 3072, 768, _lib.EPI_BIAS_BF16), ("fc2", 768, 3072, _lib.EPI_BIAS_RESID_F32),
          ("oprojbf", 768, 768, _lib.EPI_BIAS_BF16), ("fc2bf", 768, 3072, _lib.EPI_BIAS_BF16),   # main-loop studies (debug modes)
          ("patch", 768, 3072, _lib.EPI_F32), ("sq4096", 4096, 4096, _lib.EPI_BIAS_BF16)]
torch.manual_seed(0)
only = os.environ.get("ONLY")
for name, N, K, epi in shapes:
    if only and name not in only.split(","):
        continue
    m = 4096 if name == "sq4096" else M
    pad = int(os.environ.get("PAD", 0))   # extra elements in the leading dimensions of A and W
    A = torch.randn(m, K + pad, device=dev).to(torch.bfloat16)[:, :K]
    W = (torch.randn(N, K + pad, device=dev) * K ** -0.5).to(torch.bfloat16)[:, :K]
    if os.environ.get("ZERO"):   # all-zero operands: same cycles, much less switching power -> shows clock throttling
        A.zero_(); W.zero_()
    bias = torch.randn(N, device=dev)
    out = torch.zeros(m, N, device=dev, dtype=torch.float32 if epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_F32) else torch.bfloat16)
    for _ in range(3):
        ops.gemm_bf16(A, W, out, epi, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm_bf16(A, W, out, epi, bias=bias)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:8s} M={m} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*m*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)

if not only or "qkvattn" in only.split(","):
    n_seq, T, heads = M // 50, 50, 12
    d = heads * 64
    x = torch.randn(n_seq * T, d, device=dev).to(torch.bfloat16)
    w = (torch.randn(3 * d, d, device=dev) * d ** -0.5).to(torch.bfloat16)
    bias = torch.randn(3 * d, device=dev)
    for _ in range(3):
        ops.qkv_attention(x, w, bias, n_seq, T, heads)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.qkv_attention(x, w, bias, n_seq, T, heads)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"qkvattn  rows={n_seq*T} d={d}: {ms*1e3:8.1f} us  {2.0*n_seq*T*3*d*d/ms/1e9:7.1f} TFLOP/s (GEMM flops only)", flush=True)
