"""Times sgclip_gemm_tn_bf16 (C += A^T B, the weight-gradient product) on the tower shapes of a train step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from laser_amd import _lib
from laser_amd.ops import _stream

dev = torch.device("cuda:0")
Kd = int(os.environ.get("KD", 44800))
lib = _lib.load()
for name, M, N in (("qkv", 2304, 768), ("oproj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(Kd, M, device=dev).to(torch.bfloat16)
    b = torch.randn(Kd, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    f = lambda: _lib.check(lib.sgclip_gemm_tn_bf16(a.data_ptr(), M, b.data_ptr(), N, c.data_ptr(), N, M, N, Kd, _stream()))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:6s} Kd={Kd} M={M} N={N}: {ms*1e3:8.1f} us  {2.0*Kd*M*N/ms/1e9:7.1f} TFLOP/s", flush=True)
