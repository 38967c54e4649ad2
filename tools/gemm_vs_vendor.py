"""The tower GEMM shapes on this library next to the vendor BLAS behind torch (hipBLASLt / rocBLAS), in ONE process,
interleaved rounds, random data.  The vendor numbers are a yardstick only: nothing in the product path calls them.

Per shape, two vendor figures: `matmul` = the bare product with a bf16 result (what the library's main loop is read
against) and `same result` = what producing this library's epilogue result costs with vendor + elementwise kernels
(F.linear with bias, then the quick-GELU or the fp32 residual add as separate launches).  Shapes follow
transformers models/clip/modeling_clip.py:298-350 as called from laser/models/llava_clip_model_v3.py:77-83.

usage: python tools/gemm_vs_vendor.py [rows]   (default 262144)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from laser_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
B32 = [("B/32 q/k/v (plain GEMM)", 2304, 768, _lib.EPI_BIAS_BF16), ("B/32 out-proj + residual", 768, 768, _lib.EPI_BIAS_RESID_F32),
       ("B/32 fc1 + quick-GELU", 3072, 768, _lib.EPI_BIAS_GELU_BF16), ("B/32 fc2 + residual", 768, 3072, _lib.EPI_BIAS_RESID_F32)]
L14 = [("L/14 q/k/v", 3072, 1024, _lib.EPI_BIAS_BF16), ("L/14 out-proj + residual", 1024, 1024, _lib.EPI_BIAS_RESID_F32),
       ("L/14 fc1 + quick-GELU", 4096, 1024, _lib.EPI_BIAS_GELU_BF16), ("L/14 fc2 + residual", 1024, 4096, _lib.EPI_BIAS_RESID_F32)]
torch.manual_seed(0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3   # us


print(f"# rows = {M}; us per launch (TFLOP/s); best of 3 interleaved rounds")
print("# shape | this library | vendor matmul (bf16 out) | vendor, same result | library / vendor-same-result")
for name, N, K, epi in B32 + L14:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    bias_bf = bias.to(torch.bfloat16)
    resid = epi == _lib.EPI_BIAS_RESID_F32
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if resid else torch.bfloat16)
    x = torch.randn(M, N, device=dev) if resid else None

    def ours():
        ops.gemm_bf16(A, W, out, epi, bias=bias)

    def vendor_plain():
        torch.matmul(A, W.t())

    def vendor_same():
        y = F.linear(A, W, bias_bf)
        if epi == _lib.EPI_BIAS_GELU_BF16:
            y = y * torch.sigmoid(1.702 * y)
        elif resid:
            x.add_(y)
        return y
    best = [1e30, 1e30, 1e30]
    for _ in range(3):
        for i, fn in enumerate((ours, vendor_plain, vendor_same)):
            best[i] = min(best[i], timed(fn))
    fl = 2.0 * M * N * K
    tf = [fl / (t * 1e-6) / 1e12 for t in best]
    print(f"{name:28s} | {best[0]:8.1f} ({tf[0]:6.0f}) | {best[1]:8.1f} ({tf[1]:6.0f}) | {best[2]:8.1f} ({tf[2]:6.0f}) | "
          f"{best[0] / best[2]:.2f}", flush=True)
    del A, W, out, x

# the fused q/k/v projection + attention kernel against vendor q/k/v GEMM + this library's attention kernel
n_seq, T, heads = M // 50, 50, 12
d = heads * 64
xq = torch.randn(n_seq * T, d, device=dev).to(torch.bfloat16)
wq = (torch.randn(3 * d, d, device=dev) * d ** -0.5).to(torch.bfloat16)
bq = torch.randn(3 * d, device=dev)
bq_bf = bq.to(torch.bfloat16)
best = [1e30, 1e30]
for _ in range(3):
    best[0] = min(best[0], timed(lambda: ops.qkv_attention(xq, wq, bq, n_seq, T, heads)))
    best[1] = min(best[1], timed(lambda: ops.attention(F.linear(xq, wq, bq_bf), n_seq, T, heads, False)))
fl = 2.0 * n_seq * T * 3 * d * d
print(f"{'B/32 q/k/v + attention, fused':28s} | {best[0]:8.1f} ({fl / best[0] / 1e6:6.0f}) | vendor q/k/v + separate attention kernel "
      f"{best[1]:8.1f} | {best[0] / best[1]:.2f}")
