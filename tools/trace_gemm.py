"""Per-workgroup time trace of gemm_pipe64_kernel (SGCLIP_GEMM_DEBUG=5): how long main loops and epilogues take, and how
many workgroups are in their epilogue at the same time.  Usage: SGCLIP_GEMM_DEBUG=5 python tools/trace_gemm.py N K epi [M]"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from laser_amd import _lib, ops

N, K, epi = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
M = int(sys.argv[4]) if len(sys.argv) > 4 else 262144
dev = torch.device("cuda:0")
torch.manual_seed(0)
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi in (_lib.EPI_BIAS_RESID_F32, _lib.EPI_F32) else torch.bfloat16)
for _ in range(40):
    ops.gemm_bf16(A, W, out, epi, bias=bias)
torch.cuda.synchronize()
tiles = ((M + 255) // 256) * ((N + 255) // 256)
n = min(tiles, 16384)
buf = np.zeros((n, 16), dtype=np.uint64)
lib = _lib.load()
fn = lib.sgclip_debug_gemm_trace
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
fn.restype = ctypes.c_int
assert fn(buf.ctypes.data, n) == n
t0 = buf[:, 0].astype(np.int64); t1 = buf[:, 1].astype(np.int64); t2 = buf[:, 2].astype(np.int64)
base = t0.min()
t0, t1, t2 = (t0 - base) * 0.01, (t1 - base) * 0.01, (t2 - base) * 0.01   # us
hw = buf[:, 3]
xcc = (hw >> np.uint64(32)) & np.uint64(0xF)
cu = ((hw & np.uint64(0xFFFFFFFF)) >> np.uint64(8)) & np.uint64(0xF)
se = ((hw & np.uint64(0xFFFFFFFF)) >> np.uint64(13)) & np.uint64(0x7)
sh = ((hw & np.uint64(0xFFFFFFFF)) >> np.uint64(12)) & np.uint64(0x1)
slot = (xcc.astype(np.int64) * 8 + se.astype(np.int64)) * 32 + sh.astype(np.int64) * 16 + cu.astype(np.int64)
print(f"tiles traced {n} of {tiles}; span {t2.max():.1f} us; distinct CU slots {len(np.unique(slot))}")
ml, ep = t1 - t0, t2 - t1
if buf[:, 9].any():
    st = (buf[:, 9:13].astype(np.int64) - base) * 0.01
    print("first stages of a tile, mean us: %s; remaining %d stages %.2f each" % (
        " ".join("%.2f" % v for v in np.diff(np.concatenate([t0[:, None], st], 1), axis=1).mean(0)), K // 64 - 4,
        (t1 - st[:, 3]).mean() / (K // 64 - 4)))
inner = (buf[:, 4:9].astype(np.int64) - base) * 0.01
print("epilogue of wave 0, mean us: barrier %.2f, passes %s, store acknowledgement %.2f" % (
    (inner[:, 0] - t1).mean(), " ".join("%.2f" % v for v in np.diff(inner, axis=1).mean(0)), (t2 - inner[:, 4]).mean()))
print(f"main loop  us: mean {ml.mean():.2f} p10 {np.percentile(ml,10):.2f} p50 {np.percentile(ml,50):.2f} p90 {np.percentile(ml,90):.2f}")
print(f"epilogue   us: mean {ep.mean():.2f} p10 {np.percentile(ep,10):.2f} p50 {np.percentile(ep,50):.2f} p90 {np.percentile(ep,90):.2f}")
# gap between a workgroup's end and the next start on the same CU
order = np.lexsort((t0, slot))
gaps = []
for a, b in zip(order[:-1], order[1:]):
    if slot[a] == slot[b]:
        gaps.append(t0[b] - t2[a])
gaps = np.array(gaps)
print(f"turnaround us (end -> next start on the CU): mean {gaps.mean():.2f} p50 {np.percentile(gaps,50):.2f} p90 {np.percentile(gaps,90):.2f}")
# concurrency of epilogues over time, sampled every 0.5 us
grid = np.arange(0, t2.max(), 0.5)
conc = np.array([np.count_nonzero((t1 <= g) & (t2 > g)) for g in grid])
busy = np.array([np.count_nonzero((t0 <= g) & (t2 > g)) for g in grid])
mid = (grid > 0.2 * t2.max()) & (grid < 0.8 * t2.max())
print(f"workgroups in epilogue at a time (middle 60 %): mean {conc[mid].mean():.1f} p10 {np.percentile(conc[mid],10):.0f} "
      f"p50 {np.percentile(conc[mid],50):.0f} p90 {np.percentile(conc[mid],90):.0f} max {conc[mid].max()}; resident mean {busy[mid].mean():.1f}")
# main loop duration against the epilogue concurrency it saw
cm = np.array([conc[(grid >= a) & (grid < b)].mean() if b - a > 1 else 0 for a, b in zip(t0, t1)])
for lo, hi in ((0, 8), (8, 32), (32, 96), (96, 257)):
    sel = (cm >= lo) & (cm < hi)
    if sel.sum() > 10:
        print(f"  main loops that saw {lo:3d}-{hi:3d} concurrent epilogues: {sel.sum():5d}, mean {ml[sel].mean():.2f} us; their epilogues {ep[sel].mean():.2f} us")
if os.environ.get("TRACE_DUMP"):
    np.save(os.environ["TRACE_DUMP"], np.stack([t0, t1, t2, slot.astype(np.float64)], 1))
