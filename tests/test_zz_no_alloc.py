"""GPU test of the allocation-free similarity entry point (sgclip_similarity_static): same bits as
sgclip_similarity on two calls in a row, the second smaller than the first (the static scratch keeps the
first call's contents), a size beyond the static maximum is refused, and no device memory is allocated."""
import ctypes as C

import pytest
import torch

from laser_amd import _lib

pytestmark = pytest.mark.gpu


def _both(lib, K, N, E, mode, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    text = torch.randn(K, E, generator=g).cuda()
    img = (torch.randn(N, E, generator=g) * 3).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    want = torch.empty(K, N, device="cuda")
    ws = torch.empty(max(lib.sgclip_similarity_workspace_bytes(K, N), 256), dtype=torch.uint8, device="cuda")
    _lib.check(lib.sgclip_similarity(text.data_ptr(), img.data_ptr(), K, N, E, 100.0, mode, want.data_ptr(), ws.data_ptr(),
                                     ws.numel(), stream))
    got = torch.full((K, N), float("nan"), device="cuda")
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    _lib.check(lib.sgclip_similarity_static(text.data_ptr(), img.data_ptr(), K, N, E, 100.0, mode, got.data_ptr(), stream))
    torch.cuda.synchronize()
    after = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    return got, want, before, after


def test_static_similarity_matches_workspace_similarity_twice():
    lib = _lib.load()
    torch.cuda.set_device(0)
    assert lib.sgclip_similarity_static_max() == 65536
    # first call (also the one that resolves the static array's device address), then a smaller one
    got, want, _, _ = _both(lib, 200, 7680, 512, _lib.SIM_SOFTMAX, 1)
    assert torch.equal(got, want)
    assert (got.sum(0) - 1).abs().max().item() < 1e-4
    got2, want2, before, after = _both(lib, 50, 513, 512, _lib.SIM_SOFTMAX, 2)
    assert torch.equal(got2, want2)
    assert before[0] == after[0] and after[1] >= before[1], "the second call changed the device memory in use"
    for mode in (_lib.SIM_LOGIT, _lib.SIM_SIGMOID):
        g, w, _, _ = _both(lib, 7, 33, 64, mode, 3)
        assert torch.equal(g, w)


def test_static_similarity_rejects_sizes_beyond_its_scratch():
    lib = _lib.load()
    t = torch.zeros(16, 16, device="cuda")
    rc = lib.sgclip_similarity_static(t.data_ptr(), t.data_ptr(), 60000, 6000, 16, 1.0, _lib.SIM_LOGIT, t.data_ptr(), None)
    assert rc == _lib.ENOSPC
    assert b"static scratch" in lib.sgclip_last_error()
