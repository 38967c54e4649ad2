"""CPU tests of the C-ABI shared library: it loads, exports every symbol that
include/sgclip.h declares, agrees with the ctypes mirror on struct sizes, and
rejects bad arguments before touching a device (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from laser_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "sgclip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgclip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in sgclip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in laser_amd/_lib.py"
    assert set(_lib.SIGNATURES) <= set(names)


def test_struct_layouts(lib):
    assert C.sizeof(_lib.CropEntry) == 56
    assert C.sizeof(_lib.TowerDesc) == 48
    assert lib.sgclip_abi_version() == 1
    assert lib.sgclip_patch_kpad(32) == 3072 and lib.sgclip_patch_kpad(14) == 640 and lib.sgclip_patch_kpad(16) == 768


def test_tower_layout_is_a_partition(lib):
    from laser_amd.config import TINY, VIT_B32, VIT_L14_336
    from laser_amd.ops import tower_desc
    for cfg in (TINY, VIT_B32, VIT_L14_336):
        for kind in ("vision", "text"):
            d = tower_desc(cfg, kind)
            nb, nf, isb = C.c_int64(), C.c_int64(), C.c_int()
            assert lib.sgclip_tower_layout(C.byref(d), -1, 0, C.byref(nb), C.byref(nf), C.byref(isb)) == 0
            spans = {0: [], 1: []}
            layers = cfg.v_layers if kind == "vision" else cfg.t_layers
            for tid in range(21):
                for layer in range(layers):
                    off, cnt = C.c_int64(), C.c_int64()
                    rc = lib.sgclip_tower_layout(C.byref(d), tid, layer, C.byref(off), C.byref(cnt), C.byref(isb))
                    if rc != 0:
                        break
                    spans[isb.value].append((off.value, off.value + cnt.value))
                    if lib.sgclip_tower_layout(C.byref(d), tid, layer + 1, None, None, None) != 0 and layer == 0:
                        break
            for kind_bf, total in ((1, nb.value), (0, nf.value)):
                s = sorted(set(spans[kind_bf]))
                assert s[0][0] == 0 and s[-1][1] == total
                for (a0, a1), (b0, b1) in zip(s, s[1:]):
                    assert a1 == b0, "tensors must tile the blob without gaps or overlap"
    # vision B/32 parameter count: matrices + vectors = the HF vision tower + projection
    d = tower_desc(VIT_B32, "vision")
    lib.sgclip_tower_layout(C.byref(d), -1, 0, C.byref(nb), C.byref(nf), C.byref(isb))
    assert nb.value + nf.value == 87_849_216


def test_argument_validation_needs_no_device(lib):
    rc = lib.sgclip_gemm_bf16(None, 64, None, 64, None, 64, 8, 8, 64, 0, None, None, 0, None)
    assert rc == _lib.EINVAL and b"sgclip_gemm_bf16" in lib.sgclip_last_error()
    rc = lib.sgclip_gemm_bf16(16, 100, 16, 100, 16, 8, 8, 8, 100, 0, 16, None, 0, None)   # K % 64 != 0
    assert rc == _lib.EINVAL
    rc = lib.sgclip_layernorm(None, 0, None, 4, 30, None, None, 1e-5, 0, None, None)
    assert rc == _lib.EINVAL
    rc = lib.sgclip_similarity(16, 16, 4, 4, 30, 1.0, 0, 16, 16, 1 << 20, None)            # E % 16 != 0
    assert rc == _lib.EINVAL
    ent = (_lib.CropEntry * 1)()
    ent[0].frame_w, ent[0].frame_h, ent[0].x1, ent[0].x2, ent[0].y1, ent[0].y2, ent[0].mode = 10, 10, 5, 5, 0, 4, 2
    luts = (C.c_uint8 * 768)()
    nl = (C.c_float * 768)()
    rc = lib.sgclip_crop_preprocess(16, 400, None, 0, C.addressof(ent), 16, 0, 1, luts, nl, 16, 64, 0, 0, 0, 0, 16, None)
    assert rc == _lib.EBBOX and b"image bbox issue" in lib.sgclip_last_error()
    with pytest.raises(AssertionError):
        _lib.check(rc)
    assert lib.sgclip_vit_workspace_bytes(None, 4) == 0
