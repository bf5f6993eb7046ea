"""not gpu: the C-ABI library loads, exports every symbol include/ecamp_hip.h declares, and validates arguments
before touching a device (no compute is attempted here)."""
import ctypes
import os
import re

import pytest

from ecamp_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not all(os.path.exists(p) for p in _lib.LIB_PATHS.values()):
        from ecamp_amd import build
        build.build(verbose=False, half="both")
    return _lib.load()


def test_header_declares_the_hot_path_entry_points():
    protos = _lib.parse_header()
    must = ["ecamp_gemm", "ecamp_layernorm_fwd", "ecamp_layernorm_bwd", "ecamp_attn_fwd", "ecamp_attn_bwd", "ecamp_bicubic_resize",
            "ecamp_mask_indices", "ecamp_im2col_gather", "ecamp_assemble_tokens", "ecamp_unshuffle_fwd", "ecamp_unshuffle_bwd",
            "ecamp_unpatchify_mim", "ecamp_img_loss_bwd", "ecamp_sr_fwd", "ecamp_sr_bwd", "ecamp_bert_embed_fwd", "ecamp_bert_embed_bwd",
            "ecamp_ce_fwd_bwd", "ecamp_sumsq", "ecamp_adamw_grouped", "ecamp_last_error", "ecamp_abi_version"]
    for m in must:
        assert m in protos, m
    # every prototype cites the reference call site it replaces (file:line) somewhere in the header
    txt = open(_lib.HEADER).read()
    assert len(re.findall(r"\w+\.py:\d+", txt)) >= 20


def test_library_exports_every_declared_symbol(lib):
    for name in _lib.parse_header():
        assert hasattr(lib, name), name
    # the library reports the version of the header it was built against; the binding refuses any other (an older build must never be
    # called with this header's argument lists: ECAMP_LIB / tools/ab_lib.sh)
    assert lib.ecamp_abi_version() == _lib.abi_version_of_header() >= 4


def test_binding_refuses_a_library_of_another_abi_version(lib, monkeypatch):
    v = _lib.abi_version_of_header()
    monkeypatch.setattr(_lib, "abi_version_of_header", lambda path=None: v + 1)   # a header one version ahead of the built library
    monkeypatch.setattr(_lib, "_libs", {})
    with pytest.raises(_lib.EcampHipError, match="ABI version mismatch"):
        _lib.load()


def test_both_builds_export_the_whole_abi_and_say_which_format_they_store(lib, monkeypatch):
    """libecamp_hip.so (bfloat16) and libecamp_hip_f16.so (IEEE half: the reference's autocast format, main_pretrain.py:139) are the same
    sources; each exports every declared symbol, reports its format, and the binding refuses a swapped pair."""
    import torch
    assert lib.ecamp_half_format() == 0
    f16 = _lib.load("f16")
    assert f16.ecamp_half_format() == 1 and f16 is not lib
    for name in _lib.parse_header():
        assert hasattr(f16, name), name
    prev = _lib.set_half(torch.float16)
    try:
        assert _lib.half() == "f16" and _lib.load() is f16
        assert _lib.set_half(torch.float32) == "f16" and _lib.half() == "f16"      # f32 parity mode leaves the choice alone
        from ecamp_amd import hip_ops
        assert hip_ops.code(torch.float16) == _lib.BF16
        with pytest.raises(TypeError, match="set_half"):
            hip_ops.code(torch.bfloat16)                                          # a bf16 tensor must never reach the f16 build
    finally:
        _lib.set_half(prev)
    monkeypatch.setattr(_lib, "_libs", {})
    monkeypatch.setattr(_lib, "LIB_PATHS", {"bf16": _lib.LIB_PATHS["f16"], "f16": _lib.LIB_PATHS["bf16"]})
    with pytest.raises(_lib.EcampHipError, match="swapped"):
        _lib.load("bf16")


def test_signatures_have_no_torch_types():
    txt = re.sub(r"/\*.*?\*/", " ", open(_lib.HEADER).read(), flags=re.S)  # prototypes only, comments stripped
    assert "torch" not in txt.lower() and "at::" not in txt and "Tensor" not in txt
    for _name, (_ret, args) in _lib.parse_header().items():
        for ct, _an in args:
            assert ct in (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_float)


def test_argument_errors_are_reported_without_a_device(lib):
    rc = lib.ecamp_layernorm_fwd(None, None, None, None, None, None, None, None, 4, 768, 1e-6, 0.0, 0, 0, 0, None)
    assert rc < 0 and b"null pointer" in lib.ecamp_last_error()
    one = ctypes.c_void_p(16)  # never dereferenced: the shape check fails first
    rc = lib.ecamp_gemm(one, one, one, 8, 30, 30, 1, 30, 1, 30, 30, None, None, 0, None, 0, None, 0, 0, 1.0, None, 1, 0, 0, 1, None, None, None)
    assert rc < 0 and b"multiple of 4" in lib.ecamp_last_error()
    st = (ctypes.c_int64 * 3)(64, 64, 64)
    rc = lib.ecamp_attn_fwd(one, one, one, one, one, None, 1, 1, 8, 300, 48, st, st, st, st, 1.0, 0.0, 0, 0, 0, None, None)
    assert rc < 0 and b"head_dim 48" in lib.ecamp_last_error()  # (Tk > 256 in f32 is served by the long-key kernels since round 2)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ecamp_amd/ may import, call or execute it."""
    for d, _, files in os.walk(os.path.join(ROOT, "ecamp_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), os.path.join(d, f)


def test_gemm_planning_helpers_are_pure_host_arithmetic(lib):
    """ecamp_gemm_suggest_split / ecamp_set_option (no reference counterpart: they plan the persistent-kernel launches)."""
    BF16 = 1
    # weight gradient of timm Mlp.fc1 at configs[1] (dW[3072,768] += dY^T X over 12800 tokens): enough splits to fill whole
    # rounds of the chip, never more than K/512
    s = lib.ecamp_gemm_suggest_split(3072, 768, 12800, 0, 0, BF16)
    assert 1 <= s <= 25
    assert lib.ecamp_gemm_suggest_split(30000, 768, 32768, 0, 0, BF16) <= 8        # vocab projection: already hundreds of tiles
    assert lib.ecamp_gemm_suggest_split(0, 0, 0, 0, 0, BF16) == 1                    # degenerate shapes do not crash
    assert lib.ecamp_set_option(b"p8_wgrad_reserve_cus", 16) == 0
    s16 = lib.ecamp_gemm_suggest_split(3072, 768, 12800, 0, 0, BF16)
    assert 1 <= s16 <= 25
    assert lib.ecamp_set_option(b"p8_wgrad_reserve_cus", 0) == 0
    assert lib.ecamp_set_option(b"p8_wgrad", 0) == 0 and lib.ecamp_set_option(b"p8_wgrad", 1) == 0
    assert lib.ecamp_set_option(b"no_such_option", 1) < 0 and b"unknown option" in lib.ecamp_last_error()
    assert lib.ecamp_set_option(None, 1) < 0


def test_wgrad_group_item_table_covers_every_k_tile_once(lib):
    """ecamp_wgrad_group_table is pure host arithmetic: for the four linear layers of a ViT-B encoder block (12800 rows) and of a
    BERT layer (32768 rows) every (output tile, K tile) unit is covered by exactly one piece, each workgroup owns one piece, a tile's
    slabs are consecutive, the workgroups of one XCD (w mod 8) hold neighbouring tiles of one K segment, and the data-parallel CU
    reserve caps the piece count."""
    import numpy as np
    import torch
    cv = lambda a: ctypes.cast(a, ctypes.c_void_p)
    # (the third group: a ViT-L/448 encoder block -- 12608 rows = 197 K tiles, an odd count with a partial-free last tile; the fourth: a
    # row count that is no multiple of 64 either)
    for rows, shapes in ((12800, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (32768, [(768, 1536), (1536, 768), (768, 768), (2304, 768)]),
                         (12608, [(1024, 4096), (4096, 1024), (1024, 1024), (3072, 1024)]), (1000, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])):
        for reserve in (0, 32):
            assert lib.ecamp_set_option(b"p8_wgrad_reserve_cus", reserve) == 0
            n = len(shapes)
            no = (ctypes.c_int64 * n)(*[s[0] for s in shapes])
            ki = (ctypes.c_int64 * n)(*[s[1] for s in shapes])
            hb = (ctypes.c_int32 * n)(*([1] * n))
            assert lib.ecamp_wgrad_group_supported(n, cv(no), cv(ki), rows) == 1
            cap = lib.ecamp_wgrad_group_table_bytes(n, cv(no), cv(ki), rows)
            host = torch.zeros((cap,), dtype=torch.uint8)
            used = lib.ecamp_wgrad_group_table(n, cv(no), cv(ki), cv(hb), rows, 0, ctypes.c_void_p(host.data_ptr()))
            assert 0 < used <= cap
            T = sum(((a + 255) // 256) * ((b + 255) // 256) for a, b in shapes)
            P = next(p for p in range(1, 1025) if 32 * p + (4 * (p + 1) + 31) // 32 * 32 + 20 * T == used)   # one piece per workgroup
            assert P <= 256 - reserve and P >= 96
            raw = host.numpy()
            items = raw[:32 * P].view(np.int32).reshape(P, 8)
            first = raw[32 * P:32 * P + 4 * (P + 1)].view(np.int32)
            off_t = 32 * P + (4 * (P + 1) + 31) // 32 * 32
            tiles = raw[off_t:off_t + 20 * T].view(np.int32).reshape(T, 5)
            assert (first == np.arange(P + 1)).all()
            KT = (rows + 63) // 64
            assert all(int(kend) <= rows for _, _, _, _, kend, _, _, _ in items)
            cover = {}
            for prob, m0, n0, kbeg, kend, slab, flags, _ in items:
                assert 0 <= prob < n and m0 % 256 == 0 and n0 % 256 == 0 and kbeg % 64 == 0 and kend - kbeg >= 128
                assert flags == (1 if n0 == 0 else 0)
                for k in range(kbeg // 64, (kend + 63) // 64):
                    key = (int(prob), int(m0), int(n0), k)
                    assert key not in cover
                    cover[key] = int(slab)
            assert len(cover) == T * KT
            for prob, m0, n0, f, c in tiles:
                slabs = sorted({cover[(int(prob), int(m0), int(n0), k)] for k in range(KT)})
                assert slabs == list(range(int(f), int(f) + int(c)))
            # XCD locality: the pieces of workgroups w = x, x + 8, x + 16, ... start at no more than two distinct K offsets
            for x in range(8):
                assert len({int(items[w][3]) for w in range(x, P, 8)}) <= 2
    assert lib.ecamp_set_option(b"p8_wgrad_reserve_cus", 0) == 0


def test_graft_entry_build_is_what_the_driver_runs():
    """`__graft_entry__.build()` -- the driver's "does it build" check -- compiles (incrementally) and loads BOTH libraries and returns;
    `python -m ecamp_amd.build` is the same call (round 6 shipped, for two hours, a build("both") that built everything and then raised)."""
    import subprocess
    import sys
    import __graft_entry__ as g
    g.build()
    assert set(_lib._libs) >= {"bf16", "f16"}
    r = subprocess.run([sys.executable, "-m", "ecamp_amd.build"], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
