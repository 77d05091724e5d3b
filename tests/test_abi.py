"""CPU: the C-ABI library loads and exports every symbol include/ddx_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ddx_hip.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddx_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = _declared_symbols()
    for must in ("ddx_mpconv2d_fwd", "ddx_mpconv_wprep", "ddx_attn_fwd", "ddx_pixelnorm_fwd", "ddx_plan_begin", "ddx_plan_graph_launch"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from dualdiffusion_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        pytest.skip("libddx_hip.so not built (run `make` or __graft_entry__.build())")
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(handle, s)]
    assert not missing, f"declared in ddx_hip.h but not exported: {missing}"
    # and the ctypes prototype table covers the whole header
    unbound = [s for s in _declared_symbols() if s not in _lib.PROTOTYPES]
    assert not unbound, f"no ctypes prototype for: {unbound}"
    assert _lib.lib().ddx_version().decode().startswith("libddx_hip")


def test_descriptor_struct_layout_matches_header():
    """ctypes mirrors of the C descriptors: sizes follow from the field lists in the header (LP64)."""
    from dualdiffusion_amd import _lib
    assert ctypes.sizeof(_lib.WPrepDesc) == 96  # 3 pointers, float, 10 int32, 2 floats, int32 transpose, pointer row_scale, 2 int32
    assert ctypes.sizeof(_lib.WgradDesc) == 88  # 5 pointers, 11 int32 (+4 tail padding)
    assert ctypes.sizeof(_lib.MssDesc) == 112   # 7 pointers, 8 int32, 2 floats, 2 int32, pointer
    assert ctypes.sizeof(_lib.WPathJob) == 120  # 8 pointers, float, 9 int32, 2 floats, 2 int32
    assert ctypes.sizeof(_lib.LinearBwdJob) == 40  # 4 pointers, 2 int32
    assert ctypes.sizeof(_lib.DgradActDesc) == 192 + 7 * 8 + 4 * 4
    assert ctypes.sizeof(_lib.ConvDesc) == 6 * 8 + 12 * 4 + 4 * 4 + 2 * 4 + 2 * 8 + 2 * 4 + 8 + 8 + 2 * 8 + 16   # + pad_mode, prologue_rows | out2_linear, layout, 2 pointers | residual_up, out_head_norm, out_head_eps (+4 tail padding)
    assert ctypes.sizeof(_lib.LinearJob) == 3 * 8 + 2 * 4 + 4 * 4
    lib = _lib.lib() if os.path.isfile(_lib.LIB_PATH) else None
    if lib is not None:
        # every mirror against the COMPILED header: size and the offset of the last field (a mirror one trailing int32 short can
        # hide inside the tail padding of an equal sizeof)
        for which, mirror in enumerate(_lib.ABI_MIRRORS):
            assert lib.ddx_abi_sizeof(which) == ctypes.sizeof(mirror), (which, mirror.__name__)
            last = mirror._fields_[-1][0]
            assert lib.ddx_abi_offsetof_tail(which) == getattr(mirror, last).offset, (which, mirror.__name__, last)
        assert lib.ddx_abi_sizeof(len(_lib.ABI_MIRRORS)) == -1
        # the reference-side binding printed in INTEGRATION.md section 2 is the same struct
        doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
        block = doc[doc.index("class ConvDesc(C.Structure)"):]
        block = block[:block.index("_lib.ddx_mpconv2d_fwd.argtypes")]
        doc_fields = re.findall(r'\("([a-z0-9_A-Z]+)",\s*C\.(c_[a-z0-9_]+)\)', block)
        assert [(n, getattr(ctypes, t)) for n, t in doc_fields] == [(n, t) for n, t in _lib.ConvDesc._fields_]
        # wprep byte count: groups * ceil(Cg/CK) * taps * roundup(Ng,32) * CK * sizeof
        assert lib.ddx_wprep_bytes(512, 32, 3, 8, 32, _lib.DDX_BF16) == 8 * 1 * 9 * 64 * 32 * 2
        assert lib.ddx_wprep_bytes(4, 256, 3, 1, 32, _lib.DDX_F32) == 1 * 8 * 9 * 32 * 32 * 4
        assert lib.ddx_mpconv2d_pick_ck(32, 3, _lib.DDX_BF16, 0) == 32
        assert lib.ddx_mpconv2d_pick_ck(1280, 1, _lib.DDX_BF16, 0) == 128
        assert lib.ddx_mpconv2d_pick_ck(1280, 1, _lib.DDX_BF16, 344) == 64
        assert lib.ddx_mpconv2d_pick_ck(1280, 1, _lib.DDX_F32, 0) == 64


def test_null_descriptor_is_rejected_without_a_gpu():
    from dualdiffusion_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        pytest.skip("library not built")
    lib = _lib.lib()
    assert lib.ddx_mpconv2d_fwd(None, None) == -1      # DDX_ERR_ARG, no launch attempted
    assert b"null" in lib.ddx_last_error()
    d = _lib.ConvDesc()
    assert lib.ddx_mpconv2d_fwd(ctypes.byref(d), None) == -1
