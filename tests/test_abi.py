"""The C-ABI library loads on a machine without a GPU, exports every symbol include/corrla_rsvd.h
declares, and its compute entry points fail loudly (no CPU fallback) when no gfx950 device exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from corrla_rs_amd import _lib as L
from corrla_rs_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    B.build_product()
    return L.load()


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "corrla_rsvd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(corrla_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_all_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/corrla_rsvd.h but not exported"
        assert s in L.SIGNATURES, f"{s} has no ctypes signature in corrla_rs_amd/_lib.py"
    for s in L.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"


def test_struct_layouts_match_header(lib):
    assert C.sizeof(L.Opts) == 32          # u32 u32 u64 ptr i64
    assert C.sizeof(L.Timings) == 7 * 8 + 8 + 8


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.corrla_version()
    assert isinstance(lib.corrla_last_error(), bytes)


def test_no_cpu_fallback_without_device(lib):
    if lib.corrla_device_count() > 0:
        pytest.skip("a GPU is visible: the ENODEV path cannot be exercised here")
    h = C.c_void_p()
    rc = lib.corrla_ctx_create(0, C.byref(h))
    assert rc == L.ENODEV and not h.value
    assert b"no HIP device" in lib.corrla_last_error() or b"fallback" in lib.corrla_last_error()
    import corrla_rs_amd as cr
    with pytest.raises(L.CorrlaError):
        cr.Context(0)
    # NULL context -> EINVAL, not a crash, not a CPU computation
    a = np.ones((4, 4))
    u = np.empty((4, 2), order="F"); s = np.empty((2, 1)); vt = np.empty((2, 4), order="F")
    rc = lib.corrla_rsvd_f64(None, a.ctypes.data, 4, 4, 4, 1, 2, 1, 1, None, u.ctypes.data, 4, s.ctypes.data,
                             vt.ctypes.data, 2)
    assert rc == L.EINVAL


def test_product_sources_do_not_reference_oracle_or_emulation():
    """The product path must not import, link or execute anything under oracle/ or tests/."""
    pkg = os.path.join(ROOT, "corrla_rs_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(base, f)).read()
                assert "rsvd_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f
                assert "emu_backend" not in txt and "libcorrla_emu" not in txt or f == "build.py", f
