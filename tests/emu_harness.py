"""ctypes harness for tests/emu/libcorrla_emu.so (TEST INFRASTRUCTURE: host emulation of the device
backend so the real driver.hpp / capi_impl.hpp logic runs without a GPU)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from corrla_rs_amd import _lib as L  # noqa: E402
from corrla_rs_amd import build as B  # noqa: E402

_emu = None
ALLREDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_int)


def emu():
    global _emu
    if _emu is None:
        path = B.build_emu(asan=os.environ.get("CORRLA_EMU_ASAN", "0") == "1")
        _emu = C.CDLL(path)
        _emu.corrla_emu_last_error.restype = C.c_char_p
    return _emu


def _opts(omega, nt, l, dtype, seed=None, flags=0):
    if omega is None and seed is None and not flags:
        return None, None
    o = L.Opts()
    o.struct_size = C.sizeof(L.Opts)
    o.seed = int(seed or 0)
    o.flags = int(flags)
    keep = None
    if omega is not None:
        keep = np.asfortranarray(np.asarray(omega, dtype=dtype))
        assert keep.shape == (nt, l), (keep.shape, (nt, l))
        o.omega = keep.ctypes.data
        o.omega_ld = nt
    return o, keep


def emu_rsvd(a, k, q, p, omega=None, seed=None, sharded=False, return_passes=False, qr=None, fused=False, shard_cols=False,
             bad_ldu=False):
    e = emu()
    a = np.asarray(a)
    dtype = a.dtype
    suf = "f32" if dtype == np.float32 else "f64"
    m, n = a.shape
    rs, cs = a.strides[0] // a.itemsize, a.strides[1] // a.itemsize
    if a.size == 0:      # an empty shard of a sharded call: strides are immaterial
        rs, cs = max(n, 1), 1
    nt = (m if shard_cols else n) if sharded else min(m, n)
    l = min(k + p, nt)
    o, keep = _opts(omega, nt, l, dtype, seed, (L.QR_HOUSEHOLDER if qr == "householder" else 0) | (L.POWER_FUSED if fused else 0) |
                    (L.SHARD_COLS if shard_cols else 0))
    u = np.empty((m, max(k, 1)), dtype=dtype, order="F")
    s = np.empty((max(k, 1), 1), dtype=dtype, order="F")
    vt = np.empty((max(k, 1), n), dtype=dtype, order="F")
    i64 = C.c_int64
    passes = C.c_int(0)
    args = [C.c_void_p(a.ctypes.data), i64(m), i64(n), i64(rs), i64(cs), i64(k), i64(q), i64(p),
            C.byref(o) if o is not None else None, C.c_void_p(u.ctypes.data), i64(m - 1 if bad_ldu else m), C.c_void_p(s.ctypes.data),
            C.c_void_p(vt.ctypes.data), i64(max(k, 1))]
    if sharded:
        rc = getattr(e, "corrla_emu_rsvd_sharded_" + suf)(*args)
    else:
        rc = getattr(e, "corrla_emu_rsvd_" + suf)(*args, C.byref(passes))
    if rc != 0:
        msg = e.corrla_emu_last_error().decode()
        if rc == 1:
            raise ValueError(msg)
        raise RuntimeError(f"emu error {rc}: {msg}")
    if return_passes:
        return u, s, vt, passes.value
    return u, s, vt


def emu_power_iter(a, width, q, omega=None, qr=None):
    e = emu()
    a = np.asarray(a)
    suf = "f32" if a.dtype == np.float32 else "f64"
    m, n = a.shape
    rs, cs = a.strides[0] // a.itemsize, a.strides[1] // a.itemsize
    o, keep = _opts(omega, n, width, a.dtype, None, L.QR_HOUSEHOLDER if qr == "householder" else 0)
    qm = np.empty((m, width), dtype=a.dtype, order="F")
    i64 = C.c_int64
    rc = getattr(e, "corrla_emu_power_iter_" + suf)(C.c_void_p(a.ctypes.data), i64(m), i64(n), i64(rs), i64(cs),
                                                    i64(width), i64(q), C.byref(o) if o is not None else None,
                                                    C.c_void_p(qm.ctypes.data), i64(m))
    if rc != 0:
        raise (ValueError if rc == 1 else RuntimeError)(e.corrla_emu_last_error().decode())
    return qm


def emu_matmul(a, x, trans, beta=1.0):
    e = emu()
    a = np.asarray(a)
    suf = "f32" if a.dtype == np.float32 else "f64"
    m, n = a.shape
    rs, cs = a.strides[0] // a.itemsize, a.strides[1] // a.itemsize
    xin, xout = (m, n) if trans else (n, m)
    xf = np.asfortranarray(np.asarray(x, dtype=a.dtype))
    l = xf.shape[1]
    res = np.empty((xout, l), dtype=a.dtype, order="F")
    i64 = C.c_int64
    sc = C.c_float if a.dtype == np.float32 else C.c_double
    rc = getattr(e, "corrla_emu_matmul_" + suf)(C.c_int(1 if trans else 0), C.c_void_p(a.ctypes.data), i64(m), i64(n),
                                                i64(rs), i64(cs), C.c_void_p(xf.ctypes.data), i64(xin), i64(l),
                                                sc(beta), C.c_void_p(res.ctypes.data), i64(xout))
    if rc != 0:
        raise RuntimeError(e.corrla_emu_last_error().decode())
    return res


def emu_fill_normal(rows, cols, seed, dtype=np.float64, row0=0, global_cols=None, order="C"):
    e = emu()
    out = np.empty((rows, cols), dtype=dtype, order=order)
    rs, cs = out.strides[0] // out.itemsize, out.strides[1] // out.itemsize
    suf = "f32" if dtype == np.float32 else "f64"
    i64 = C.c_int64
    rc = getattr(e, "corrla_emu_fill_normal_" + suf)(C.c_void_p(out.ctypes.data), i64(rows), i64(cols), i64(rs), i64(cs),
                                                     C.c_uint64(seed), i64(row0), i64(global_cols or cols))
    assert rc == 0
    return out


def emu_pca(x, rank, q, p, omega=None, center=None, sharded=False):
    """center: None (library default), "fused" (CORRLA_PCA_CENTER_FUSED = 0x2) or "copy" (0x4)."""
    e = emu()
    x = np.asarray(x)
    suf = "f32" if x.dtype == np.float32 else "f64"
    m, n = x.shape
    rs, cs = x.strides[0] // x.itemsize, x.strides[1] // x.itemsize
    nt = n if sharded else min(m, n)
    l = min(rank + p, nt)
    o, keep = _opts(omega, nt, l, x.dtype)
    if center is not None:
        if o is None:
            o, keep = _opts(np.random.default_rng(0).standard_normal((nt, l)), nt, l, x.dtype)
        o.flags |= {"fused": 0x2, "copy": 0x4, "both": 0x6}[center]
    means = np.empty((1, n), dtype=x.dtype)
    s = np.empty((rank, 1), dtype=x.dtype)
    comps = np.empty((rank, n), dtype=x.dtype, order="F")
    i64 = C.c_int64
    rc = getattr(e, ("corrla_emu_pca_sharded_" if sharded else "corrla_emu_pca_") + suf)(C.c_void_p(x.ctypes.data), i64(m), i64(n), i64(rs), i64(cs), i64(rank), i64(q),
                                             i64(p), C.byref(o) if o is not None else None, C.c_void_p(means.ctypes.data),
                                             C.c_void_p(s.ctypes.data), C.c_void_p(comps.ctypes.data), i64(rank))
    if rc != 0:
        raise (ValueError if rc == 1 else RuntimeError)(e.corrla_emu_last_error().decode())
    return means, s, comps
