"""Host-side mirror of the reference's RSVD surfaces over the HIP library.

  rsvd(a_mat, n_rank, n_iters, n_oversamples)        <- corrla_rs.rsvd, src/lib_math_utils_py.rs:21-36
  random_svd(a_mat, omega_rank, n_iter, n_oversamples) <- random_svd.rs:63-66 (Rust name / argument names)
  power_iter(a_mat, omega_rank, n_iter)              <- random_svd.rs:15-18

Same positional order (rank, iters, oversamples), same return shapes: U (m, k), S (k, 1) -- a 2-D
column, not 1-D -- and Vt (k, n), column-major in memory like the reference's owned faer `Mat`s
(numpy: F-ordered).  Additive, keyword-only: `seed`, `omega` (shared sketch, the parity-test hook),
`ctx`.  numpy inputs take the host-pointer entry points (H2D + D2H around the device path); torch
CUDA tensors take the device-pointer entry points and return torch tensors on the same device.
float32 input runs the f32 path (the pyo3 surface is f64-only; anything that is not f32 is
converted to f64, as PyReadonlyArray2<f64> extraction would require).
"""
import atexit
import ctypes as C
import sys
import threading
import weakref

import numpy as np

from . import _lib as L

__all__ = ["Context", "default_context", "rsvd", "random_svd", "power_iter", "rpca", "PcaRsvd", "algorithmic_flops"]


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch"


# Contexts own HIP streams / allocations / RCCL communicators: destroy them before the interpreter (and
# the HIP runtime's own static destructors) tear down, never from a late __del__.
_live = weakref.WeakSet()


def _close_all():
    for c in list(_live):
        try:
            c.close()
        except Exception:
            pass


atexit.register(_close_all)


class Context:
    """One device, one stream, one workspace arena (corrla_ctx).  Not thread-parallel: calls on one
    context serialise, as calls on one faer global thread pool do in the reference."""

    def __init__(self, device=0):
        self._lib = L.load()
        h = C.c_void_p()
        L.check(self._lib.corrla_ctx_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)
        _live.add(self)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.corrla_ctx_destroy(h)

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return  # too late to touch the HIP runtime; atexit already closed live contexts
        try:
            self.close()
        except Exception:
            pass

    # ---- communicator ------------------------------------------------------------------
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        L.check(L.load().corrla_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), L.UNIQUE_ID_BYTES)
        L.check(self._lib.corrla_ctx_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_info(self):
        """(rank, nranks) of the context's communicator as RCCL reports them; nranks = 0 without a communicator."""
        r, n = C.c_int(0), C.c_int(0)
        L.check(self._lib.corrla_ctx_comm_info(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def set_phase_timings(self, on):
        """Per-phase device times (hipEvents at the phase boundaries, ~5 us of idle GPU each).  Off: timings() keeps
        total_ms, sketch_kernel_ms and the counters, the phase entries read 0."""
        L.check(self._lib.corrla_ctx_set_phase_timings(self._h, 1 if on else 0))

    def timings(self):
        t = L.Timings()
        L.check(self._lib.corrla_ctx_get_timings(self._h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in t._fields_}

    # ---- helpers -----------------------------------------------------------------------
    @staticmethod
    def _qr_flag(qr):
        """qr=None / "cholesky": the default CholeskyQR2; "householder": Householder TSQR with an explicit thin Q
        (CORRLA_QR_HOUSEHOLDER; the reference's `qr().compute_thin_q()` as written, random_svd.rs:38,57)."""
        if qr in (None, "cholesky"):
            return 0
        if qr == "householder":
            return L.QR_HOUSEHOLDER
        raise ValueError("qr must be None, 'cholesky' or 'householder'")

    @staticmethod
    def _mixed_flag(mixed):
        """mixed=None: exact f32 / f64 products (default).  "bf16x6" / "bf16x3": the tall products of the range finder
        (random_svd.rs:31, 42-51) on the bf16 matrix units with f32 accumulation, every f32 operand split on the fly into
        three / two bf16 pieces (CORRLA_SKETCH_BF16X6 / X3; f32 row-major inputs with l <= 144, ignored elsewhere)."""
        if mixed in (None, False, "f32"):
            return 0
        if mixed == "bf16x6":
            return L.SKETCH_BF16X6
        if mixed == "bf16x3":
            return L.SKETCH_BF16X3
        raise ValueError("mixed must be None, 'bf16x3' or 'bf16x6'")

    def _opts(self, seed, omega, nt, l, dtype, on_device, extra_flags=0):
        if seed is None and omega is None and not extra_flags:
            return None, None
        o = L.Opts()
        o.struct_size = C.sizeof(L.Opts)
        o.seed = int(seed) if seed is not None else 0
        o.flags = int(extra_flags) | (L.SEED_EXPLICIT if seed is not None else 0)   # seed=0 is a real seed
        keep = None
        if omega is not None:
            if on_device:
                import torch
                om = omega if _is_torch(omega) else torch.as_tensor(np.asarray(omega))
                om = om.to(device=f"cuda:{self.device}", dtype=dtype)
                if tuple(om.shape) != (nt, l):
                    raise ValueError(f"omega must have shape {(nt, l)}, got {tuple(om.shape)}")
                keep = om.t().contiguous()  # (l, nt) row-major == (nt, l) column-major
                o.omega = keep.data_ptr()
                o.flags |= L.OMEGA_ON_DEVICE
            else:
                om = np.asarray(omega, dtype=dtype)
                if om.shape != (nt, l):
                    raise ValueError(f"omega must have shape {(nt, l)}, got {om.shape}")
                keep = np.asfortranarray(om)
                o.omega = keep.ctypes.data
            o.omega_ld = nt
        return o, keep

    # ---- random_svd ----------------------------------------------------------------------
    def rsvd(self, a_mat, n_rank, n_iters, n_oversamples, *, seed=None, omega=None, qr=None, fused=False, mixed=None):
        """fused=True: CORRLA_POWER_FUSED (one-sweep A^T (A Z) power iteration; f32 row-major inputs with <= 512 columns).
        mixed="bf16x6" | "bf16x3": see _mixed_flag."""
        n_rank, n_iters, n_oversamples = int(n_rank), int(n_iters), int(n_oversamples)
        if _is_torch(a_mat) and a_mat.is_cuda:
            return self._rsvd_torch(a_mat, n_rank, n_iters, n_oversamples, seed, omega, qr=qr, fused=fused, mixed=mixed)
        a = np.asarray(a_mat.detach().cpu().numpy() if _is_torch(a_mat) else a_mat)
        if a.ndim != 2:
            raise ValueError("a_mat must be 2-D")
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        m, n = a.shape
        if m == 0 or n == 0:
            raise ValueError("a_mat must be non-empty")
        suf = "f32" if a.dtype == np.float32 else "f64"
        if any(s < 0 for s in a.strides):
            a = np.ascontiguousarray(a)
        rs, cs = a.strides[0] // a.itemsize, a.strides[1] // a.itemsize
        k = n_rank
        nt = min(m, n)
        l = min(k + max(n_oversamples, 0), nt)
        o, keep = self._opts(seed, omega, nt, l, a.dtype, False, self._qr_flag(qr) | (L.POWER_FUSED if fused else 0) | self._mixed_flag(mixed))
        kk = max(k, 1)
        u = np.empty((m, kk), dtype=a.dtype, order="F")
        s = np.empty((kk, 1), dtype=a.dtype, order="F")
        vt = np.empty((kk, n), dtype=a.dtype, order="F")
        fn = getattr(self._lib, "corrla_rsvd_" + suf)
        L.check(fn(self._h, a.ctypes.data, m, n, rs, cs, k, n_iters, n_oversamples,
                   C.byref(o) if o is not None else None, u.ctypes.data, m, s.ctypes.data, vt.ctypes.data, kk))
        del keep
        return u, s, vt

    def _rsvd_torch(self, a, k, q, p, seed, omega, sharded=False, qr=None, fused=False, shard_cols=False, mixed=None):
        import torch
        if a.dim() != 2:
            raise ValueError("a_mat must be 2-D")
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float64)
        if a.device.index != self.device:
            raise ValueError(f"tensor is on {a.device}, context is on cuda:{self.device}")
        m, n = a.shape
        # an EMPTY shard (no rows; no columns with shard="cols") is legal on the sharded entry point: the rank takes part
        # in the collectives with zero contributions and gets a 0-row (0-column) block of the sharded factor
        empty_shard = sharded and ((n == 0 and m > 0) if shard_cols else (m == 0 and n > 0))
        if (m == 0 or n == 0) and not empty_shard:
            raise ValueError("a_mat must be non-empty")
        if any(s < 0 for s in a.stride()):
            a = a.contiguous()
        rs, cs = (max(n, 1), 1) if empty_shard else a.stride()
        suf = "f32" if a.dtype == torch.float32 else "f64"
        nt = (m if shard_cols else n) if sharded else min(m, n)
        l = min(k + max(p, 0), nt)
        o, keep = self._opts(seed, omega, nt, l, a.dtype, True, self._qr_flag(qr) | (L.POWER_FUSED if fused else 0) |
                             (L.SHARD_COLS if shard_cols else 0) | self._mixed_flag(mixed))
        kk = max(k, 1)
        dev = a.device
        u = torch.empty((kk, m), dtype=a.dtype, device=dev).t()     # (m, k) column-major
        s = torch.empty((kk, 1), dtype=a.dtype, device=dev)
        vt = torch.empty((n, kk), dtype=a.dtype, device=dev).t()    # (k, n) column-major
        torch.cuda.current_stream(dev).synchronize()  # inputs produced on torch's stream are complete
        name = ("corrla_rsvd_sharded_dev_" if sharded else "corrla_rsvd_dev_") + suf
        fn = getattr(self._lib, name)
        L.check(fn(self._h, a.data_ptr(), m, n, rs, cs, k, q, p, C.byref(o) if o is not None else None,
                   u.data_ptr(), m, s.data_ptr(), vt.data_ptr(), kk))
        del keep
        return u, s, vt

    def rsvd_sharded(self, a_local, n_rank, n_iters, n_oversamples, *, seed=None, omega=None, fused=False, shard="rows", qr=None,
                     mixed=None):
        """Sharded random_svd, one process per GPU.  shard="rows": `a_local` holds this rank's
        rows of a TALL matrix (torch CUDA tensor); returns (U_local, S, Vt) with S, Vt replicated.  shard="cols":
        `a_local` holds this rank's COLUMNS of a FAT matrix; returns (U, S, Vt_local) with U, S replicated
        (CORRLA_SHARD_COLS: the long side is what gets sharded, after the reference's fat -> tall transpose)."""
        if shard not in ("rows", "cols"):
            raise ValueError("shard must be 'rows' or 'cols'")
        return self._rsvd_torch(a_local, int(n_rank), int(n_iters), int(n_oversamples), seed, omega, sharded=True, fused=fused,
                                shard_cols=(shard == "cols"), qr=qr, mixed=mixed)

    # ---- PCA caller (pca_rsvd.rs:56-82) ---------------------------------------------------
    def pca_sharded(self, x_local, rank, n_iter=None, n_oversamples=None, *, seed=None, omega=None, center=None):
        """PcaRsvd::new on SAMPLE-sharded data (one process per GPU, `comm_init` done): `x_local` = this rank's samples
        (CUDA tensor m_local x n_dim).  Returns (means, S, components), replicated on every rank."""
        import torch
        if not (_is_torch(x_local) and x_local.is_cuda):
            raise ValueError("pca_sharded takes torch CUDA tensors")
        cflags = {None: 0, "fused": L.PCA_CENTER_FUSED, "copy": L.PCA_CENTER_COPY}[center]
        x = x_local if x_local.dtype in (torch.float32, torch.float64) else x_local.to(torch.float64)
        m, n = x.shape
        rank = int(rank)
        q = 20 if n_iter is None else int(n_iter)
        p = min(n, 10) if n_oversamples is None else int(n_oversamples)
        rs, cs = x.stride()
        l = min(rank + max(p, 0), n)
        o, keep = self._opts(seed, omega, n, l, x.dtype, True, cflags)
        kk = max(rank, 1)
        means = torch.empty((1, n), dtype=x.dtype, device=x.device)
        s = torch.empty((kk, 1), dtype=x.dtype, device=x.device)
        comps = torch.empty((n, kk), dtype=x.dtype, device=x.device).t()
        torch.cuda.current_stream(x.device).synchronize()
        fn = getattr(self._lib, "corrla_pca_sharded_dev_" + ("f32" if x.dtype == torch.float32 else "f64"))
        L.check(fn(self._h, x.data_ptr(), m, n, rs, cs, rank, q, p, C.byref(o) if o is not None else None,
                   means.data_ptr(), s.data_ptr(), comps.data_ptr(), kk))
        del keep
        return means, s, comps

    def pca(self, x_mat, rank, n_iter=None, n_oversamples=None, *, seed=None, omega=None, center=None):
        """PcaRsvd::new(x, rank): returns (means (1, n), singular values (k, 1), components (k, n)).
        n_iter / n_oversamples default to the reference's hard-coded 20 / min(n_dim, 10) (pca_rsvd.rs:65-66).
        center: None (library default: implicit rank-1 corrections for f64, a centred copy for f32), "fused" or
        "copy" (CORRLA_PCA_CENTER_* in include/corrla_rsvd.h)."""
        rank = int(rank)
        if center not in (None, "fused", "copy"):
            raise ValueError("center must be None, 'fused' or 'copy'")
        cflags = {None: 0, "fused": L.PCA_CENTER_FUSED, "copy": L.PCA_CENTER_COPY}[center]
        if _is_torch(x_mat) and x_mat.is_cuda:
            import torch
            x = x_mat if x_mat.dtype in (torch.float32, torch.float64) else x_mat.to(torch.float64)
            if x.dim() != 2:
                raise ValueError("x_mat must be 2-D")
            m, n = x.shape
            q = 20 if n_iter is None else int(n_iter)
            p = min(n, 10) if n_oversamples is None else int(n_oversamples)
            rs, cs = x.stride()
            nt = min(m, n)
            l = min(rank + max(p, 0), nt)
            o, keep = self._opts(seed, omega, nt, l, x.dtype, True, cflags)
            kk = max(rank, 1)
            means = torch.empty((1, n), dtype=x.dtype, device=x.device)
            s = torch.empty((kk, 1), dtype=x.dtype, device=x.device)
            comps = torch.empty((n, kk), dtype=x.dtype, device=x.device).t()
            torch.cuda.current_stream(x.device).synchronize()
            fn = getattr(self._lib, "corrla_pca_dev_" + ("f32" if x.dtype == torch.float32 else "f64"))
            L.check(fn(self._h, x.data_ptr(), m, n, rs, cs, rank, q, p, C.byref(o) if o is not None else None,
                       means.data_ptr(), s.data_ptr(), comps.data_ptr(), kk))
            del keep
            return means, s, comps
        x = np.asarray(x_mat.detach().cpu().numpy() if _is_torch(x_mat) else x_mat)
        if x.ndim != 2:
            raise ValueError("x_mat must be 2-D")
        if x.dtype != np.float32:
            x = x.astype(np.float64, copy=False)
        if any(st < 0 for st in x.strides):
            x = np.ascontiguousarray(x)
        m, n = x.shape
        q = 20 if n_iter is None else int(n_iter)
        p = min(n, 10) if n_oversamples is None else int(n_oversamples)
        nt = min(m, n)
        l = min(rank + max(p, 0), nt)
        o, keep = self._opts(seed, omega, nt, l, x.dtype, False, cflags)
        kk = max(rank, 1)
        means = np.empty((1, n), dtype=x.dtype)
        s = np.empty((kk, 1), dtype=x.dtype)
        comps = np.empty((kk, n), dtype=x.dtype, order="F")
        fn = getattr(self._lib, "corrla_pca_" + ("f32" if x.dtype == np.float32 else "f64"))
        L.check(fn(self._h, x.ctypes.data, m, n, x.strides[0] // x.itemsize, x.strides[1] // x.itemsize, rank, q, p,
                   C.byref(o) if o is not None else None, means.ctypes.data, s.ctypes.data, comps.ctypes.data, kk))
        del keep
        return means, s, comps

    # ---- active-subspace gradient stage ---------------------------------------
    def grad_mat(self, x_mat, y, est_order, n_nbrs, x_query=None, *, scale=1.0):
        """``ActiveSsRsvd::create_grad_mat`` with a ``PolyGradientEstimator(x_mat, y, est_order, n_nbrs)``
        (active_subspaces.rs:66-141, 215-229): returns (G, n_regularised) with G the k x n_q gradient matrix (column i =
        gradient at query i; queries default to the support points), scaled by `scale`.  numpy in -> numpy out;
        torch CUDA tensors in -> torch CUDA tensor out (no host copies)."""
        est_order, n_nbrs = int(est_order), int(n_nbrs)
        nreg = C.c_int(0)
        if _is_torch(x_mat) and x_mat.is_cuda:
            import torch
            x = x_mat.to(torch.float64).contiguous()
            yv = (y if _is_torch(y) else torch.as_tensor(np.asarray(y))).to(device=x.device, dtype=torch.float64).reshape(-1).contiguous()
            xq = x if x_query is None else x_query.to(device=x.device, dtype=torch.float64).contiguous()
            if x.dim() != 2 or xq.dim() != 2 or xq.shape[1] != x.shape[1] or yv.numel() != x.shape[0]:
                raise ValueError("x_mat (n, k), y (n,), x_query (n_q, k) expected")
            g = torch.empty((xq.shape[0], x.shape[1]), dtype=torch.float64, device=x.device)
            torch.cuda.current_stream(x.device).synchronize()
            L.check(self._lib.corrla_grad_mat_dev_f64(self._h, x.data_ptr(), x.shape[0], x.shape[1], yv.data_ptr(), xq.data_ptr(),
                                                      xq.shape[0], est_order, n_nbrs, float(scale), g.data_ptr(), x.shape[1],
                                                      C.byref(nreg)))
            return g.t(), nreg.value
        x = np.ascontiguousarray(np.asarray(x_mat, dtype=np.float64))
        yv = np.ascontiguousarray(np.asarray(y, dtype=np.float64).reshape(-1))
        xq = x if x_query is None else np.ascontiguousarray(np.asarray(x_query, dtype=np.float64))
        if x.ndim != 2 or xq.ndim != 2 or xq.shape[1] != x.shape[1] or yv.size != x.shape[0]:
            raise ValueError("x_mat (n, k), y (n,), x_query (n_q, k) expected")
        g = np.empty((xq.shape[0], x.shape[1]), dtype=np.float64)
        L.check(self._lib.corrla_grad_mat_f64(self._h, x.ctypes.data, x.shape[0], x.shape[1], yv.ctypes.data, xq.ctypes.data,
                                              xq.shape[0], est_order, n_nbrs, float(scale), g.ctypes.data, x.shape[1],
                                              C.byref(nreg)))
        return g.T, nreg.value

    # ---- power_iter ----------------------------------------------------------------------
    def power_iter(self, a_mat, omega_rank, n_iter, *, seed=None, omega=None, qr=None):
        a = np.asarray(a_mat)
        if a.ndim != 2:
            raise ValueError("a_mat must be 2-D")
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        m, n = a.shape
        w = int(omega_rank)
        suf = "f32" if a.dtype == np.float32 else "f64"
        rs, cs = a.strides[0] // a.itemsize, a.strides[1] // a.itemsize
        o, keep = self._opts(seed, omega, n, w, a.dtype, False, self._qr_flag(qr))
        q = np.empty((m, max(w, 1)), dtype=a.dtype, order="F")
        fn = getattr(self._lib, "corrla_power_iter_" + suf)
        L.check(fn(self._h, a.ctypes.data, m, n, rs, cs, w, int(n_iter), C.byref(o) if o is not None else None,
                   q.ctypes.data, m))
        del keep
        return q

    # ---- low-level hooks used by tests / bench ---------------------------------------------
    def matmul(self, a, x, trans=False, beta=1.0):
        """res = beta * op(a) @ x on device tensors (par_matmul_helper, mat_utils.rs:20-33)."""
        import torch
        m, n = a.shape
        rs, cs = a.stride()
        xin, xout = (m, n) if trans else (n, m)
        assert x.shape[0] == xin and x.dtype == a.dtype
        l = x.shape[1]
        xc = x.t().contiguous()  # column-major (xin, l)
        res = torch.empty((l, xout), dtype=a.dtype, device=a.device)
        suf = "f32" if a.dtype == torch.float32 else "f64"
        torch.cuda.current_stream(a.device).synchronize()
        fn = getattr(self._lib, "corrla_matmul_dev_" + suf)
        L.check(fn(self._h, 1 if trans else 0, a.data_ptr(), m, n, rs, cs, xc.data_ptr(), xin, l, beta,
                   res.data_ptr(), xout))
        return res.t()

    def fill_normal(self, t, seed, row0=0, global_cols=None):
        """In-place N(0,1) fill of a 2-D device tensor (random_mat_normal, mat_utils.rs:161-175)."""
        import torch
        rows, cols = t.shape
        rs, cs = t.stride()
        suf = "f32" if t.dtype == torch.float32 else "f64"
        torch.cuda.current_stream(t.device).synchronize()
        fn = getattr(self._lib, "corrla_fill_normal_dev_" + suf)
        L.check(fn(self._h, t.data_ptr(), rows, cols, rs, cs, int(seed), int(row0),
                   int(global_cols if global_cols is not None else cols)))
        return t

    def time_sketch(self, a, x, reps=10):
        """Average duration (ms) of the sketch GEMM Y = A @ X measured with hipEvents on the library's
        stream; returns (ms, Y)."""
        import torch
        m, n = a.shape
        rs, cs = a.stride()
        l = x.shape[1]
        xc = x.t().contiguous()
        y = torch.empty((l, m), dtype=a.dtype, device=a.device)
        ms = C.c_double()
        suf = "f32" if a.dtype == torch.float32 else "f64"
        torch.cuda.current_stream(a.device).synchronize()
        fn = getattr(self._lib, "corrla_time_sketch_dev_" + suf)
        L.check(fn(self._h, a.data_ptr(), m, n, rs, cs, xc.data_ptr(), n, l, y.data_ptr(), m, int(reps), C.byref(ms)))
        return ms.value, y.t()


_default = None
_default_lock = threading.Lock()


def default_context():
    global _default
    with _default_lock:
        if _default is None:
            _default = Context(0)
        return _default


def rsvd(a_mat, n_rank, n_iters, n_oversamples, *, seed=None, omega=None, ctx=None, qr=None):
    """corrla_rs.rsvd(a_mat, n_rank, n_iters, n_oversamples) -> (U (m,k), S (k,1), Vt (k,n)).
    src/lib_math_utils_py.rs:21-36.  qr="householder" selects the Householder TSQR thin-Q (default: CholeskyQR2)."""
    return (ctx or default_context()).rsvd(a_mat, n_rank, n_iters, n_oversamples, seed=seed, omega=omega, qr=qr)


def random_svd(a_mat, omega_rank, n_iter, n_oversamples, *, seed=None, omega=None, ctx=None, qr=None):
    """random_svd(a_mat, omega_rank, n_iter, n_oversamples), random_svd.rs:63-66."""
    return rsvd(a_mat, omega_rank, n_iter, n_oversamples, seed=seed, omega=omega, ctx=ctx, qr=qr)


def power_iter(a_mat, omega_rank, n_iter, *, seed=None, omega=None, ctx=None, qr=None):
    """power_iter(a_mat, omega_rank, n_iter) -> Q (m, omega_rank), random_svd.rs:15-18.  `omega_rank` is
    the already-oversampled sketch width."""
    return (ctx or default_context()).power_iter(a_mat, omega_rank, n_iter, seed=seed, omega=omega, qr=qr)


def rpca(a_mat, n_rank, n_iters=None, n_oversamples=None, *, seed=None, omega=None, ctx=None):
    """pyo3 ``rpca(a_mat, n_rank, n_iters, n_oversamples) -> (singular_values (k, 1), components (k, n))``,
    src/lib_math_utils_py.rs:38-55.  Like the reference, the last two positional arguments are accepted and
    IGNORED: PcaRsvd::new hard-codes n_iter = 20 and n_oversamples = min(n_dim, 10) (pca_rsvd.rs:65-66)."""
    del n_iters, n_oversamples
    _means, s, comps = (ctx or default_context()).pca(a_mat, n_rank, seed=seed, omega=omega)
    return s, comps


class PcaRsvd:
    """Mirror of ``PcaRsvd`` (src/lib_math_utils/pca_rsvd.rs:13-112): fit on construction, keeps the means, the
    singular values (k, 1) and ``components_`` (k, n_dim).  The fit (means, centring, RSVD) runs on the GPU; the
    two k-wide projections below are plain numpy on the stored k x n_dim factors."""

    def __init__(self, x_mat, rank, *, seed=None, omega=None, ctx=None):
        self.pca_rank = int(rank)
        x = np.asarray(x_mat)
        self.n_samples = x.shape[0]
        self.means, self.pca_s, self.components_ = (ctx or default_context()).pca(x, rank, seed=seed, omega=omega)

    def fit(self, x_mat, rank, **kw):  # pca_rsvd.rs:85-88
        self.__init__(x_mat, rank, **kw)

    def explained_var(self):  # pca_rsvd.rs:91-99: s^2 / (n_samples - 1)
        return self.pca_s * self.pca_s / (self.n_samples - 1.0)

    def components(self):
        return self.components_

    def singular_values(self):
        return self.pca_s

    def apply_tr(self, targ_mat):  # pca_rsvd.rs:43-46: centres the TARGET by its own column means
        t = np.asarray(targ_mat, dtype=self.components_.dtype)
        return (t - t.mean(axis=0, keepdims=True)) @ self.components_.T

    def apply_inv_tr(self, red_mat):  # pca_rsvd.rs:49-52
        return np.asarray(red_mat, dtype=self.components_.dtype) @ self.components_ + self.means


def algorithmic_flops(m, n, k, q, p):
    """(4q+4) m n l + 2 m l^2 + (1 + max(0, q-3)) (4 m l^2 - 4/3 l^3), unpadded l."""
    if m < n:
        m, n = n, m
    l = min(k + p, n)
    return (4 * q + 4) * m * n * l + 2.0 * m * l * l + (1 + max(0, q - 3)) * (4.0 * m * l * l - 4.0 / 3.0 * l ** 3)
