"""ctypes binding of libcorrla_rsvd.so (include/corrla_rsvd.h).  There is no fallback: if the
HIP library is missing or cannot be loaded, importing the compute API raises."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# CORRLA_RSVD_LIB: another build of the same library (kernel A/B measurements)
LIB_PATH = os.environ.get("CORRLA_RSVD_LIB") or os.path.join(PKG, "lib", "libcorrla_rsvd.so")

i64, u64, u32, i32, dbl, flt = C.c_int64, C.c_uint64, C.c_uint32, C.c_int32, C.c_double, C.c_float
vp = C.c_void_p

OK, EINVAL, ENOMEM, EHIP, ECOMM, ENUMERIC, ENODEV = range(7)
OMEGA_ON_DEVICE = 0x1
PCA_CENTER_FUSED = 0x2
PCA_CENTER_COPY = 0x4
QR_HOUSEHOLDER = 0x8
SEED_EXPLICIT = 0x10
POWER_FUSED = 0x20
SHARD_COLS = 0x40
SKETCH_BF16X3 = 0x80
SKETCH_BF16X6 = 0x100
UNIQUE_ID_BYTES = 128


class Opts(C.Structure):
    _fields_ = [("struct_size", u32), ("flags", u32), ("seed", u64), ("omega", vp), ("omega_ld", i64)]


class Timings(C.Structure):
    _fields_ = [("total_ms", dbl), ("sketch_ms", dbl), ("power_ms", dbl), ("qr_ms", dbl), ("project_ms", dbl),
                ("small_svd_ms", dbl), ("finalize_ms", dbl), ("qr_passes", i32), ("n_collectives", i32),
                ("sketch_kernel_ms", dbl), ("host_enqueue_ms", dbl), ("collective_bytes", dbl), ("n_mixed_products", i32),
                ("reserved_", i32), ("knn_ms", dbl), ("fit_ms", dbl)]


# symbol -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
def _sigs():
    s = {
        "corrla_version": (C.c_char_p, []),
        "corrla_last_error": (C.c_char_p, []),
        "corrla_device_count": (C.c_int, []),
        "corrla_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "corrla_ctx_destroy": (None, [vp]),
        "corrla_ctx_synchronize": (C.c_int, [vp]),
        "corrla_ctx_get_timings": (C.c_int, [vp, C.POINTER(Timings)]),
        "corrla_ctx_set_phase_timings": (C.c_int, [vp, C.c_int]),
        "corrla_ctx_comm_info": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
        "corrla_comm_unique_id": (C.c_int, [vp]),
        "corrla_ctx_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    }
    for suf, sc in (("f32", flt), ("f64", dbl)):
        rsvd = [vp, vp, i64, i64, i64, i64, i64, i64, i64, C.POINTER(Opts), vp, i64, vp, vp, i64]
        s["corrla_rsvd_" + suf] = (C.c_int, rsvd)
        s["corrla_rsvd_dev_" + suf] = (C.c_int, rsvd)
        s["corrla_rsvd_sharded_dev_" + suf] = (C.c_int, rsvd)
        pca = [vp, vp, i64, i64, i64, i64, i64, i64, i64, C.POINTER(Opts), vp, vp, vp, i64]
        s["corrla_pca_" + suf] = (C.c_int, pca)
        s["corrla_pca_dev_" + suf] = (C.c_int, pca)
        s["corrla_pca_sharded_dev_" + suf] = (C.c_int, pca)
        pw = [vp, vp, i64, i64, i64, i64, i64, i64, C.POINTER(Opts), vp, i64]
        s["corrla_power_iter_" + suf] = (C.c_int, pw)
        s["corrla_power_iter_dev_" + suf] = (C.c_int, pw)
        s["corrla_matmul_dev_" + suf] = (C.c_int, [vp, C.c_int, vp, i64, i64, i64, i64, vp, i64, i64, sc, vp, i64])
        s["corrla_fill_normal_dev_" + suf] = (C.c_int, [vp, vp, i64, i64, i64, i64, u64, i64, i64])
        s["corrla_time_sketch_dev_" + suf] = (C.c_int, [vp, vp, i64, i64, i64, i64, vp, i64, i64, vp, i64, C.c_int,
                                                         C.POINTER(dbl)])
    grad = [vp, vp, i64, i64, vp, vp, i64, C.c_int, i64, dbl, vp, i64, C.POINTER(C.c_int)]
    s["corrla_grad_mat_f64"] = (C.c_int, grad)
    s["corrla_grad_mat_dev_f64"] = (C.c_int, grad)
    return s


SIGNATURES = _sigs()
_lib = None


class CorrlaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"corrla_rsvd error {code}: {msg}")
        self.code = code


def load():
    """Load the HIP library.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m corrla_rs_amd.build` (hipcc, gfx950). "
            "corrla_rs_amd has no CPU fallback.")
    # torch wheels bundle their own ROCm runtime (libamdhip64 / libhsa-runtime64 / librccl, same SONAMEs as
    # /opt/rocm).  One process must use ONE of them: when torch is installed, import it first so this
    # library binds to the copies torch already loaded (device pointers and RCCL are then shared with
    # torch.distributed); RTLD_GLOBAL keeps the reverse order working when torch is absent at load time.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != OK:
        msg = load().corrla_last_error().decode("utf-8", "replace")
        if code == EINVAL:
            raise ValueError(f"corrla_rsvd: {msg}")
        if code == ENOMEM:
            raise MemoryError(f"corrla_rsvd: {msg}")
        raise CorrlaError(code, msg)
