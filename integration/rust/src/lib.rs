//! `extern "C"` binding of include/corrla_rsvd.h plus wrappers that keep the reference's public signatures
//! (wgurecky/CORRLA_RS @ 2024_10_08):
//!
//!   pub fn random_svd<T>(a_mat: MatRef<T>, omega_rank: usize, n_iter: usize, n_oversamples: usize)
//!       -> (Mat<T>, Mat<T>, Mat<T>)                              src/lib_math_utils/random_svd.rs:63-66
//!   pub fn power_iter<T>(a_mat: MatRef<T>, omega_rank: usize, n_iter: usize) -> Mat<T>      random_svd.rs:15-18
//!
//! The `MatRef` is forwarded as (ptr, nrows, ncols, row_stride, col_stride) -- no copy, any strides, exactly the view
//! the pyo3 layer builds from a numpy array (src/lib_math_utils_py.rs:27-28).  The three results are freshly owned
//! column-major `Mat`s like the reference's (random_svd.rs:96-109); S is the k x 1 column.  A non-zero status panics
//! with the library's message, which is what the reference's unwrap()/slice panics do (random_svd.rs:98-107).
//!
//! NOT compiled here (no Rust toolchain in the build image); the C side of every prototype below is checked against
//! the header by tests/test_abi.py through the ctypes table, which this file mirrors one to one.
use faer::{Mat, MatRef};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct CorrlaOpts {
    pub struct_size: u32,
    pub flags: u32,
    pub seed: u64,
    pub omega: *const c_void,
    pub omega_ld: i64,
}
pub const CORRLA_OMEGA_ON_DEVICE: u32 = 0x1;
pub const CORRLA_PCA_CENTER_FUSED: u32 = 0x2;
pub const CORRLA_PCA_CENTER_COPY: u32 = 0x4;
pub const CORRLA_QR_HOUSEHOLDER: u32 = 0x8;
pub const CORRLA_SEED_EXPLICIT: u32 = 0x10;
pub const CORRLA_POWER_FUSED: u32 = 0x20;
pub const CORRLA_SHARD_COLS: u32 = 0x40;
pub const CORRLA_SKETCH_BF16X3: u32 = 0x80; // opt-in: range finder on the bf16-split kernels (f32 inputs only)
pub const CORRLA_SKETCH_BF16X6: u32 = 0x100;

extern "C" {
    fn corrla_ctx_create(device: c_int, out: *mut *mut c_void) -> c_int;
    fn corrla_ctx_destroy(ctx: *mut c_void);
    fn corrla_last_error() -> *const c_char;
    fn corrla_rsvd_f64(ctx: *mut c_void, a: *const f64, m: i64, n: i64, rs: i64, cs: i64, rank: i64, n_iter: i64,
                       n_over: i64, opts: *const CorrlaOpts, u: *mut f64, ldu: i64, s: *mut f64, vt: *mut f64,
                       ldvt: i64) -> c_int;
    fn corrla_rsvd_f32(ctx: *mut c_void, a: *const f32, m: i64, n: i64, rs: i64, cs: i64, rank: i64, n_iter: i64,
                       n_over: i64, opts: *const CorrlaOpts, u: *mut f32, ldu: i64, s: *mut f32, vt: *mut f32,
                       ldvt: i64) -> c_int;
    fn corrla_power_iter_f64(ctx: *mut c_void, a: *const f64, m: i64, n: i64, rs: i64, cs: i64, width: i64,
                             n_iter: i64, opts: *const CorrlaOpts, q: *mut f64, ldq: i64) -> c_int;
    fn corrla_power_iter_f32(ctx: *mut c_void, a: *const f32, m: i64, n: i64, rs: i64, cs: i64, width: i64,
                             n_iter: i64, opts: *const CorrlaOpts, q: *mut f32, ldq: i64) -> c_int;
    fn corrla_pca_f64(ctx: *mut c_void, x: *const f64, m: i64, n: i64, rs: i64, cs: i64, rank: i64, n_iter: i64,
                      n_over: i64, opts: *const CorrlaOpts, means: *mut f64, s: *mut f64, comps: *mut f64,
                      ldc: i64) -> c_int;
    fn corrla_grad_mat_f64(ctx: *mut c_void, x: *const f64, n_pts: i64, k: i64, y: *const f64, xq: *const f64,
                           n_q: i64, est_order: c_int, n_nbrs: i64, out_scale: f64, g: *mut f64, ldg: i64,
                           n_regularised: *mut c_int) -> c_int;
}

/// One context per thread (device 0): replaces faer's process-global `Parallelism` (mat_utils.rs:31).
struct Ctx(*mut c_void);
impl Drop for Ctx {
    fn drop(&mut self) {
        unsafe { corrla_ctx_destroy(self.0) }
    }
}
thread_local! {
    static CTX: Ctx = unsafe {
        let mut c = std::ptr::null_mut();
        check(corrla_ctx_create(0, &mut c));
        Ctx(c)
    };
}

fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(corrla_last_error()) }.to_string_lossy().into_owned();
        panic!("corrla_rsvd (status {}): {}", rc, msg);
    }
}

/// Element types the library is built for (the reference is generic over `faer::RealField + Float`, instantiated
/// with f64 everywhere and f32 in tests).
pub trait RsvdScalar: faer::Entity + Copy + Default {
    unsafe fn rsvd(ctx: *mut c_void, a: *const Self, m: i64, n: i64, rs: i64, cs: i64, k: i64, q: i64, p: i64,
                   u: *mut Self, ldu: i64, s: *mut Self, vt: *mut Self, ldvt: i64) -> c_int;
    unsafe fn power(ctx: *mut c_void, a: *const Self, m: i64, n: i64, rs: i64, cs: i64, w: i64, q: i64,
                    out: *mut Self, ldq: i64) -> c_int;
}
impl RsvdScalar for f64 {
    unsafe fn rsvd(ctx: *mut c_void, a: *const f64, m: i64, n: i64, rs: i64, cs: i64, k: i64, q: i64, p: i64,
                   u: *mut f64, ldu: i64, s: *mut f64, vt: *mut f64, ldvt: i64) -> c_int {
        corrla_rsvd_f64(ctx, a, m, n, rs, cs, k, q, p, std::ptr::null(), u, ldu, s, vt, ldvt)
    }
    unsafe fn power(ctx: *mut c_void, a: *const f64, m: i64, n: i64, rs: i64, cs: i64, w: i64, q: i64,
                    out: *mut f64, ldq: i64) -> c_int {
        corrla_power_iter_f64(ctx, a, m, n, rs, cs, w, q, std::ptr::null(), out, ldq)
    }
}
impl RsvdScalar for f32 {
    unsafe fn rsvd(ctx: *mut c_void, a: *const f32, m: i64, n: i64, rs: i64, cs: i64, k: i64, q: i64, p: i64,
                   u: *mut f32, ldu: i64, s: *mut f32, vt: *mut f32, ldvt: i64) -> c_int {
        corrla_rsvd_f32(ctx, a, m, n, rs, cs, k, q, p, std::ptr::null(), u, ldu, s, vt, ldvt)
    }
    unsafe fn power(ctx: *mut c_void, a: *const f32, m: i64, n: i64, rs: i64, cs: i64, w: i64, q: i64,
                    out: *mut f32, ldq: i64) -> c_int {
        corrla_power_iter_f32(ctx, a, m, n, rs, cs, w, q, std::ptr::null(), out, ldq)
    }
}

/// random_svd.rs:63-110 -- same name, argument order and result shapes: (U m x k, S k x 1, Vt k x n).
pub fn random_svd<T: RsvdScalar>(a_mat: MatRef<T>, omega_rank: usize, n_iter: usize, n_oversamples: usize)
    -> (Mat<T>, Mat<T>, Mat<T>)
{
    let (m, n, k) = (a_mat.nrows(), a_mat.ncols(), omega_rank);
    let mut u = Mat::<T>::zeros(m, k);
    let mut s = Mat::<T>::zeros(k, 1); // k x 1, as s_diagonal().as_2d() at random_svd.rs:99,106
    let mut vt = Mat::<T>::zeros(k, n);
    let rc = CTX.with(|c| unsafe {
        T::rsvd(c.0, a_mat.as_ptr(), m as i64, n as i64, a_mat.row_stride() as i64, a_mat.col_stride() as i64,
                k as i64, n_iter as i64, n_oversamples as i64, u.as_ptr_mut(), u.col_stride() as i64,
                s.as_ptr_mut(), vt.as_ptr_mut(), vt.col_stride() as i64)
    });
    check(rc);
    (u, s, vt)
}

/// random_svd.rs:15-59 -- `omega_rank` is the already-oversampled sketch width; returns the orthonormal Q (m x width).
pub fn power_iter<T: RsvdScalar>(a_mat: MatRef<T>, omega_rank: usize, n_iter: usize) -> Mat<T> {
    let (m, n) = (a_mat.nrows(), a_mat.ncols());
    let mut q = Mat::<T>::zeros(m, omega_rank);
    let rc = CTX.with(|c| unsafe {
        T::power(c.0, a_mat.as_ptr(), m as i64, n as i64, a_mat.row_stride() as i64, a_mat.col_stride() as i64,
                 omega_rank as i64, n_iter as i64, q.as_ptr_mut(), q.col_stride() as i64)
    });
    check(rc);
    q
}

/// The fit of `PcaRsvd::new(x_mat, rank)` (pca_rsvd.rs:56-82) in one call: column means, centring and
/// `random_svd(cx, rank, 20, min(n_dim, 10))`.  Returns (means 1 x n_dim, singular values k x 1, components k x n_dim).
pub fn pca_rsvd_fit(x_mat: MatRef<f64>, rank: usize) -> (Mat<f64>, Mat<f64>, Mat<f64>) {
    let (m, n) = (x_mat.nrows(), x_mat.ncols());
    let mut means = Mat::<f64>::zeros(n, 1); // n contiguous values; read as the reference's 1 x n row
    let mut s = Mat::<f64>::zeros(rank, 1);
    let mut comps = Mat::<f64>::zeros(rank, n);
    let rc = CTX.with(|c| unsafe {
        corrla_pca_f64(c.0, x_mat.as_ptr(), m as i64, n as i64, x_mat.row_stride() as i64, x_mat.col_stride() as i64,
                       rank as i64, 20, std::cmp::min(n, 10) as i64, std::ptr::null(), means.as_ptr_mut(),
                       s.as_ptr_mut(), comps.as_ptr_mut(), comps.col_stride() as i64)
    });
    check(rc);
    (means.transpose().to_owned(), s, comps)
}

/// `ActiveSsRsvd::create_grad_mat` with a `PolyGradientEstimator(x, y, est_order, n_nbrs)`
/// (active_subspaces.rs:66-141, 215-229).  `x` and `xq` must be row-major n x k (contiguous rows); returns the
/// column-major k x n_q gradient matrix and the number of queries whose fit needed the ridge.
pub fn create_grad_mat(x: MatRef<f64>, y: &[f64], xq: MatRef<f64>, est_order: i32, n_nbrs: usize) -> (Mat<f64>, i32) {
    assert!(x.col_stride() == 1 && xq.col_stride() == 1, "row-major point sets expected");
    assert!(x.row_stride() as usize == x.ncols() && xq.row_stride() as usize == xq.ncols());
    let (n, k, nq) = (x.nrows(), x.ncols(), xq.nrows());
    assert_eq!(y.len(), n);
    let mut g = Mat::<f64>::zeros(k, nq);
    let mut nreg: c_int = 0;
    let rc = CTX.with(|c| unsafe {
        corrla_grad_mat_f64(c.0, x.as_ptr(), n as i64, k as i64, y.as_ptr(), xq.as_ptr(), nq as i64, est_order,
                            n_nbrs as i64, 1.0, g.as_ptr_mut(), g.col_stride() as i64, &mut nreg)
    });
    check(rc);
    (g, nreg)
}

#[cfg(test)]
mod tests {
    use super::*;
    use faer::mat;

    /// test_rsvd_lowrank, random_svd.rs:153-196: S = (3, 2.2360679, 2, 0, 0) to 1e-3
    #[test]
    fn known_answer_5x5() {
        let a = mat![
            [1.0, 0.0, 0.0, 0.0, 2.0],
            [0.0, 0.0, 3.0, 0.0, 0.0],
            [0.0, 0.0, 0.0, 0.0, 0.0],
            [0.0, 0.0, 0.0, 0.0, 0.0],
            [0.0, 2.0, 0.0, 0.0, 0.0f64]
        ];
        let (_u, s, _vt) = random_svd(a.as_ref(), 5, 12, 10);
        let want = [3.0, 2.2360679, 2.0, 0.0, 0.0];
        for i in 0..5 {
            assert!((s.read(i, 0) - want[i]).abs() < 1e-3);
        }
    }
}
