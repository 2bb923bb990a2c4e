"""Shared comparison helpers for the parity tests."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    out = {k: d[k] for k in d.files}
    for k in ("k", "q", "p"):
        out[k] = int(out[k])
    return out


def align_signs(u, vt, u_ref, vt_ref):
    """Singular vectors are defined up to a common sign of (u_i, v_i);
    flip ours to match the reference (the reference fixes no convention)."""
    u = np.array(u, dtype=np.float64, copy=True)
    vt = np.array(vt, dtype=np.float64, copy=True)
    for i in range(u.shape[1]):
        d = float(np.dot(u[:, i], u_ref[:, i])) + float(np.dot(vt[i, :], vt_ref[i, :]))
        if d < 0:
            u[:, i] *= -1
            vt[i, :] *= -1
    return u, vt


def check_factorization(a, u, s, vt, k, tol_orth):
    m0, n0 = a.shape
    assert u.shape == (m0, k)
    assert s.shape == (k, 1)
    assert vt.shape == (k, n0)
    sv = s.ravel()
    assert np.all(sv >= 0)
    assert np.all(np.diff(sv) <= 1e-6 * max(sv[0], 1e-300)), "S must be descending"


def orth_err(u):
    u = np.asarray(u, np.float64)
    return float(np.max(np.abs(u.T @ u - np.eye(u.shape[1]))))
