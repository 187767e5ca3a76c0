"""Partial-fraction approximations of x^(-alpha), 0 < alpha < 1, on a spectral interval -- the coefficients the RHMC path feeds to
the multi-shift solver `shiftedcg` (the reference's general-Nf staggered runs, README.md:112,132,
test/test_Nf2.toml:8, test/test_Nf3.toml:8, obtain theirs from its rational-HMC package).

    x^(-alpha)  ~=  a0 + sum_k  r_k / (x + p_k),      a0 >= 0, r_k > 0,  p_k > 0            for  lam_min <= x <= lam_max

so that  (D'D)^(-alpha) phi ~= a0 phi + sum_k r_k (D'D + p_k)^(-1) phi  is ONE multi-shift solve.  The fit itself is computed by the
library (lqcd_rational_fit, csrc/rational.hip: AAA algorithm, no LAPACK / scipy) so that the Julia binding and this mirror get the
same coefficients through the same C export; host-only numerics, no GPU needed."""
import ctypes as C

import numpy as np

from . import lib as _l


def inverse_power_partial_fractions(alpha, lam_min, lam_max, tol=1e-10, max_poles=40):
    """Returns (a0, residues r_k, poles p_k, max relative error on [lam_min, lam_max]).  ValueError: arguments outside the domain;
    RuntimeError: the accuracy is not reachable in double precision on this interval."""
    if not (0.0 < alpha < 1.0):
        raise ValueError("alpha must lie in (0, 1)")
    if not (0.0 < lam_min < lam_max):
        raise ValueError("need 0 < lam_min < lam_max")
    a0, n, err = C.c_double(0), C.c_int(0), C.c_double(0)
    res, poles = (C.c_double * max_poles)(), (C.c_double * max_poles)()
    st = _l.lib().lqcd_rational_fit(C.c_double(alpha), C.c_double(lam_min), C.c_double(lam_max), C.c_double(tol), int(max_poles),
                                    C.byref(a0), res, poles, C.byref(n), C.byref(err))
    if st == _l.ERR_NOT_CONVERGED:
        raise RuntimeError(_l.lib().lqcd_last_error().decode("utf-8", "replace"))
    _l.check(st)
    return float(a0.value), np.array(res[:n.value]), np.array(poles[:n.value]), float(err.value)


def evaluate(a0, res, poles, x):
    x = np.asarray(x, dtype=np.float64)
    return (a0 + (np.asarray(res)[None, :] / (x.reshape(-1, 1) + np.asarray(poles)[None, :])).sum(axis=1)).reshape(x.shape)


def tridiag_ritz(diag, offdiag, index):
    """(theta, |last component of its normalised eigenvector|) for the index-th eigenvalue (ascending) of the symmetric tridiagonal -- the host
    half of the Lanczos certificate of the rational actions (lqcd_tridiag_ritz): beta_n * last_component bounds |theta - eigenvalue of D'D|."""
    d = np.ascontiguousarray(diag, dtype=np.float64)
    e = np.ascontiguousarray(offdiag, dtype=np.float64)
    if len(e) != max(len(d) - 1, 0):
        raise ValueError("offdiag must hold len(diag) - 1 entries")
    theta, last = C.c_double(0), C.c_double(0)
    _l.check(_l.lib().lqcd_tridiag_ritz(int(len(d)), d.ctypes.data_as(C.POINTER(C.c_double)), e.ctypes.data_as(C.POINTER(C.c_double)), int(index),
                                        C.byref(theta), C.byref(last)))
    return float(theta.value), float(last.value)
