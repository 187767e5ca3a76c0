"""Host-side: partial fractions of x^(-alpha) for the RHMC path, fitted by the library through its C export lqcd_rational_fit
(csrc/rational.hip; latticeqcd.jl_amd/rational.py is the ctypes stub) and checked against an independent numpy / scipy fit
(tests/rational_scipy.py).  No GPU needed: the fit is host arithmetic."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("alpha", [0.5, 0.25, 0.125, 0.75])
@pytest.mark.parametrize("interval", [(0.25, 16.25), (0.0025, 16.0025), (1e-4, 4.6)])
def test_partial_fractions_accuracy_and_signs(lq, alpha, interval):
    tol = 1e-8 if interval[1] / interval[0] > 1e4 else 1e-9        # double precision: condition 5e4 costs a digit
    a0, res, poles, err = lq.rational.inverse_power_partial_fractions(alpha, *interval, tol=tol)
    assert err < tol and a0 >= 0 and (res > 0).all() and (poles > 0).all() and len(poles) <= 30
    x = np.exp(np.random.default_rng(1).uniform(np.log(interval[0]), np.log(interval[1]), 500))
    assert np.abs(lq.rational.evaluate(a0, res, poles, x) * x ** alpha - 1.0).max() < tol
    with pytest.raises(ValueError):
        lq.rational.inverse_power_partial_fractions(1.5, *interval)


# the exponents and intervals of the reference's general-Nf staggered runs (test/test_Nf2.toml:8, test/test_Nf3.toml:8: mass 0.5 -> [0.25, 16.25],
# alpha = Nf/8 for the action, 1 - Nf/16 for the heat bath) and of a Wilson Nf = 1 action
@pytest.mark.parametrize("alpha,lo,hi,tol", [(2 / 8, 0.25, 16.25, 1e-10), (3 / 8, 0.25, 16.25, 1e-12), (1 - 2 / 16, 0.25, 16.25, 1e-12),
                                             (1 - 3 / 16, 0.25, 16.25, 1e-12), (0.5, 0.01, 3.0, 1e-10), (0.75, 0.01, 3.0, 1e-12),
                                             (2 / 8, 0.0025, 16.0025, 1e-10)])
def test_c_export_against_independent_scipy_fit(lq, alpha, lo, hi, tol):
    import rational_scipy as ref
    L = lq.lib.lib()
    a0, n, err = C.c_double(0), C.c_int(0), C.c_double(0)
    res, poles = (C.c_double * 40)(), (C.c_double * 40)()
    st = L.lqcd_rational_fit(C.c_double(alpha), C.c_double(lo), C.c_double(hi), C.c_double(tol), 40, C.byref(a0), res, poles, C.byref(n), C.byref(err))
    assert st == 0, L.lqcd_last_error()
    res, poles = np.array(res[:n.value]), np.array(poles[:n.value])
    assert a0.value >= 0 and (res > 0).all() and (poles > 0).all() and (np.diff(poles) > 0).all()          # signs, ordering
    x = np.exp(np.linspace(np.log(lo), np.log(hi), 20011))                                                     # neither the fit grid nor the library's verification grid
    mine = a0.value + (res[None, :] / (x[:, None] + poles[None, :])).sum(axis=1)
    assert np.abs(mine * x ** alpha - 1.0).max() <= tol and err.value <= tol
    b0, bres, bpoles, berr = ref.inverse_power_partial_fractions(alpha, lo, hi, tol)
    # two fits of the same function to the same accuracy: same pole count (+-1), and they agree with each other to the sum of their errors
    assert abs(len(bpoles) - n.value) <= 1
    assert np.abs(mine / ref.evaluate(b0, bres, bpoles, x) - 1.0).max() <= err.value + berr + 1e-15


def test_c_export_argument_errors_and_unreachable_accuracy(lq):
    L = lq.lib.lib()
    a0, n = C.c_double(0), C.c_int(0)
    res, poles = (C.c_double * 40)(), (C.c_double * 40)()

    def fit(alpha, lo, hi, tol, cap=40):
        return L.lqcd_rational_fit(C.c_double(alpha), C.c_double(lo), C.c_double(hi), C.c_double(tol), cap, C.byref(a0), res, poles, C.byref(n), None)
    assert fit(1.5, 0.25, 16.25, 1e-10) == lq.lib.ERR_ARG and fit(0.5, 2.0, 1.0, 1e-10) == lq.lib.ERR_ARG and fit(0.5, 0.0, 1.0, 1e-10) == lq.lib.ERR_ARG
    assert fit(0.5, 1e-8, 16.0, 1e-15) == lq.lib.ERR_NOT_CONVERGED and b"rational fit" in L.lqcd_last_error()
    assert fit(0.5, 0.25, 16.25, 1e-10, cap=3) == lq.lib.ERR_NOT_CONVERGED          # three poles cannot reach 1e-10
    assert fit(0.5, 0.25, 16.25, 1e-10) == 0 and 6 <= n.value <= 14


# the host half of the Lanczos certificate behind the Wilson rational action's fit interval (csrc/rational.hip lanczos_certified): Ritz value and
# |last eigenvector component| of the Lanczos tridiagonal against LAPACK, and the residual bound |beta_k s_k| against the true spectrum of a dense matrix
@pytest.mark.parametrize("n", [1, 2, 3, 17, 60, 200])
def test_tridiag_ritz_pair_against_dense_eigensolver(lq, n):
    rng = np.random.default_rng(n)
    d, e = rng.uniform(0.5, 3.0, n), rng.uniform(0.1, 1.0, max(n - 1, 0)) * rng.choice([-1.0, 1.0], max(n - 1, 0))
    T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
    w, V = np.linalg.eigh(T)
    for index in sorted({0, n // 2, n - 1}):
        theta, last = lq.rational.tridiag_ritz(d, e, index)
        assert abs(theta - w[index]) <= 1e-12 * np.abs(w).max()
        gap = min([abs(w[index] - w[j]) for j in range(n) if j != index] or [1.0])
        assert abs(last - abs(V[-1, index])) <= 1e-9 / min(gap, 1.0) + 1e-13
    with pytest.raises(lq.LQCDError):
        lq.rational.tridiag_ritz(d, e, n)


@pytest.mark.parametrize("cond", [1e2, 1e5])
def test_ritz_bound_contains_an_eigenvalue(lq, cond):
    rng = np.random.default_rng(5)
    N = 300
    lam = np.exp(rng.uniform(np.log(1.0 / cond), 0.0, N))          # an ill-conditioned positive spectrum, as D'D near the critical kappa
    Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
    A = (Q * lam) @ Q.T
    v = rng.standard_normal(N)
    v /= np.linalg.norm(v)
    vp, beta, al, be = np.zeros(N), 0.0, [], []
    for k in range(1, 61):
        w = A @ v
        a = v @ w
        w -= a * v + beta * vp
        al.append(a)
        beta = np.linalg.norm(w)
        if k % 10 == 0:
            for index in (0, k - 1):
                theta, last = lq.rational.tridiag_ritz(al, be, index)
                dist = np.abs(lam - theta).min()
                assert dist <= beta * last * (1 + 1e-6) + 1e-12, (k, index, dist, beta * last)
            assert theta <= lam.max() * (1 + 1e-12) and lq.rational.tridiag_ritz(al, be, 0)[0] >= lam.min() * (1 - 1e-12)
        be.append(beta)
        vp, v = v, w / beta
