"""Oracle-side checks of the rational (RHMC) staggered action, the one behind the reference's Nf = 2 and Nf = 3 runs
(test/test_Nf2.toml:8, test/test_Nf3.toml:8): the partial-fraction action against the exact spectral function of the dense
D^+D on a small lattice, the heat-bath identity, and the force against finite differences of the exact action."""
import numpy as np
import pytest

L = (4, 2, 2, 2)
MASS = 0.5
BC = (1, 1, 1, -1)


@pytest.fixture(scope="module")
def setup(orc):
    U = orc.hot_gauge(L, 811)
    A = orc.dense_DdagD(orc.STAGGERED, U, L, MASS, 1.0, BC)
    assert np.abs(A - A.conj().T).max() < 1e-13
    w, V = np.linalg.eigh(A)
    return U, A, w, V


def exact_power(w, V, v, p):
    flat = v.reshape(-1)
    return (V @ ((w ** p) * (V.conj().T @ flat))).reshape(v.shape)


def test_spectrum_lies_in_the_fit_interval(setup):
    _, _, w, _ = setup
    assert w.min() >= MASS ** 2 - 1e-12 and w.max() <= MASS ** 2 + 16.0


@pytest.mark.parametrize("nf", [2, 3])
def test_rational_action_and_heat_bath_match_the_exact_spectral_function(lq, orc, setup, nf):
    U, A, w, V = setup
    lo, hi = MASS ** 2, MASS ** 2 + 16.0
    phi = orc.gaussian_spinor(orc.staggered_shape(L), 812)
    a0, res, poles, err = lq.rational.inverse_power_partial_fractions(nf / 8.0, lo, hi, 1e-12)
    y, _ = orc.rational_apply(orc.STAGGERED, U, phi, L, MASS, a0, res, poles, 1.0, BC)
    ex = exact_power(w, V, phi, -nf / 8.0)
    assert np.abs(y - ex).max() < 1e-10 * np.abs(ex).max()
    # heat bath: phi = (D^+D)^(Nf/16) xi = D^+D r(D^+D) xi with the fit of x^(Nf/16 - 1); then S_f(phi) = xi^+ xi
    b0, bres, bpoles, _ = lq.rational.inverse_power_partial_fractions(1.0 - nf / 16.0, lo, hi, 1e-12)
    xi = orc.gaussian_spinor(orc.staggered_shape(L), 813)
    t, _ = orc.rational_apply(orc.STAGGERED, U, xi, L, MASS, b0, bres, bpoles, 1.0, BC)
    hb = orc.staggered_D(U, orc.staggered_D(U, t, L, MASS, BC), L, MASS, BC, True)
    assert np.abs(hb - exact_power(w, V, xi, nf / 16.0)).max() < 1e-10 * np.abs(hb).max()
    S, _ = orc.rational_apply(orc.STAGGERED, U, hb, L, MASS, a0, res, poles, 1.0, BC)
    assert abs(np.vdot(hb, S).real / np.vdot(xi, xi).real - 1.0) < 1e-10


def test_rational_force_is_the_derivative_of_the_exact_action(lq, orc, setup):
    U, _, _, _ = setup
    nf = 3
    lo, hi = MASS ** 2, MASS ** 2 + 16.0
    phi = orc.gaussian_spinor(orc.staggered_shape(L), 814)
    a0, res, poles, _ = lq.rational.inverse_power_partial_fractions(nf / 8.0, lo, hi, 1e-11)
    G = orc.rational_force(orc.STAGGERED, U, phi, L, MASS, res, poles, 1.0, BC)
    rng = np.random.default_rng(815)

    def action(Ut):
        w, V = np.linalg.eigh(orc.dense_DdagD(orc.STAGGERED, Ut, L, MASS, 1.0, BC))
        return np.vdot(phi, exact_power(w, V, phi, -nf / 8.0)).real

    for _ in range(3):
        mu, t, z, y, x = rng.integers(4), rng.integers(L[3]), rng.integers(L[2]), rng.integers(L[1]), rng.integers(L[0])
        T = sum(c * g for c, g in zip(rng.normal(size=8), orc.GELLMANN))
        eps = 1e-4
        fd = orc_fd(orc, action, U, (mu, t, z, y, x), T, eps)
        an = -2.0 * np.trace(T @ orc_mat(G[mu, t, z, y, x])).imag
        assert abs(fd - an) < 2e-6 * max(1.0, abs(an)), (fd, an)


def orc_mat(m):
    return m.T          # same transposition as tests/test_oracle_identities.py


def orc_fd(orc, action, U, site, T, eps):
    wv, Vv = np.linalg.eigh(T)
    out = []
    for sgn in (+1, -1):
        E = (Vv * np.exp(1j * sgn * eps * wv)) @ Vv.conj().T
        Ut = U.copy()
        Ut[site] = (E @ U[site].T).T
        out.append(action(Ut))
    return (out[0] - out[1]) / (2 * eps)


def test_wilson_one_flavour_rational_action_and_force(lq, orc):
    """Wilson Nf = 1: S_f = phi^+ (D^+D)^(-1/2) phi.  Partial fractions on the exact spectral interval of the dense D^+D (768 x 768),
    action and heat bath against the exact spectral function, force against central differences of the exact action."""
    from scipy.linalg import expm
    Lw, kappa = (4, 2, 2, 2), 0.141139
    U = orc.hot_gauge(Lw, 841)
    A = orc.dense_DdagD(orc.WILSON, U, Lw, kappa, 1.0, BC)
    w, V = np.linalg.eigh(A)
    lo, hi = 0.8 * w.min(), 1.2 * w.max()
    phi = orc.gaussian_spinor(orc.wilson_shape(Lw), 842)
    a0, res, poles, _ = lq.rational.inverse_power_partial_fractions(0.5, lo, hi, 1e-11)
    y, _ = orc.rational_apply(orc.WILSON, U, phi, Lw, kappa, a0, res, poles, 1.0, BC)
    ex = exact_power(w, V, phi, -0.5)
    assert np.abs(y - ex).max() < 1e-9 * np.abs(ex).max()
    b0, bres, bpoles, _ = lq.rational.inverse_power_partial_fractions(0.75, lo, hi, 1e-11)       # x^(1/4) = x * x^(-3/4)
    xi = orc.gaussian_spinor(orc.wilson_shape(Lw), 843)
    t, _ = orc.rational_apply(orc.WILSON, U, xi, Lw, kappa, b0, bres, bpoles, 1.0, BC)
    hb = orc.wilson_D(U, orc.wilson_D(U, t, Lw, kappa, 1.0, BC), Lw, kappa, 1.0, BC, True)
    S, _ = orc.rational_apply(orc.WILSON, U, hb, Lw, kappa, a0, res, poles, 1.0, BC)
    assert abs(np.vdot(hb, S).real / np.vdot(xi, xi).real - 1.0) < 1e-9
    G = orc.rational_force(orc.WILSON, U, phi, Lw, kappa, res, poles, 1.0, BC)

    def action(Ut):
        wt, Vt = np.linalg.eigh(orc.dense_DdagD(orc.WILSON, Ut, Lw, kappa, 1.0, BC))
        return np.vdot(phi, exact_power(wt, Vt, phi, -0.5)).real

    rng = np.random.default_rng(844)
    for site in [(0, 1, 1, 0, 3), (3, 1, 0, 1, 2)]:                  # (mu, t, z, y, x); the second one crosses the t boundary
        T = sum(c * g for c, g in zip(rng.normal(size=8), orc.GELLMANN))
        fd = orc_fd(orc, action, U, site, T, 1e-4)
        an = -2.0 * np.trace(T @ orc_mat(G[site])).imag
        assert abs(fd - an) < 2e-6 * max(1.0, abs(an)), (site, fd, an)
