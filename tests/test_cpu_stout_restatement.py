"""The numpy restatement of stout smearing (plaquette staples) and of its back-propagation, checked without reference to itself: the smeared links are
unitary and raise the plaquette, the staple sum agrees with the C oracle's gauge force, and the back-propagated force is the finite-difference
derivative of a test action of the smeared links."""
import numpy as np
import scipy.linalg as sla

L = (4, 4, 2, 4)


def _perturb(U, idx, T, e):
    V = U.copy()
    V[idx] = V[idx] @ sla.expm(1j * e * T).T        # host image [b, a] is the transpose of the matrix
    return V


def test_smearing_is_unitary_smooths_and_uses_the_gauge_forces_staples(orc):
    U = orc.hot_gauge(L, 3)
    W = orc._mat(orc.gauge_force(U, L, -6.0))                        # G = -(beta/6) U A: beta = -6 gives U A
    for mu in range(4):
        assert np.abs(orc._mat(U)[mu] @ orc._staple_sum(orc._mat(U), L, mu) - W[mu]).max() < 1e-12
    Us = orc.stout_smear(U, L, 0.1)
    assert orc.unitarity_dev(Us, L) < 1e-13
    assert orc.plaquette(Us, L) > orc.plaquette(U, L) + 0.05
    assert np.abs(orc.stout_smear(U, L, 0.0) - U).max() < 1e-15


def test_backprop_is_the_derivative_through_the_smearing(orc):
    rho = 0.12
    U = orc.hot_gauge(L, 4)
    rng = np.random.default_rng(5)
    Kh = rng.standard_normal(orc.gauge_shape(L)) + 1j * rng.standard_normal(orc.gauge_shape(L))      # S(U') = sum Re tr(K U'), K[.., a, b] = Kh[.., b, a]

    def S(V):
        return float(np.sum(orc._mat(Kh) * np.swapaxes(orc._mat(orc.stout_smear(V, L, rho)), -1, -2)).real)

    Us = orc.stout_smear(U, L, rho)
    Gs = np.ascontiguousarray(orc._mat(0.5 * orc._mat(Us) @ orc._mat(Kh)))       # Re tr(K i T U') = -Im tr(T U' K) = -2 Im tr(T G'), G' = U' K / 2
    G = orc._mat(orc.stout_backprop(Gs, U, L, rho))
    for _ in range(4):
        idx = tuple(int(rng.integers(n)) for n in (4, L[3], L[2], L[1], L[0]))
        T = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
        T = T + T.conj().T
        h = 1e-5
        fd = (S(_perturb(U, idx, T, h)) - S(_perturb(U, idx, T, -h))) / (2 * h)
        an = -2.0 * np.imag(np.trace(T @ G[idx]))
        assert abs(fd - an) < 1e-6 * max(1.0, abs(fd)), (fd, an)
    # rho = 0: the smearing is the identity and so is its back-propagation
    assert np.abs(orc.stout_backprop(Gs, U, L, 0.0) - Gs).max() < 1e-14


def test_polyakov_loop_restatement(orc):
    """oracle.polyakov_loop: cold links give 1; the reference's 4^4 Wilson fixture gives a small complex number; a centre transformation of one time slice multiplies
    the loop by exp(2 pi i / 3) and leaves the plaquette alone; a gauge transformation leaves it alone."""
    import os
    from conftest import GOLDEN
    import latticeqcd_jl_amd as lq
    Lc = (4, 4, 2, 6)
    assert abs(orc.polyakov_loop(orc.unit_gauge(Lc), Lc) - 1.0) < 1e-15
    U = orc.hot_gauge(Lc, 12)
    p = orc.polyakov_loop(U, Lc)
    z = np.exp(2j * np.pi / 3)
    Uz = U.copy()
    Uz[3, 2] *= z
    assert abs(orc.polyakov_loop(Uz, Lc) - z * p) < 1e-14 and abs(orc.plaquette(Uz, Lc) - orc.plaquette(U, Lc)) < 1e-14
    # gauge transformation U_mu(n) -> g(n) U_mu(n) g(n + mu)^+ (host image [b, a] = the transpose)
    rng = np.random.default_rng(13)
    g = orc.random_su3(rng, Lc[0] * Lc[1] * Lc[2] * Lc[3]).reshape(Lc[3], Lc[2], Lc[1], Lc[0], 3, 3)
    Um = orc._mat(U)
    Ug = np.empty_like(Um)
    for mu in range(4):
        Ug[mu] = g @ Um[mu] @ orc._dag(orc._sh(g, Lc, mu, 1))
    Ugh = np.ascontiguousarray(orc._mat(Ug))
    assert abs(orc.polyakov_loop(Ugh, Lc) - p) < 1e-13 and abs(orc.plaquette(Ugh, Lc) - orc.plaquette(U, Lc)) < 1e-13
    L4 = (4, 4, 4, 4)
    Uf = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L4)
    assert abs(orc.polyakov_loop(Uf, L4)) < 0.3


def test_stout_smearing_is_gauge_covariant(orc):
    """smear(g U g^+) = g smear(U) g^+ : the smeared links transform like links (what makes the smeared fermion action gauge invariant)."""
    Lc = (4, 4, 2, 4)
    U = orc.hot_gauge(Lc, 21)
    rng = np.random.default_rng(22)
    g = orc.random_su3(rng, Lc[0] * Lc[1] * Lc[2] * Lc[3]).reshape(Lc[3], Lc[2], Lc[1], Lc[0], 3, 3)

    def transform(V):
        Vm = orc._mat(V)
        out = np.empty_like(Vm)
        for mu in range(4):
            out[mu] = g @ Vm[mu] @ orc._dag(orc._sh(g, Lc, mu, 1))
        return np.ascontiguousarray(orc._mat(out))

    assert np.abs(orc.stout_smear(transform(U), Lc, 0.13) - transform(orc.stout_smear(U, Lc, 0.13))).max() < 1e-13
