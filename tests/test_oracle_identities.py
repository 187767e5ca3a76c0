"""The oracle is parity-unpinned at the Dslash/CG level (the reference's arithmetic lives in un-vendored Julia
packages); these convention-independent identities are its known-answer tests.  The same identities
are run against the HIP path in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import rel_err

L4 = (4, 4, 4, 4)
LA = (4, 6, 2, 8)  # anisotropic: catches any x/y/z/t mix-up
KAPPA, MASS = 0.141139, 0.5
BC = (1, 1, 1, -1)


def test_gamma_algebra(orc):
    g = orc.GAMMA
    for mu in range(4):
        assert np.allclose(g[mu], g[mu].conj().T)
        for nu in range(4):
            acomm = g[mu] @ g[nu] + g[nu] @ g[mu]
            assert np.allclose(acomm, 2 * np.eye(4) * (mu == nu))
    assert np.allclose(g[4], g[0] @ g[1] @ g[2] @ g[3])


@pytest.mark.parametrize("L", [L4, LA])
@pytest.mark.parametrize("r", [1.0, 0.7])
def test_wilson_adjoint_and_gamma5_hermiticity(orc, L, r):
    U = orc.hot_gauge(L, 1)
    a = orc.gaussian_spinor(orc.wilson_shape(L), 2)
    b = orc.gaussian_spinor(orc.wilson_shape(L), 3)
    Db = orc.wilson_D(U, b, L, KAPPA, r, BC, dagger=False)
    Dda = orc.wilson_D(U, a, L, KAPPA, r, BC, dagger=True)   # oracle: literally gamma5 D gamma5
    lhs = np.vdot(a, Db)
    rhs = np.conj(np.vdot(b, Dda))
    assert abs(lhs - rhs) < 1e-10 * abs(lhs)


def plane_wave(L, k, bc):
    """phase[t,z,y,x] = exp(i p.n), p_mu = 2 pi (k_mu + (bc_mu == -1)/2) / L_mu"""
    p = [2 * np.pi * (k[mu] + (0.5 if bc[mu] == -1 else 0.0)) / L[mu] for mu in range(4)]
    x, y, z, t = np.meshgrid(*[np.arange(L[mu]) for mu in range(4)], indexing="ij")
    ph = np.exp(1j * (p[0] * x + p[1] * y + p[2] * z + p[3] * t))  # [x,y,z,t]
    return np.ascontiguousarray(ph.transpose(3, 2, 1, 0)), p


@pytest.mark.parametrize("k", [(0, 0, 0, 0), (1, 0, 2, 1), (3, 2, 1, 3)])
def test_wilson_free_field_plane_wave(orc, k):
    L, r = LA, 1.0
    U = orc.unit_gauge(L)
    ph, p = plane_wave(L, k, BC)
    rng = np.random.default_rng(5)
    u = rng.standard_normal((4, 3)) + 1j * rng.standard_normal((4, 3))
    psi = np.ascontiguousarray(u[:, None, None, None, None, :] * ph[None, ..., None])
    out = orc.wilson_D(U, psi, L, KAPPA, r, BC)
    M = np.eye(4, dtype=complex)
    for mu in range(4):
        M = M - 2 * KAPPA * (r * np.cos(p[mu]) * np.eye(4) - 1j * np.sin(p[mu]) * orc.GAMMA[mu])
    expect = np.einsum("st,tc->sc", M, u)[:, None, None, None, None, :] * ph[None, ..., None]
    assert rel_err(out, expect) < 1e-13


def gauge_transform(U, g, L):
    """U^g_mu(n) = g(n) U_mu(n) g(n+mu)^+ ; arrays U[mu,t,z,y,x,b,a] (matrix element [a,b] = U[..., b, a])."""
    Ug = np.empty_like(U)
    axis = {0: 3, 1: 2, 2: 1, 3: 0}
    for mu in range(4):
        M = np.swapaxes(U[mu], -1, -2)                      # [t,z,y,x,a,b]
        gs = np.roll(g, -1, axis=axis[mu])                  # g(n+mu), periodic links
        Mg = np.einsum("...ab,...bc,...dc->...ad", g, M, gs.conj())
        Ug[mu] = np.swapaxes(Mg, -1, -2)
    return np.ascontiguousarray(Ug)


def test_wilson_gauge_covariance(orc):
    L = LA
    U = orc.hot_gauge(L, 7)
    rng = np.random.default_rng(8)
    V = L[0] * L[1] * L[2] * L[3]
    g = orc.random_su3(rng, V).reshape(L[3], L[2], L[1], L[0], 3, 3)    # g[t,z,y,x,a,b]
    psi = orc.gaussian_spinor(orc.wilson_shape(L), 9)
    psig = np.ascontiguousarray(np.einsum("...ab,s...b->s...a", g, psi))
    lhs = orc.wilson_D(gauge_transform(U, g, L), psig, L, KAPPA, 1.0, BC)
    rhs = np.einsum("...ab,s...b->s...a", g, orc.wilson_D(U, psi, L, KAPPA, 1.0, BC))
    assert rel_err(lhs, rhs) < 1e-13
    assert abs(orc.plaquette(gauge_transform(U, g, L), L) - orc.plaquette(U, L)) < 1e-13


def test_wilson_hop_parity_decomposition(orc):
    L = LA
    U = orc.hot_gauge(L, 11)
    psi = orc.gaussian_spinor(orc.wilson_shape(L), 12)
    for dag in (False, True):
        he = orc.wilson_hop_parity(U, psi, L, 1.0, BC, dag, 0)
        ho = orc.wilson_hop_parity(U, psi, L, 1.0, BC, dag, 1)
        full = orc.wilson_D(U, psi, L, KAPPA, 1.0, BC, dag)
        assert rel_err(psi - KAPPA * (he + ho), full) < 1e-13
    # even output sites only depend on odd input sites
    x, y, z, t = np.meshgrid(*[np.arange(L[mu]) for mu in range(4)], indexing="ij")
    par = ((x + y + z + t) & 1).transpose(3, 2, 1, 0)
    psi_e = psi * (par == 0)[None, ..., None]
    assert np.abs(orc.wilson_hop_parity(U, np.ascontiguousarray(psi_e), L, 1.0, BC, False, 0)).max() == 0.0


def test_staggered_identities(orc):
    L = LA
    U = orc.hot_gauge(L, 21)
    a = orc.gaussian_spinor(orc.staggered_shape(L), 22)
    b = orc.gaussian_spinor(orc.staggered_shape(L), 23)
    Hb = orc.staggered_D(U, b, L, 0.0, BC)
    Ha = orc.staggered_D(U, a, L, 0.0, BC)
    # D_hop^+ = -D_hop
    assert abs(np.vdot(a, Hb) + np.conj(np.vdot(b, Ha))) < 1e-10 * abs(np.vdot(a, Hb))
    # D^+ as implemented is the true adjoint
    Db = orc.staggered_D(U, b, L, MASS, BC)
    Dda = orc.staggered_D(U, a, L, MASS, BC, dagger=True)
    assert abs(np.vdot(a, Db) - np.conj(np.vdot(b, Dda))) < 1e-10 * abs(np.vdot(a, Db))
    # free field: D^+ D = m^2 + sum sin^2 p on plane waves
    U1 = orc.unit_gauge(L)
    ph, p = plane_wave(L, (1, 2, 0, 3), BC)
    psi = np.ascontiguousarray(ph[..., None] * np.array([1.0, 2.0 - 1j, 0.5j]))
    out = orc.staggered_D(U1, orc.staggered_D(U1, psi, L, MASS, BC), L, MASS, BC, dagger=True)
    lam = MASS ** 2 + sum(np.sin(p[mu]) ** 2 for mu in range(4))
    assert rel_err(out, lam * psi) < 1e-13


def dense_matrix(apply, shape):
    n = int(np.prod(shape))
    M = np.empty((n, n), dtype=np.complex128)
    e = np.zeros(n, dtype=np.complex128)
    for j in range(n):
        e[j] = 1.0
        M[:, j] = apply(e.reshape(shape)).reshape(-1)
        e[j] = 0.0
    return M


@pytest.fixture(scope="module")
def wilson_dense(orc):
    """Dense 384x384 Wilson matrix on a 2x2x2x4 hot lattice (brute-force anchor for the Krylov solvers)."""
    L = (2, 2, 2, 4)
    U = orc.hot_gauge(L, 31)
    M = dense_matrix(lambda v: orc.wilson_D(U, np.ascontiguousarray(v), L, KAPPA, 1.0, BC), orc.wilson_shape(L))
    return L, U, M


def test_wilson_dense_dagger(orc, wilson_dense):
    L, U, M = wilson_dense
    Md = dense_matrix(lambda v: orc.wilson_D(U, np.ascontiguousarray(v), L, KAPPA, 1.0, BC, dagger=True), orc.wilson_shape(L))
    assert np.abs(Md - M.conj().T).max() < 1e-14


def test_cg_vs_dense_solve(orc, wilson_dense):
    L, U, M = wilson_dense
    b = orc.gaussian_spinor(orc.wilson_shape(L), 32)
    x, it, rr, st = orc.cg_DdagD(orc.WILSON, U, b, L, KAPPA, 1.0, BC, eps=1e-22, maxiter=3000)
    assert st == 0 and rr < 1e-22
    xd = np.linalg.solve(M.conj().T @ M, b.reshape(-1)).reshape(b.shape)
    assert rel_err(x, xd) < 1e-10


@pytest.mark.parametrize("dagger", [False, True])
def test_bicgstab_and_eo_vs_dense_solve(orc, wilson_dense, dagger):
    L, U, M = wilson_dense
    b = orc.gaussian_spinor(orc.wilson_shape(L), 33)
    A = M.conj().T if dagger else M
    xd = np.linalg.solve(A, b.reshape(-1)).reshape(b.shape)
    x, it, rr, st = orc.bicgstab(orc.WILSON, U, b, L, KAPPA, 1.0, BC, dagger, eps=1e-22)
    assert st == 0 and rel_err(x, xd) < 1e-9
    xe, ite, rre, ste = orc.wilson_bicgstab_eo(U, b, L, KAPPA, 1.0, BC, dagger, eps=1e-22)
    assert ste == 0 and rel_err(xe, xd) < 1e-9
    assert ite <= it  # Schur preconditioning never needs more iterations here


def test_cg_on_reference_fixture(orc, lq):
    """CG on the reference's thermalised 4^4 Wilson configuration with the reference's parameters
    (test/test_wilson.toml: kappa = 0.141139, eps = 1e-19, BC = [1,1,1,-1])."""
    import os
    from conftest import GOLDEN
    U = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L4)
    b = orc.gaussian_spinor(orc.wilson_shape(L4), 41)
    x, it, rr, st = orc.cg_DdagD(orc.WILSON, U, b, L4, KAPPA, 1.0, BC, eps=1e-19, maxiter=3000)
    assert st == 0 and rr < 1e-19 and 10 < it < 500
    res = b - orc.wilson_D(U, orc.wilson_D(U, x, L4, KAPPA, 1.0, BC), L4, KAPPA, 1.0, BC, dagger=True)
    assert np.vdot(res, res).real < 1e-18


def test_staggered_cg(orc):
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 51)
    b = orc.gaussian_spinor(orc.staggered_shape(L), 52)
    x, it, rr, st = orc.cg_DdagD(orc.STAGGERED, U, b, L, MASS, 1.0, BC, eps=1e-10)
    assert st == 0 and rr < 1e-10
    res = b - orc.staggered_D(U, orc.staggered_D(U, x, L, MASS, BC), L, MASS, BC, dagger=True)
    assert np.vdot(res, res).real < 2e-10


@pytest.mark.parametrize("kind_name", ["wilson", "staggered"])
def test_multishift_cg_true_residuals(orc, kind_name):
    """Multi-shift CG (RHMC solver): every shifted system is solved from one Krylov space."""
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 61)
    kind, km, shape = (orc.WILSON, KAPPA, orc.wilson_shape(L)) if kind_name == "wilson" else (orc.STAGGERED, MASS, orc.staggered_shape(L))
    b = orc.gaussian_spinor(shape, 62)
    sig = [0.0, 0.003, 0.05, 0.7, 3.0]
    x0, xs, it, resid, st = orc.multishift_cg(kind, U, b, L, km, sig, eps=1e-20)
    assert st == 0 and resid < 1e-20

    def A(v):
        return orc.apply_D(kind, U, orc.apply_D(kind, U, np.ascontiguousarray(v), L, km), L, km, dagger=True)

    assert np.vdot(A(x0) - b, A(x0) - b).real < 2e-20
    for s, x in zip(sig, xs):
        res = A(x) + s * x - b
        assert np.vdot(res, res).real < 2e-20
    assert rel_err(xs[0], x0) < 1e-12                      # sigma = 0 reproduces the base solve
    xcg, itcg, rrcg, stcg = orc.cg_DdagD(kind, U, b, L, km, eps=1e-20)
    assert rel_err(x0, xcg) < 1e-9 and abs(it - itcg) <= 1


def test_multishift_cg_survives_zeta_underflow(orc):
    """A large shift converges within a few iterations and its zeta keeps shrinking geometrically: after a few hundred iterations of the
    base system it underflows; the recurrences must freeze that shift instead of producing 0/0 (seen at 48^3x96 with 18 poles)."""
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 63)
    b = orc.gaussian_spinor(orc.staggered_shape(L), 64)
    sig = [0.0, 1e-4, 30.0, 3000.0]
    x0, xs, it, resid, st = orc.multishift_cg(orc.STAGGERED, U, b, L, 0.01, sig, eps=1e-22)
    assert st == 0 and it > 120 and all(np.isfinite(x).all() for x in xs) and np.isfinite(x0).all()
    for s, x in zip(sig, xs):
        res = orc.staggered_D(U, orc.staggered_D(U, x, L, 0.01), L, 0.01, dagger=True) + s * x - b
        assert np.vdot(res, res).real < 1e-20, s


def _random_hermitian(rng):
    m = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    return 0.5 * (m + m.conj().T)


@pytest.mark.parametrize("kind_name", ["wilson", "staggered"])
@pytest.mark.parametrize("bc", [(1, 1, 1, -1), (1, 1, 1, 1)])
def test_fermion_force_is_the_derivative_of_the_action(orc, kind_name, bc):
    """The force field is defined by dS_f/d eps [U_mu(n) -> exp(i eps T) U_mu(n)] = -2 Im tr(T G_mu(n)); check it against
    central differences of S_f = eta^+ (D^+D)^-1 eta itself (no convention enters), including links that cross the boundary."""
    from scipy.linalg import expm
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 71)
    kind, km, shape = (orc.WILSON, KAPPA, orc.wilson_shape(L)) if kind_name == "wilson" else (orc.STAGGERED, MASS, orc.staggered_shape(L))
    eta = orc.gaussian_spinor(shape, 72)
    S0, X, Y, it, st = orc.fermi_action(kind, U, eta, L, km, bc=bc, eps=1e-26)
    assert st == 0 and S0 > 0
    G = orc.fermion_force(kind, U, X, Y, L, km, bc=bc)
    rng = np.random.default_rng(73)
    eps = 1e-4
    links = [(0, 1, 2, 3, 0), (3, 3, 0, 1, 3), (1, 3, 3, 0, 2), (2, 0, 3, 2, 1), (3, 2, 1, 3, 0)]   # (mu, t, z, y, x); incl. wraps
    for (mu, t, z, y, x) in links:
        T = _random_hermitian(rng)
        Uab = U[mu, t, z, y, x].T.copy()           # oracle layout stores [b, a]
        vals = []
        for sgn in (+1, -1):
            Up = U.copy()
            Up[mu, t, z, y, x] = (expm(1j * sgn * eps * T) @ Uab).T
            S, _, _, _, st = orc.fermi_action(kind, Up, eta, L, km, bc=bc, eps=1e-26)
            assert st == 0
            vals.append(S)
        fd = (vals[0] - vals[1]) / (2 * eps)
        Gab = G[mu, t, z, y, x].T
        an = -2.0 * np.trace(T @ Gab).imag
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (mu, t, z, y, x, fd, an)
        assert abs(an) > 1e-6                      # the check is not vacuous
