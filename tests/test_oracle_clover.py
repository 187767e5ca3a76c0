"""Oracle self-checks of the clover (Sheikholeslami-Wohlert) term -- SURVEY.md 8(f) rank 2.  The reference rejects this operator
(src/system/universe.jl:129-131), so there is nothing to be pinned against: the textbook definition is checked through its
defining properties."""
import numpy as np
import pytest

KAPPA, CSW = 0.141139, 1.3
BC = (1, 1, 1, -1)


def _random_gauge_transform(orc, L, seed):
    rng = np.random.default_rng(seed)
    V = L[0] * L[1] * L[2] * L[3]
    return orc.random_su3(rng, V).reshape(L[3], L[2], L[1], L[0], 3, 3)       # g[t,z,y,x,a,b]


def _transform_links(U, g, L):
    """U_mu(x) -> g(x) U_mu(x) g(x+mu)^+   (oracle layout U[mu,t,z,y,x,b,a])."""
    out = np.empty_like(U)
    for mu in range(4):
        Um = np.swapaxes(U[mu], -1, -2)                       # [.., a, b]
        gs = np.roll(g, -1, axis=3 - mu)                      # g(x+mu)
        out[mu] = np.swapaxes(g @ Um @ gs.conj().swapaxes(-1, -2), -1, -2)
    return out


def test_clover_matrix_properties(orc):
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 401)
    A = orc.clover_build(U, L, KAPPA, CSW)
    assert np.abs(A - A.conj().swapaxes(-1, -2)).max() < 1e-14                                      # Hermitian
    g5 = np.kron(orc.GAMMA[4], np.eye(3))
    assert np.abs(g5 @ A @ g5 - A).max() < 1e-14                                                    # commutes with gamma5
    assert np.abs(np.trace(A, axis1=-2, axis2=-1) - 12.0).max() < 1e-13                             # sigma F is traceless
    assert np.abs(A - np.eye(12)).max() > 1e-2                                                      # and not trivial
    assert np.abs(orc.clover_build(U, L, KAPPA, 0.0) - np.eye(12)).max() == 0.0                     # c_sw = 0
    assert np.abs(orc.clover_build(orc.unit_gauge(L), L, KAPPA, CSW) - np.eye(12)).max() < 1e-15    # F = 0 on a trivial field
    # pure gauge U_mu(x) = g(x) g(x+mu)^+ has F = 0 as well
    g = _random_gauge_transform(orc, L, 402)
    Upg = _transform_links(orc.unit_gauge(L), g, L)
    assert np.abs(orc.clover_build(Upg, L, KAPPA, CSW) - np.eye(12)).max() < 1e-13


def test_clover_operator_identities(orc):
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 403)
    A = orc.clover_build(U, L, KAPPA, CSW)
    a = orc.gaussian_spinor(orc.wilson_shape(L), 404)
    b = orc.gaussian_spinor(orc.wilson_shape(L), 405)
    Db = orc.wilson_clover_D(U, A, b, L, KAPPA, 1.0, BC)
    Dda = orc.wilson_clover_D(U, A, a, L, KAPPA, 1.0, BC, dagger=True)
    assert abs(np.vdot(a, Db) - np.conj(np.vdot(b, Dda))) < 1e-10 * abs(np.vdot(a, Db))            # true adjoint
    # c_sw = 0 is the Wilson operator
    A0 = orc.clover_build(U, L, KAPPA, 0.0)
    assert np.array_equal(orc.wilson_clover_D(U, A0, b, L, KAPPA, 1.0, BC), orc.wilson_D(U, b, L, KAPPA, 1.0, BC))
    # gauge covariance: D[U^g] psi^g = (D[U] psi)^g  (periodic boundary conditions so that g need not respect the twist)
    per = (1, 1, 1, 1)
    g = _random_gauge_transform(orc, L, 406)
    Ug = _transform_links(U, g, L)
    Ag = orc.clover_build(Ug, L, KAPPA, CSW)
    rot = lambda psi: np.einsum("tzyxab,stzyxb->stzyxa", g, psi)
    lhs = orc.wilson_clover_D(Ug, Ag, np.ascontiguousarray(rot(b)), L, KAPPA, 1.0, per)
    rhs = rot(orc.wilson_clover_D(U, A, b, L, KAPPA, 1.0, per))
    assert np.abs(lhs - rhs).max() < 1e-12
    # CG on D_sw^+ D_sw: true residual
    x, it, rr, st = orc.cg_clover(U, A, b, L, KAPPA, 1.0, BC, eps=1e-20)
    assert st == 0
    r = orc.wilson_clover_D(U, A, orc.wilson_clover_D(U, A, x, L, KAPPA, 1.0, BC), L, KAPPA, 1.0, BC, dagger=True) - b
    assert np.vdot(r, r).real < 2e-20


@pytest.mark.parametrize("dagger", [False, True])
def test_clover_even_odd_solve_equals_dense_solve(orc, dagger):
    """Schur identity with the clover term: the preconditioned solve gives D_sw^-1 b of the dense matrix."""
    L = (4, 2, 2, 2)
    U = orc.hot_gauge(L, 411)
    A = orc.clover_build(U, L, KAPPA, CSW)
    Ainv = orc.clover_invert(A, L)
    assert np.abs(A @ Ainv - np.eye(12)).max() < 1e-13
    shape = orc.wilson_shape(L)
    n = int(np.prod(shape))
    M = np.zeros((n, n), dtype=np.complex128)
    e = np.zeros(n, dtype=np.complex128)
    for j in range(n):
        e[:] = 0
        e[j] = 1
        M[:, j] = orc.wilson_clover_D(U, A, e.reshape(shape), L, KAPPA, 1.0, BC, dagger).reshape(n)
    b = orc.gaussian_spinor(shape, 412)
    x, it, rr, st = orc.wilson_clover_bicgstab_eo(U, A, b, L, KAPPA, 1.0, BC, dagger, eps=1e-24)
    assert st == 0 and it > 3
    xd = np.linalg.solve(M, b.reshape(n)).reshape(shape)
    assert np.abs(x - xd).max() < 1e-10 * np.abs(xd).max()
    # c_sw = 0 reduces to the Wilson even-odd solver, iteration for iteration
    A0 = orc.clover_build(U, L, KAPPA, 0.0)
    x0, it0, _, st0 = orc.wilson_clover_bicgstab_eo(U, A0, b, L, KAPPA, 1.0, BC, dagger, eps=1e-24)
    xw, itw, _, stw = orc.wilson_bicgstab_eo(U, b, L, KAPPA, 1.0, BC, dagger, eps=1e-24)
    assert st0 == 0 and stw == 0 and it0 == itw and np.abs(x0 - xw).max() < 1e-13


@pytest.mark.parametrize("csw", [CSW, 0.0])
def test_clover_fermion_force_is_the_derivative_of_the_action(orc, csw):
    """dS_f/d eps [U_mu(n) -> exp(i eps T) U_mu(n)] = -2 Im tr(T G_mu(n)) for S_f = phi^+ (D_sw^+ D_sw)^-1 phi, hopping + clover part,
    against central differences of the action itself (A is rebuilt from the varied links)."""
    from scipy.linalg import expm
    L = (4, 4, 4, 4)
    U = orc.hot_gauge(L, 421)
    phi = orc.gaussian_spinor(orc.wilson_shape(L), 422)

    def action(Ut):
        At = orc.clover_build(Ut, L, KAPPA, csw)
        X, _, _, st = orc.cg_clover(Ut, At, phi, L, KAPPA, 1.0, BC, eps=1e-26)
        assert st == 0
        return np.vdot(phi, X).real

    A = orc.clover_build(U, L, KAPPA, csw)
    S0, G, X, Y = orc.clover_fermion_force(U, A, phi, L, KAPPA, csw, 1.0, BC, eps=1e-26)
    assert abs(S0 - action(U)) < 1e-10 * S0
    if csw == 0.0:
        assert np.array_equal(G, orc.fermion_force(orc.WILSON, U, X, Y, L, KAPPA, 1.0, BC))
    else:
        Gc = G - orc.fermion_force(orc.WILSON, U, X, Y, L, KAPPA, 1.0, BC)
        assert np.abs(Gc).max() > 1e-3 * np.abs(G).max()          # the clover part is not negligible in this check
    rng = np.random.default_rng(423)
    eps = 1e-4
    for (mu, t, z, y, x) in [(0, 1, 2, 3, 0), (3, 3, 0, 1, 3), (1, 3, 3, 0, 2), (2, 0, 3, 2, 1)]:       # incl. links on the wrap
        T = sum(c * g for c, g in zip(rng.normal(size=8), orc.GELLMANN))
        Uab = U[mu, t, z, y, x].T.copy()
        vals = []
        for sgn in (+1, -1):
            Up = U.copy()
            Up[mu, t, z, y, x] = (expm(1j * sgn * eps * T) @ Uab).T
            vals.append(action(Up))
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = -2.0 * np.trace(T @ G[mu, t, z, y, x].T).imag
        assert abs(fd - an) < 1e-6 * max(1.0, abs(an)), (mu, t, z, y, x, fd, an)
        assert abs(an) > 1e-6
