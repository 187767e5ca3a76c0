"""The CPU restatement of the Domainwall operator, action and force (oracle/oracle.py) checked against identities that do not depend on it: the adjoint,
the heat-bath identity, the trivial m = 1 action, and finite differences of the action for the force.  (Parity with the reference is unpinned here as for
the Wilson operator underneath: the arithmetic lives in LatticeDiracOperators.jl.)"""
import numpy as np
import pytest
import scipy.linalg as sla

L, L5, M = (4, 4, 2, 4), 4, -1.0
BC = (1, 1, 1, -1)


@pytest.fixture(scope="module")
def setup(orc):
    U = orc.hot_gauge(L, 5)
    rng = np.random.default_rng(6)
    shp = (L5,) + orc.wilson_shape(L)
    f = lambda: rng.standard_normal(shp) + 1j * rng.standard_normal(shp)
    return U, f(), f()


def test_adjoint_and_gamma5_R_hermiticity(orc, setup):
    U, a, b = setup
    for mass in (0.1, 1.0):
        lhs = np.vdot(a, orc.domainwall_D(U, b, L, M, mass, BC))
        rhs = np.vdot(orc.domainwall_D(U, a, L, M, mass, BC, dagger=True), b)
        assert abs(lhs - rhs) < 1e-11 * abs(lhs)
        # D^+ = (g5 R) D (g5 R), R the reflection s -> L5 - 1 - s
        g5R = lambda v: np.stack([orc._spin(orc.GAMMA[4], v[s]) for s in range(L5)])[::-1]
        assert np.abs(g5R(orc.domainwall_D(U, g5R(a), L, M, mass, BC)) - orc.domainwall_D(U, a, L, M, mass, BC, dagger=True)).max() < 1e-12


def test_heat_bath_identity_and_trivial_action_at_pauli_villars_mass(orc, setup):
    U, xi, phi = setup
    ph = orc.domainwall_sample(U, xi, L, M, 0.1, BC)
    S, _, _ = orc.domainwall_action(U, ph, L, M, 0.1, BC, eps=1e-24)
    assert abs(S - np.vdot(xi, xi).real) < 1e-9 * S
    S1, _, _ = orc.domainwall_action(U, phi, L, M, 1.0, BC, eps=1e-24)
    assert abs(S1 - np.vdot(phi, phi).real) < 1e-9 * S1                     # D = D_PV: the reference's test (Domainwall_m = 1) has S = phi^+ phi
    assert np.abs(orc.domainwall_force(U, phi, L, M, 1.0, BC)).max() < 1e-9  # ... and no force


def test_force_is_the_derivative_of_the_action(orc, setup):
    U, _, phi = setup
    mass = 0.2
    G = orc.domainwall_force(U, phi, L, M, mass, BC, eps=1e-24)
    rng = np.random.default_rng(8)
    for _ in range(3):
        mu, t, z, y, x = rng.integers(4), rng.integers(L[3]), rng.integers(L[2]), rng.integers(L[1]), rng.integers(L[0])
        T = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
        T = T + T.conj().T
        h = 1e-4
        S = []
        for e in (h, -h):
            V = U.copy()
            # host image U[mu,t,z,y,x,b,a] is the transpose of the matrix: (exp(i e T) U)^T = U^T exp(i e T)^T
            V[mu, t, z, y, x] = V[mu, t, z, y, x] @ sla.expm(1j * e * T).T
            S.append(orc.domainwall_action(V, phi, L, M, mass, BC, eps=1e-24)[0])
        fd = (S[0] - S[1]) / (2 * h)
        an = -2.0 * np.imag(np.trace(T @ G[mu, t, z, y, x].T))
        assert abs(fd - an) < 2e-6 * max(1.0, abs(fd)), (fd, an)
