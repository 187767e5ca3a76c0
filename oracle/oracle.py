"""ctypes/numpy front end of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see lqcd_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED at the Dslash/CG level; pinned on the reference's gauge fixtures (formats, index
order, plaquette) -- see lqcd_oracle.h for the full statement.

Array conventions (numpy, C order, complex128) -- same memory image as the reference's Julia arrays:
  gauge     U[mu, t, z, y, x, b, a]      == Julia U[mu][a,b,x,y,z,t]
  Wilson    psi[s, t, z, y, x, c]        == Julia psi[c,x,y,z,t,s]
  staggered psi[t, z, y, x, c]           == Julia psi[c,x,y,z,t,1]
L is always given as (NX, NY, NZ, NT).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

WILSON, STAGGERED = 0, 1


def build():
    """Compile liblqcd_oracle.so with gcc (recipe: oracle/Makefile)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liblqcd_oracle.so")
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        _LIB.orc_plaquette.restype = C.c_double
        _LIB.orc_unitarity_dev.restype = C.c_double
        _LIB.orc_fermi_action.restype = C.c_double
        _LIB.orc_gauge_action.restype = C.c_double
        _LIB.orc_momentum_action.restype = C.c_double
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _i4(v):
    return (C.c_int * 4)(*[int(x) for x in v])


def set_threads(n):
    lib().orc_set_threads(int(n))


def numa_copy(a, nblk):
    """A copy of `a` whose pages are first touched by the threads that own them in the oracle's threaded loops (set_threads first): nblk blocks --
    the four directions of a gauge field, the four spin components of a Wilson spinor (reference layout: they are the slowest index)."""
    a = np.ascontiguousarray(a)
    out = np.empty_like(a)
    lib().orc_numa_copy(_p(out), _p(a), C.c_long(a.size), int(nblk))
    return out


def gauge_shape(L):
    return (4, L[3], L[2], L[1], L[0], 3, 3)


def wilson_shape(L):
    return (4, L[3], L[2], L[1], L[0], 3)


def staggered_shape(L):
    return (L[3], L[2], L[1], L[0], 3)


def _chk(a, shape):
    assert a.dtype == np.complex128 and a.flags.c_contiguous and a.shape == tuple(shape), (a.dtype, a.shape, shape)


def plaquette(U, L):
    _chk(U, gauge_shape(L))
    return lib().orc_plaquette(_p(U), _i4(L))


def unitarity_dev(U, L):
    _chk(U, gauge_shape(L))
    return lib().orc_unitarity_dev(_p(U), _i4(L))


def wilson_D(U, psi, L, kappa, r=1.0, bc=(1, 1, 1, -1), dagger=False):
    _chk(U, gauge_shape(L)); _chk(psi, wilson_shape(L))
    out = np.empty_like(psi)
    lib().orc_wilson_D(_p(out), _p(U), _p(psi), _i4(L), C.c_double(kappa), C.c_double(r), _i4(bc), int(dagger))
    return out


def wilson_hop_parity(U, psi, L, r=1.0, bc=(1, 1, 1, -1), dagger=False, out_parity=0):
    _chk(U, gauge_shape(L)); _chk(psi, wilson_shape(L))
    out = np.empty_like(psi)
    lib().orc_wilson_hop_parity(_p(out), _p(U), _p(psi), _i4(L), C.c_double(r), _i4(bc), int(dagger), int(out_parity))
    return out


def staggered_D(U, psi, L, mass, bc=(1, 1, 1, -1), dagger=False):
    _chk(U, gauge_shape(L)); _chk(psi, staggered_shape(L))
    out = np.empty_like(psi)
    lib().orc_staggered_D(_p(out), _p(U), _p(psi), _i4(L), C.c_double(mass), _i4(bc), int(dagger))
    return out


def apply_D(kind, U, psi, L, km, r=1.0, bc=(1, 1, 1, -1), dagger=False):
    if kind == WILSON:
        return wilson_D(U, psi, L, km, r, bc, dagger)
    return staggered_D(U, psi, L, km, bc, dagger)


def cg_DdagD(kind, U, b, L, km, r=1.0, bc=(1, 1, 1, -1), eps=1e-19, maxiter=3000, x0=None):
    """Returns (x, iters, final_rr, status) with status 0 = converged."""
    x = np.zeros_like(b) if x0 is None else x0.copy()
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_cg_DdagD(int(kind), _p(x), _p(U), _p(b), _i4(L), C.c_double(km), C.c_double(r), _i4(bc),
                            C.c_double(eps), int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def cg_DdagD_fixed(kind, U, b, L, km, r=1.0, bc=(1, 1, 1, -1), niter=1):
    x = np.zeros_like(b)
    lib().orc_cg_DdagD_fixed(int(kind), _p(x), _p(U), _p(b), _i4(L), C.c_double(km), C.c_double(r), _i4(bc),
                             int(niter))
    return x


def multishift_cg(kind, U, b, L, km, sigmas, r=1.0, bc=(1, 1, 1, -1), eps=1e-19, maxiter=3000):
    """Returns (x0, [x_j], iters, resid, status): (D^+D + sigma_j) x_j = b and the unshifted solution x0."""
    sig = np.ascontiguousarray(sigmas, dtype=np.float64)
    x0 = np.zeros_like(b)
    xs = np.zeros((len(sig),) + b.shape, dtype=np.complex128)
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_multishift_cg(int(kind), _p(x0), _p(xs), _p(U), _p(b), _i4(L), C.c_double(km), C.c_double(r), _i4(bc),
                                 _p(sig), len(sig), C.c_double(eps), int(maxiter), C.byref(it), C.byref(rr))
    return x0, [xs[j] for j in range(len(sig))], it.value, rr.value, st


def fermi_action(kind, U, eta, L, km, r=1.0, bc=(1, 1, 1, -1), eps=1e-19, maxiter=3000):
    """S_f = eta^+ (D^+D)^-1 eta.  Returns (S_f, X, Y, iters, status) with X = (D^+D)^-1 eta, Y = D X."""
    X, Y = np.zeros_like(eta), np.zeros_like(eta)
    it, st = C.c_int(0), C.c_int(0)
    S = lib().orc_fermi_action(int(kind), _p(X), _p(Y), _p(U), _p(eta), _i4(L), C.c_double(km), C.c_double(r), _i4(bc),
                               C.c_double(eps), int(maxiter), C.byref(it), C.byref(st))
    return S, X, Y, it.value, st.value


def fermion_force(kind, U, X, Y, L, km, r=1.0, bc=(1, 1, 1, -1)):
    """G_mu(n) = "U dS_f/dU" in the gauge layout: dS_f/d eps under U_mu(n) -> exp(i eps T) U_mu(n) is -2 Im tr(T G_mu(n))."""
    G = np.zeros(gauge_shape(L), dtype=np.complex128)
    if kind == WILSON:
        lib().orc_wilson_force(_p(G), _p(U), _p(X), _p(Y), _i4(L), C.c_double(km), C.c_double(r), _i4(bc))
    else:
        lib().orc_staggered_force(_p(G), _p(U), _p(X), _p(Y), _i4(L), _i4(bc))
    return G


def rational_apply(kind, U, x, L, km, a0, res, poles, r=1.0, bc=(1, 1, 1, -1), eps=1e-22, maxiter=3000):
    """y = a0 x + sum_k res_k (D^+D + pole_k)^-1 x -- the rational (RHMC) action of the reference's general-Nf staggered runs
    (test/test_Nf2.toml:8, test/test_Nf3.toml:8, README.md:132) composed from the multi-shift CG.  Returns (y, [X_k])."""
    _, xs, _, _, st = multishift_cg(kind, U, x, L, km, poles, r, bc, eps, maxiter)
    assert st == 0
    y = a0 * x
    for rk, xk in zip(res, xs):
        y = y + rk * xk
    return y, xs


def rational_force(kind, U, phi, L, km, res, poles, r=1.0, bc=(1, 1, 1, -1), eps=1e-22, maxiter=3000):
    """"U dS/dU" of S = phi^+ [a0 + sum_k res_k (D^+D + pole_k)^-1] phi: sum_k res_k G[X_k, D X_k] (term-by-term derivative)."""
    _, xs, _, _, st = multishift_cg(kind, U, phi, L, km, poles, r, bc, eps, maxiter)
    assert st == 0
    G = np.zeros(gauge_shape(L), dtype=np.complex128)
    for rk, xk in zip(res, xs):
        G += rk * fermion_force(kind, U, xk, apply_D(kind, U, xk, L, km, r, bc), L, km, r, bc)
    return G


def dense_DdagD(kind, U, L, km, r=1.0, bc=(1, 1, 1, -1)):
    """The matrix of D^+D on a small lattice, column by column (exact spectral functions for the rational-action tests)."""
    shape = staggered_shape(L) if kind == STAGGERED else wilson_shape(L)
    n = int(np.prod(shape))
    A = np.zeros((n, n), dtype=np.complex128)
    e = np.zeros(n, dtype=np.complex128)
    for j in range(n):
        e[:] = 0.0
        e[j] = 1.0
        v = apply_D(kind, U, e.reshape(shape), L, km, r, bc)
        A[:, j] = apply_D(kind, U, v, L, km, r, bc, True).reshape(n)
    return A


def gauge_action(U, L, beta):
    return lib().orc_gauge_action(_p(U), _i4(L), C.c_double(beta))


def gauge_force(U, L, beta):
    G = np.zeros(gauge_shape(L), dtype=np.complex128)
    lib().orc_gauge_force(_p(G), _p(U), _i4(L), C.c_double(beta))
    return G


def momentum_add_ta(P, c, G, L):
    lib().orc_momentum_add_ta(_p(P), C.c_double(c), _p(G), _i4(L))
    return P


def momentum_action(P, L):
    return lib().orc_momentum_action(_p(P), _i4(L))


def link_update(U, P, dt, L):
    lib().orc_link_update(_p(U), _p(P), C.c_double(dt), _i4(L))
    return U


GELLMANN = np.zeros((8, 3, 3), dtype=np.complex128)
GELLMANN[0][0, 1] = GELLMANN[0][1, 0] = 1
GELLMANN[1][0, 1] = -1j; GELLMANN[1][1, 0] = 1j
GELLMANN[2][0, 0] = 1; GELLMANN[2][1, 1] = -1
GELLMANN[3][0, 2] = GELLMANN[3][2, 0] = 1
GELLMANN[4][0, 2] = -1j; GELLMANN[4][2, 0] = 1j
GELLMANN[5][1, 2] = GELLMANN[5][2, 1] = 1
GELLMANN[6][1, 2] = -1j; GELLMANN[6][2, 1] = 1j
GELLMANN[7] = np.diag([1, 1, -2]) / np.sqrt(3)


def gaussian_momenta(L, seed):
    """P = i sum_a pi_a lambda_a / 2 with pi_a ~ N(0,1): K = -sum tr P^2 = sum pi_a^2 / 2.  Oracle layout [.., b, a]."""
    rng = np.random.default_rng(seed)
    pi = rng.standard_normal((4, L[3], L[2], L[1], L[0], 8))
    P = 1j * np.einsum("...a,aij->...ij", pi, GELLMANN / 2)          # [.., a, b]
    return np.ascontiguousarray(np.swapaxes(P, -1, -2))


def clover_build(U, L, kappa, csw):
    """A(x) = 1 + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu}(x): array [t,z,y,x,12,12], row/column index s*3+c."""
    A = np.zeros((L[3], L[2], L[1], L[0], 12, 12), dtype=np.complex128)
    lib().orc_clover_build(_p(A), _p(U), _i4(L), C.c_double(kappa), C.c_double(csw))
    return A


def wilson_clover_D(U, A, psi, L, kappa, r=1.0, bc=(1, 1, 1, -1), dagger=False):
    out = np.empty_like(psi)
    lib().orc_wilson_clover_D(_p(out), _p(U), _p(A), _p(psi), _i4(L), C.c_double(kappa), C.c_double(r), _i4(bc), int(dagger))
    return out


def cg_clover(U, A, b, L, kappa, r=1.0, bc=(1, 1, 1, -1), eps=1e-19, maxiter=3000):
    x = np.zeros_like(b)
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_cg_clover(_p(x), _p(U), _p(A), _p(b), _i4(L), C.c_double(kappa), C.c_double(r), _i4(bc), C.c_double(eps),
                             int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def clover_fermion_force(U, A, phi, L, kappa, csw, r=1.0, bc=(1, 1, 1, -1), eps=1e-22):
    """S_f = phi^+ (D_sw^+ D_sw)^-1 phi and its "U dS/dU": hopping part (orc_wilson_force) + clover part.  Returns (S_f, G, X, Y)."""
    X, _, _, st = cg_clover(U, A, phi, L, kappa, r, bc, eps=eps)
    assert st == 0
    Y = wilson_clover_D(U, A, X, L, kappa, r, bc)
    G = fermion_force(WILSON, U, X, Y, L, kappa, r, bc)
    lib().orc_clover_force(_p(G), _p(U), _p(X), _p(Y), _i4(L), C.c_double(kappa), C.c_double(csw), 1)
    return np.vdot(phi, X).real, G, X, Y


def clover_invert(A, L):
    inv = np.zeros_like(A)
    lib().orc_clover_invert(_p(inv), _p(A), _i4(L))
    return inv


def wilson_clover_bicgstab_eo(U, A, b, L, kappa, r=1.0, bc=(1, 1, 1, -1), dagger=False, eps=1e-19, maxiter=3000):
    """Even-odd preconditioned BiCGStab for D_sw x = b (or D_sw^+).  Returns (x, iters, resid of the Schur system, status)."""
    x = np.zeros_like(b)
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_wilson_clover_bicgstab_eo(_p(x), _p(U), _p(A), _p(b), _i4(L), C.c_double(kappa), C.c_double(r), _i4(bc),
                                             int(bool(dagger)), C.c_double(eps), int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def bicg(kind, U, b, L, km, r=1.0, bc=(1, 1, 1, -1), dagger=False, eps=1e-19, maxiter=3000):
    x = np.zeros_like(b)
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_bicg(int(kind), _p(x), _p(U), _p(b), _i4(L), C.c_double(km), C.c_double(r), _i4(bc), int(bool(dagger)), C.c_double(eps),
                        int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def bicgstab(kind, U, b, L, km, r=1.0, bc=(1, 1, 1, -1), dagger=False, eps=1e-19, maxiter=3000, x0=None):
    x = np.zeros_like(b) if x0 is None else x0.copy()
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_bicgstab(int(kind), _p(x), _p(U), _p(b), _i4(L), C.c_double(km), C.c_double(r), _i4(bc),
                            int(dagger), C.c_double(eps), int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def wilson_bicgstab_eo(U, b, L, kappa, r=1.0, bc=(1, 1, 1, -1), dagger=False, eps=1e-19, maxiter=3000):
    x = np.zeros_like(b)
    it, rr = C.c_int(0), C.c_double(0)
    st = lib().orc_wilson_bicgstab_eo(_p(x), _p(U), _p(b), _i4(L), C.c_double(kappa), C.c_double(r), _i4(bc),
                                      int(dagger), C.c_double(eps), int(maxiter), C.byref(it), C.byref(rr))
    return x, it.value, rr.value, st


def dot(a, b):
    re, im = C.c_double(0), C.c_double(0)
    lib().orc_dot(_p(a), _p(b), C.c_long(a.size), C.byref(re), C.byref(im))
    return complex(re.value, im.value)


# ---------------------------------------------------------------- pure-numpy helpers for tests
GAMMA = np.zeros((5, 4, 4), dtype=np.complex128)
GAMMA[0][0, 3] = -1j; GAMMA[0][1, 2] = -1j; GAMMA[0][2, 1] = 1j; GAMMA[0][3, 0] = 1j
GAMMA[1][0, 3] = -1; GAMMA[1][1, 2] = 1; GAMMA[1][2, 1] = 1; GAMMA[1][3, 0] = -1
GAMMA[2][0, 2] = -1j; GAMMA[2][1, 3] = 1j; GAMMA[2][2, 0] = 1j; GAMMA[2][3, 1] = -1j
GAMMA[3] = np.diag([1, 1, -1, -1])
GAMMA[4][0, 2] = -1; GAMMA[4][1, 3] = -1; GAMMA[4][2, 0] = -1; GAMMA[4][3, 1] = -1


def random_su3(rng, n):
    """n random SU(3) matrices, row-wise Gram-Schmidt (the reference's hot start)."""
    m = rng.standard_normal((n, 3, 3)) + 1j * rng.standard_normal((n, 3, 3))
    r0 = m[:, 0] / np.linalg.norm(m[:, 0], axis=1, keepdims=True)
    r1 = m[:, 1] - np.sum(np.conj(r0) * m[:, 1], axis=1, keepdims=True) * r0
    r1 /= np.linalg.norm(r1, axis=1, keepdims=True)
    r2 = np.conj(np.cross(r0, r1))
    return np.stack([r0, r1, r2], axis=1)


def hot_gauge(L, seed):
    """Seeded hot-start gauge field in oracle layout U[mu,t,z,y,x,b,a]."""
    rng = np.random.default_rng(seed)
    V = L[0] * L[1] * L[2] * L[3]
    m = random_su3(rng, 4 * V).reshape(4, L[3], L[2], L[1], L[0], 3, 3)  # [.., a, b]
    return np.ascontiguousarray(np.swapaxes(m, -1, -2))  # -> [.., b, a]


def unit_gauge(L):
    U = np.zeros(gauge_shape(L), dtype=np.complex128)
    for a in range(3):
        U[..., a, a] = 1.0
    return U


def gaussian_spinor(shape, seed):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex128)


# ---------------------------------------------------------------- Domainwall (Shamir) operator: numpy over the C Wilson operator
# Restates the definition csrc/domainwall.hip gives (textbook Shamir operator in the conventions of SURVEY.md Appendix A; the reference's own
# arithmetic is in LatticeDiracOperators.jl, not under the reference tree: [EXT-RECALL], parity unpinned like the Wilson operator underneath).
# Call sites in the reference: src/system/universe.jl:116-128 (params "mass", "L5", "M"), test/test_domainwallhmc.toml, test/runtests.jl:132-137.
# Five-dimensional fields: psi5[s5, spin, t, z, y, x, c].
def _spin(M, psi):
    return np.einsum("ab,b...->a...", M, psi)


def domainwall_D(U, psi5, L, M, mass, bc=(1, 1, 1, -1), dagger=False):
    """(D5 psi)(s) = D4 psi(s) + psi(s) - P_- psi(s+1) - P_+ psi(s-1), psi(L5+1) = -m psi(1), psi(0) = -m psi(L5); D4 = (4 + M) - H/2.
    The adjoint exchanges P_+ and P_- in the fifth-direction hops."""
    L5 = psi5.shape[0]
    one = np.eye(4, dtype=np.complex128)
    Pp, Pm = 0.5 * (one + GAMMA[4]), 0.5 * (one - GAMMA[4])
    PA, PB = (Pp, Pm) if dagger else (Pm, Pp)
    out = np.empty_like(psi5)
    for s in range(L5):
        out[s] = (4.0 + M) * psi5[s] + wilson_D(U, np.ascontiguousarray(psi5[s]), L, 0.5, 1.0, bc, dagger)      # wilson_D(kappa = 1/2) = psi - H psi / 2
        cu = -1.0 if s + 1 < L5 else mass
        cd = -1.0 if s >= 1 else mass
        out[s] += cu * _spin(PA, psi5[(s + 1) % L5]) + cd * _spin(PB, psi5[(s - 1) % L5])
    return out


def domainwall_cg(U, b5, L, M, mass, bc=(1, 1, 1, -1), eps=1e-19, maxiter=3000):
    """x = (D5^+ D5)^-1 b, CG from a zero guess with the reference's stopping rule real(r.r) < eps.  Returns (x, iterations, r.r)."""
    A = lambda v: domainwall_D(U, domainwall_D(U, v, L, M, mass, bc), L, M, mass, bc, dagger=True)
    x = np.zeros_like(b5)
    r = b5.copy()
    p = r.copy()
    rr = np.vdot(r, r).real
    it = 0
    while rr >= eps and it < maxiter:
        q = A(p)
        alpha = rr / np.vdot(p, q).real
        x += alpha * p
        r -= alpha * q
        rr_new = np.vdot(r, r).real
        p = r + (rr_new / rr) * p
        rr = rr_new
        it += 1
    return x, it, rr


def domainwall_action(U, phi5, L, M, mass, bc=(1, 1, 1, -1), eps=1e-19):
    """S = psi^+ (D^+D)^-1 psi, psi = D_PV^+ phi, D_PV = D5(m = 1).  Returns (S, X, Y = D X)."""
    psi = domainwall_D(U, phi5, L, M, 1.0, bc, dagger=True)
    X, _, _ = domainwall_cg(U, psi, L, M, mass, bc, eps)
    return np.vdot(psi, X).real, X, domainwall_D(U, X, L, M, mass, bc)


def domainwall_sample(U, xi5, L, M, mass, bc=(1, 1, 1, -1), eps=1e-22):
    """phi = D_PV^-+ D^+ xi = D_PV (D_PV^+ D_PV)^-1 D^+ xi, so that S(phi) = xi^+ xi."""
    w = domainwall_D(U, xi5, L, M, mass, bc, dagger=True)
    z, _, _ = domainwall_cg(U, w, L, M, 1.0, bc, eps)
    return domainwall_D(U, z, L, M, 1.0, bc)


def domainwall_force(U, phi5, L, M, mass, bc=(1, 1, 1, -1), eps=1e-22):
    """G in the convention of fermion_force: dS = -2 Re[(Y - phi)^+ dD X], only the four-dimensional hops (coefficient 1/2) carry links."""
    _, X, Y = domainwall_action(U, phi5, L, M, mass, bc, eps)
    Z = Y - phi5
    G = np.zeros(gauge_shape(L), dtype=np.complex128)
    for s in range(phi5.shape[0]):
        G += fermion_force(WILSON, U, np.ascontiguousarray(X[s]), np.ascontiguousarray(Z[s]), L, 0.5, 1.0, bc)
    return G


# ---------------------------------------------------------------- stout smearing and its back-propagation (plaquette staples), pure numpy
# Call sites in the reference: src/system/universe.jl:147-171 (CovNeuralnet(U), STOUT_Layer(p.stout_loops, p.stout_ρ, U), push!),
# src/md/standardMD.jl:192-227 (P_update_fermion! with a CovNeuralnet: calc_smearedU, calc_UdSfdU! on the smeared links, back_prop),
# src/updates/standardHMC.jl:67-68.  The arithmetic is Gaugefields.jl's (not under the reference tree); this is Morningstar-Peardon's definition,
#     U'_mu(n) = exp(i Q_mu(n)) U_mu(n),   i Q = -rho TA(U_mu(n) A_mu(n)),   A = the six staples of the link (as in gauge_force),   TA(W) = (W - W^+)/2 - tr(W - W^+)/6,
# and the chain rule written with Frechet derivatives of exp instead of the closed-form B matrices (the same linear map).  Matrices below are [.., a, b]
# (the transpose of the host image [.., b, a]).
def _mat(U):
    return np.swapaxes(U, -1, -2)


def _sh(F, L, nu, step):
    """F(n + step nu_hat): axes of a link field F[t,z,y,x,a,b] are (t,z,y,x) = (3,2,1,0 in L order)."""
    return np.roll(F, -step, axis=3 - nu)


def _ta(W):
    X = 0.5 * (W - np.conj(np.swapaxes(W, -1, -2)))
    tr = np.trace(X, axis1=-2, axis2=-1) / 3.0
    return X - tr[..., None, None] * np.eye(3)


def _dag(A):
    return np.conj(np.swapaxes(A, -1, -2))


def _staple_sum(Um, L, mu):
    """A_mu(n) = sum_{nu != mu} [ U_nu(n+mu) U_mu(n+nu)^+ U_nu(n)^+ + U_nu(n+mu-nu)^+ U_mu(n-nu)^+ U_nu(n-nu) ]  (periodic links)."""
    A = np.zeros_like(Um[mu])
    for nu in range(4):
        if nu == mu:
            continue
        A += _sh(Um[nu], L, mu, 1) @ _dag(_sh(Um[mu], L, nu, 1)) @ _dag(Um[nu])
        lo = _dag(_sh(Um[nu], L, mu, 1)) @ _dag(Um[mu]) @ Um[nu]          # at n: U_nu(n+mu)^+ U_mu(n)^+ U_nu(n); needed at n - nu
        A += _sh(lo, L, nu, -1)
    return A


def _expm_batch(Z):
    import scipy.linalg as sla
    flat = Z.reshape(-1, 3, 3)
    return np.stack([sla.expm(z) for z in flat]).reshape(Z.shape)


def _expm_frechet_batch(Z, E):
    import scipy.linalg as sla
    fz, fe = Z.reshape(-1, 3, 3), E.reshape(-1, 3, 3)
    return np.stack([sla.expm_frechet(z, e, compute_expm=False) for z, e in zip(fz, fe)]).reshape(Z.shape)


def stout_smear(U, L, rho):
    """One STOUT layer with plaquette staples.  Returns the smeared links in the host layout."""
    Um = _mat(U)
    out = np.empty_like(Um)
    for mu in range(4):
        Z = -rho * _ta(Um[mu] @ _staple_sum(Um, L, mu))
        out[mu] = _expm_batch(Z) @ Um[mu]
    return np.ascontiguousarray(_mat(out))


def stout_backprop(Gs, U, L, rho):
    """Gs = the force field "U' dS/dU'" at the smeared links (convention of fermion_force: dS/d eps under U' -> exp(i eps T) U' is -2 Im tr(T Gs));
    returns G = "U dS/dU" at the thin links for the same S seen as a function of U.
        G_i = e^{-Z_i} Gs_i e^{Z_i} + (force of  S~ = -2 rho sum_j Re tr(U_j A_j N_j),  N_j = TA(L(Z_j, e^{-Z_j} Gs_j))  held fixed),
    L(Z, K) the Frechet derivative of exp at Z in direction K."""
    Um, Gm = _mat(U), _mat(Gs)
    Z = np.empty_like(Um)
    N = np.empty_like(Um)
    G = np.empty_like(Um)
    for mu in range(4):
        Z[mu] = -rho * _ta(Um[mu] @ _staple_sum(Um, L, mu))
        Em = _expm_batch(-Z[mu])
        K = Em @ Gm[mu]
        N[mu] = _ta(_expm_frechet_batch(Z[mu], K))
        G[mu] = K @ _dag(Em)                                   # e^{-Z} Gs e^{Z}  (Z anti-Hermitian: e^{Z} = (e^{-Z})^+)
    c0 = -2.0 * rho
    for mu in range(4):
        a = Um[mu]
        acc = np.zeros_like(a)
        for nu in range(4):
            if nu == mu:
                continue
            # the plaquette (n; mu, nu) with this link as `a`: b = U_nu(n+mu), c = U_mu(n+nu), d = U_nu(n)
            b, c, d = _sh(Um[nu], L, mu, 1), _sh(Um[mu], L, nu, 1), Um[nu]
            Na, Nb, Nc, Nd = N[mu], _sh(N[nu], L, mu, 1), _sh(N[mu], L, nu, 1), N[nu]
            acc += a @ b @ _dag(c) @ _dag(d) @ Na + a @ Nb @ b @ _dag(c) @ _dag(d) - Nd @ d @ c @ _dag(b) @ _dag(a) - d @ Nc @ c @ _dag(b) @ _dag(a)
            # the plaquette (n - nu; mu, nu) with this link as `c`: a' = U_mu(n-nu), b' = U_nu(n-nu+mu), d' = U_nu(n-nu)
            a2, b2, d2 = _sh(Um[mu], L, nu, -1), _sh(_sh(Um[nu], L, mu, 1), L, nu, -1), _sh(Um[nu], L, nu, -1)
            Na2, Nb2, Nd2, Nc2 = _sh(N[mu], L, nu, -1), _sh(_sh(N[nu], L, mu, 1), L, nu, -1), _sh(N[nu], L, nu, -1), N[mu]
            cc = a
            acc += cc @ _dag(b2) @ _dag(a2) @ Nd2 @ d2 + cc @ _dag(b2) @ _dag(a2) @ d2 @ Nc2 \
                - _dag(d2) @ Na2 @ a2 @ b2 @ _dag(cc) - _dag(d2) @ a2 @ Nb2 @ b2 @ _dag(cc)
        G[mu] += 0.5 * c0 * acc
    return np.ascontiguousarray(_mat(G))


def polyakov_loop(U, L):
    """1/(NC NX NY NZ) sum_x tr prod_t U_4(x, t) (calculate_Polyakov_loop of the reference's Polyakov_loop measurement; periodic links)."""
    Ut = _mat(U)[3]                      # [t, z, y, x, a, b]
    acc = Ut[0]
    for t in range(1, L[3]):
        acc = acc @ Ut[t]
    return complex(np.trace(acc, axis1=-2, axis2=-1).sum() / (3.0 * L[0] * L[1] * L[2]))
