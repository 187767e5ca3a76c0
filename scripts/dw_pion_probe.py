"""Physics check of the Domainwall operator (one-off, not in the suite): quenched beta = 6.0, 16^3 x 32, L5 = 8, M = -1.8 (domain-wall height 1.8).  The physical quark
field lives on the walls -- for THIS operator's hops (-P_- psi(s+1) - P_+ psi(s-1)) the left-handed mode is bound to s = 0 and the right-handed one to s = L5 - 1:
q = P_- psi(0) + P_+ psi(L5-1), source B(s) = delta(s, L5-1) P_- eta + delta(s, 0) P_+ eta.  The pion made of q must become light with the quark mass: m_pi^2 linear in
m_f with an intercept at -m_res, m_res of order 1e-2 at L5 = 8 (the published residual mass at L5 = 16 is 0.00124; Blum et al., Phys. Rev. D 69 (2004) 074502)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import latticeqcd_jl_amd as lq
from test_gpu_quenched_literature import _quenched_configs, _cosh_mass

L, L5, M, beta = (16, 16, 16, 32), 8, -1.8, 6.0
masses = (0.02, 0.06)
G5 = np.array([[0, 0, -1, 0], [0, 0, 0, -1], [-1, 0, 0, 0], [0, -1, 0, 0]], dtype=np.complex128)
Pp, Pm = 0.5 * (np.eye(4) + G5), 0.5 * (np.eye(4) - G5)
cors = {m: [] for m in masses}
t0 = time.time()
for lat, U in _quenched_configs(lq, L, beta, 400, 4, 25, seed=66):
    for mf in masses:
        x5 = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=L5)
        D = lq.Dirac_operator(U, x5, {"Dirac_operator": "Domainwall", "mass": mf, "L5": L5, "M": M, "eps_CG": 1e-14, "MaxCGstep": 20000})
        b5, y5 = x5.similar(), x5.similar()
        C = np.zeros(L[3])
        its = []
        for ic in range(3):
            for isp in range(4):
                eta = np.zeros((4,) + (L[3], L[2], L[1], L[0], 3), dtype=np.complex128)
                eta[isp, 0, 0, 0, 0, ic] = 1.0
                B = np.zeros((L5,) + eta.shape, dtype=np.complex128)
                B[L5 - 1] = np.einsum("ab,b...->a...", Pm, eta)
                B[0] = np.einsum("ab,b...->a...", Pp, eta)
                b5.upload(B)
                lq.mul_(y5, D.adjoint(), b5)                       # psi = (D^+D)^-1 D^+ B
                lq.clear_fermion_(x5)
                it, rr = lq.solve_DinvX_(x5, lq.DdagD_operator(D), y5, return_info=True)
                its.append(it)
                w = x5.w
                q = np.einsum("ab,b...->a...", Pm, w[0].download()) + np.einsum("ab,b...->a...", Pp, w[L5 - 1].download())
                C += (np.abs(q) ** 2).sum(axis=(0, 2, 3, 4, 5))
        cors[mf].append(C)
        print("config %d m_f %.2f: CG iterations %d..%d, %.0f s" % (len(cors[mf]), mf, min(its), max(its), time.time() - t0), flush=True)
        for o in (b5, y5, x5, D):
            o.close()
m2 = {}
for mf in masses:
    Cs = np.array(cors[mf])
    m = _cosh_mass(Cs.mean(axis=0), 8, 14)
    m2[mf] = m * m
    print("m_f %.2f: m_pi a = %.4f, m_pi^2 = %.4f" % (mf, m, m * m))
slope = (m2[masses[1]] - m2[masses[0]]) / (masses[1] - masses[0])
mres = m2[masses[0]] / slope - masses[0]
print("m_pi^2 = %.3f (m_f + m_res), m_res = %.4f; ratio m_pi^2(%.2f) / m_pi^2(%.2f) = %.3f" % (slope, mres, masses[0], masses[1], m2[masses[0]] / m2[masses[1]]))
