"""BASELINE.json's full sizes through size-independent properties (the oracle does not finish these in seconds):
configs[3] 32^3x64 Wilson-clover and configs[4] 48^3x96 staggered, hot start seed 111."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAPPA, CSW = 0.141139, 1.0


def test_wilson_clover_32x32x32x64_identities(lq):
    assert lq.lib.device_count() > 0
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW, "eps_CG": 1e-16})
    a, b = lq.Fermionfields(lat, lq.WILSON), lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(a, 1)
    lq.gauss_distribution_fermion_(b, 112)
    Db, Dda, t = a.similar(), a.similar(), a.similar()
    lq.mul_(Db, D, b)
    lq.mul_(Dda, D.adjoint(), a)
    lhs, rhs = lq.dot(a, Db), np.conj(lq.dot(b, Dda))
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)                               # <a, D b> = conj <b, D^+ a>
    assert lat.get_param("recon_active") == 1                              # hot-start links are unitary: 12-real kernel
    # the same operator from the 18 stored reals and from the separate A x pass
    for key, val in (("gauge_recon", 18), ("clover_fused", 0)):
        lat.set_param(key, val)
        lq.mul_(t, D, b)
        lq.add_fermion_(t, -1.0, Db)
        assert lq.dot(t, t).real < 1e-24 * lq.dot(Db, Db).real, key
    lat.set_param("gauge_recon", 12)
    lat.set_param("clover_fused", 1)
    # the clover sums by plaquette transport (the partitioned build) give the same term
    lat.set_param("clover_transport", 1)
    D2 = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW})
    lq.mul_(t, D2, b)
    lq.add_fermion_(t, -1.0, Db)
    assert lq.dot(t, t).real < 1e-24 * lq.dot(Db, Db).real
    lat.set_param("clover_transport", 0)
    # Schur identity: the even-odd preconditioned solve and the plain solve give the same D^-1 b
    x1, x2 = b.similar(), b.similar()
    D.method_CG = "bicgstab_evenodd"
    it1, _ = lq.solve_DinvX_(x1, D, b, return_info=True)
    D.method_CG = "bicgstab"
    it2, _ = lq.solve_DinvX_(x2, D, b, return_info=True)
    assert it1 < it2
    for x in (x1, x2):
        lq.mul_(t, D, x)
        lq.add_fermion_(t, -1.0, b)
        assert lq.dot(t, t).real < 1e-15
    lq.add_fermion_(x1, -1.0, x2)
    assert lq.dot(x1, x1).real < 1e-14 * lq.dot(x2, x2).real


def test_staggered_48x48x48x96_identities(lq):
    L = (48, 48, 48, 96)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    mass = 0.05
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "eps_CG": 1e-12, "MaxCGstep": 3000})
    a, b = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(a, 1)
    lq.gauss_distribution_fermion_(b, 112)
    Db, Dda, t = a.similar(), a.similar(), a.similar()
    lq.mul_(Db, D, b)
    lq.mul_(Dda, D.adjoint(), a)
    lhs, rhs = lq.dot(a, Db), np.conj(lq.dot(b, Dda))
    assert abs(lhs - rhs) < 1e-11 * abs(lhs)
    # D + D^+ = 2 m (the hopping part is anti-Hermitian)
    lq.mul_(t, D.adjoint(), b)
    lq.add_fermion_(t, 1.0, Db)
    lq.add_fermion_(t, -2.0 * mass, b)
    assert lq.dot(t, t).real < 1e-24 * lq.dot(Db, Db).real
    # 12-real and 18-real links agree
    assert lat.get_param("recon_active") == 1
    lat.set_param("gauge_recon", 18)
    lq.mul_(t, D, b)
    lat.set_param("gauge_recon", 12)
    lq.add_fermion_(t, -1.0, Db)
    assert lq.dot(t, t).real < 1e-24 * lq.dot(Db, Db).real
    # CG and the mixed-precision CG reach the same solution; true residual recomputed
    A = lq.DdagD_operator(D)
    x1, x2 = b.similar(), b.similar()
    it, rr = lq.solve_DinvX_(x1, A, b, return_info=True)
    itm, outer, rrm = lq.solve_mixed_DinvX_(x2, A, b, return_info=True)
    assert rr < 1e-12 and rrm < 1e-12 and outer >= 2
    lq.mul_(t, A, x1)
    lq.add_fermion_(t, -1.0, b)
    assert lq.dot(t, t).real < 1e-11
    lq.add_fermion_(x1, -1.0, x2)
    assert lq.dot(x1, x1).real < 1e-12 * lq.dot(x2, x2).real
    # heat bath of the rational action (Nf = 2): S_f((D^+D)^(Nf/16) xi) = xi^+ xi
    fa = lq.FermiAction(D, {"Nf": 2, "rhmc_tol_action": 1e-10})
    xi, phi = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_sampling_in_action_(xi, U, fa, 113)
    D.eps_CG = 1e-14
    lq.sample_pseudofermions_(phi, U, fa, xi)
    S = lq.evaluate_FermiAction(fa, U, phi)
    assert abs(S / lq.dot(xi, xi).real - 1.0) < 1e-7
