"""BASELINE.json's full sizes: one D and one D^+ per configuration AGAINST THE ORACLE (all host threads: 16^3x32 and 32^3x64 Wilson with the
12-real and the 18-real kernel instance, the fp32 site-pair kernel, 32^3x64 Wilson-clover, 48^3x96 staggered, the CG solution at 16^3x32), and
size-independent properties for what the oracle cannot afford (trajectories); hot start seed 111."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

KAPPA, CSW = 0.141139, 1.0
BC = (1, 1, 1, -1)


@pytest.fixture()
def orc_all_threads(orc):
    orc.set_threads(os.cpu_count() or 1)       # the oracle's site loops are OpenMP-parallel: same arithmetic per site whatever the thread count
    yield orc
    orc.set_threads(1)


@pytest.mark.parametrize("L", [(16, 16, 16, 32), (32, 32, 32, 64)])
def test_wilson_dslash_matches_oracle_at_baseline_sizes(lq, orc_all_threads, L):
    """configs[2] / configs[3] lattices, bench seeds: D b and D^+ b of BOTH kernel instances (12-real links, all 18 reals) against the oracle.  A
    defect that is consistent across the kernel's own variants (index width, the workgroup map at many chunks per plane) cannot hide here."""
    orc = orc_all_threads
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": BC})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    bh = b.download()
    y = b.similar()
    assert lat.get_param("dslash_variant") == 1 and lat.get_param("dslash_pipe") == 2
    for dagger in (False, True):
        ref = orc.wilson_D(Uh, bh, L, KAPPA, 1.0, BC, dagger=dagger)
        for recon, active in ((12, 1), (18, 0)):
            lat.set_param("gauge_recon", recon)
            lq.mul_(y, D.adjoint() if dagger else D, b)
            assert lat.get_param("recon_active") == active           # which instance ran: wilson_dirsplit_s<.,true,.> (12 reals) | wilson_dirsplit<.,false,.>
            assert rel_err(y.download(), ref) < 1e-13, (L, dagger, recon)
        lat.set_param("gauge_recon", 12)
    if L == (32, 32, 32, 64):
        # the fp32 site-pair kernel of the mixed-precision solvers (pairs sites t and t + T/2: a large-extent defect of that pairing shows here only)
        ref = orc.wilson_D(Uh, bh, L, KAPPA, 1.0, BC)
        for dagger in (False, True):
            if dagger:
                ref = orc.wilson_D(Uh, bh, L, KAPPA, 1.0, BC, dagger=True)
            lq.mul_f32_(y, D.adjoint() if dagger else D, b)
            assert lat.get_param("pair32_active") == 1
            assert rel_err(y.download(), ref) < 5e-6, dagger


def test_cg_solution_matches_oracle_16x16x16x32(lq, orc_all_threads):
    """configs[2] lattice: the fused CG (D and the update-mode D^+ kernels, deferred x update) against the oracle's plain CG -- solution, iteration
    count and the true residual."""
    orc = orc_all_threads
    L = (16, 16, 16, 32)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "r": 1.0, "boundarycondition": BC, "eps_CG": 1e-19})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    bh = b.download()
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, bh, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st == 0 and rr < 1e-19 and abs(it - ito) <= 1
    assert rel_err(x.download(), xo) < 1e-9
    t = orc.wilson_D(Uh, orc.wilson_D(Uh, x.download(), L, KAPPA, 1.0, BC), L, KAPPA, 1.0, BC, dagger=True) - bh
    assert np.vdot(t, t).real < 4e-19                       # the true residual, recomputed by the oracle


def test_wilson_clover_dslash_matches_oracle_32x32x32x64(lq, orc_all_threads):
    """configs[3]: D_sw b and D_sw^+ b (fused A x epilogue, 12-real links) against the oracle's clover term + Wilson D."""
    orc = orc_all_threads
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW, "boundarycondition": BC})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    bh = b.download()
    y = b.similar()
    A = orc.clover_build(Uh, L, KAPPA, CSW)
    for dagger in (False, True):
        lq.mul_(y, D.adjoint() if dagger else D, b)
        assert lat.get_param("recon_active") == 1 and lat.get_param("clover_fused") == 1
        assert rel_err(y.download(), orc.wilson_clover_D(Uh, A, bh, L, KAPPA, 1.0, BC, dagger=dagger)) < 1e-13, dagger


def test_staggered_dslash_matches_oracle_48x48x48x96(lq, orc_all_threads):
    """configs[4]: D b and D^+ b at 48^3 x 96 (18 chunks per z-plane: the workgroup map's non-power-of-two branch), 12- and 18-real links."""
    orc = orc_all_threads
    L = (48, 48, 48, 96)
    mass = 0.05
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "boundarycondition": BC})
    b = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(b, 112)
    bh = b.download()
    y = b.similar()
    for dagger in (False, True):
        ref = orc.staggered_D(Uh, bh, L, mass, BC, dagger=dagger)
        for recon, active in ((12, 1), (18, 0)):
            lat.set_param("gauge_recon", recon)
            lq.mul_(y, D.adjoint() if dagger else D, b)
            assert lat.get_param("recon_active") == active
            assert rel_err(y.download(), ref) < 1e-13, (dagger, recon)
        lat.set_param("gauge_recon", 12)


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


def _leapfrog(lq, U, p, G, fa, phi, dtau, steps, beta):
    """runMD_QPQ! (standardMD.jl:127-144) with the fused four-direction calls; dtau < 0 retraces the trajectory."""
    for _ in range(steps):
        lq.U_update_(U, p, 0.5 * dtau)
        lq.P_update_(U, p, dtau, beta)
        for f, ph in zip(fa, phi):
            lq.calc_UdSfdU_(G, f, U, ph)
            lq.Traceless_antihermitian_add_(p, dtau, G)
        lq.U_update_(U, p, 0.5 * dtau)


def _trajectory_checks(lq, L, op_pars, fa_pars, dtau, steps, halve, seed=5):
    """One short HMC trajectory at a full-size lattice from a hot start: |dH| shrinks ~4x when dtau is halved at fixed length (leapfrog),
    and the trajectory integrated back with -dtau returns to the starting links and energy."""
    beta = 5.7
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    p, G, U0 = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
    lq.substitute_U_(U0, U)
    fa, phi = [], []
    for k, (op_par, fa_par) in enumerate(zip(op_pars, fa_pars)):
        D = lq.Dirac_operator(U, None, dict(op_par, boundarycondition=(1, 1, 1, -1)))
        f = lq.FermiAction(D, fa_par)
        xi, ph = lq.Fermionfields(lat, D.kind), lq.Fermionfields(lat, D.kind)
        lq.gauss_sampling_in_action_(xi, U, f, seed + 10 * k)
        lq.sample_pseudofermions_(ph, U, f, xi)
        fa.append(f)
        phi.append(ph)

    def H():
        return lq.momentum_action(p) + lq.evaluate_GaugeAction(U, beta) + sum(lq.evaluate_FermiAction(f, U, ph) for f, ph in zip(fa, phi))

    dH = []
    for div in ((1, 2) if halve else (1,)):
        lq.substitute_U_(U, U0)
        lq.gauss_distribution_(p, seed + 1)
        H0 = H()
        _leapfrog(lq, U, p, G, fa, phi, dtau / div, steps * div, beta)
        dH.append(H() - H0)
    assert lat.get_param("recon_active") == 1                  # the links stayed on the group through the trajectory (md_reunitarize)
    if halve:
        assert 2.5 < dH[0] / dH[1] < 6.0, dH                   # second-order integrator
    assert abs(dH[-1]) < 1e-5 * abs(H0), (dH, H0)
    div = 2 if halve else 1
    _leapfrog(lq, U, p, G, fa, phi, -dtau / div, steps * div, beta)
    assert abs(H() - H0) < 1e-9 * abs(H0)
    assert np.max(np.abs(U.download() - U0.download())) < 1e-9
    return dH


def test_wilson_clover_32x32x32x64_short_trajectory(lq):
    """configs[3] at full size on one GPU: two-flavour Wilson-clover HMC, 4 (8) leapfrog steps"""
    dH = _trajectory_checks(lq, (32, 32, 32, 64), [{"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW, "eps_CG": 1e-18, "MaxCGstep": 3000}],
                            [{}], dtau=0.02, steps=4, halve=True)
    print("32^3x64 clover dH(dtau), dH(dtau/2) =", dH)


def test_staggered_rhmc_48x48x48x96_short_trajectory(lq):
    """configs[4] at full size on one GPU: staggered RHMC Nf = 2 + 1 (two rational pseudofermion actions), 2 leapfrog steps and back"""
    dH = _trajectory_checks(lq, (48, 48, 48, 96), [{"Dirac_operator": "Staggered", "mass": 0.05, "eps_CG": 1e-14, "MaxCGstep": 5000},
                                                  {"Dirac_operator": "Staggered", "mass": 0.1, "eps_CG": 1e-14, "MaxCGstep": 5000}],
                            [{"Nf": 2}, {"Nf": 1}], dtau=0.01, steps=2, halve=False)
    print("48^3x96 staggered 2+1 dH =", dH)


def test_action_solve_32x32x32x64_evenodd_route_against_cg_and_oracle_residual(lq, orc_all_threads):
    """configs[3]'s lattice, plain Wilson: evaluate_FermiAction / calc_UdSfdU! solve (D^+D) X = eta as Y = D^-+ eta, X = D^-1 Y through the even-odd
    BiCGStab (action_eo_solver = 1; 16384 chunks per parity: inner products from the Schur operator's epilogue, reductions as separate launches --
    the form the 1024-chunk tests do not reach).  Same X, Y and S_f as the CG on the normal equations, fewer than half its operator applications,
    and the reference's stopping rule on the residual it is stated for, recomputed by the oracle."""
    orc = orc_all_threads
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    eps = 1e-16
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC, "eps_CG": eps})
    fa = lq.FermiAction(D)
    eta = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(eta, 113)
    out = {}
    for mode in (1, 0):
        lat.set_param("action_eo_solver", mode)
        S, it = lq.evaluate_FermiAction(fa, U, eta, return_info=True)
        out[mode] = (S, it, fa._temporary_fermionfields[0].download(), fa._temporary_fermionfields[1].download())
    lat.set_param("action_eo_solver", 1)
    assert abs(out[1][0] - out[0][0]) < 1e-10 * abs(out[0][0])
    assert 2 * out[1][1] < out[0][1]                                   # iterations of 2 Schur applications (= 2 Dslash) each, both routes
    assert rel_err(out[1][2], out[0][2]) < 1e-9 and rel_err(out[1][3], out[0][3]) < 1e-9
    etah = eta.download()
    for mode in (1, 0):
        res = etah - orc.wilson_D(Uh, orc.wilson_D(Uh, out[mode][2], L, KAPPA, 1.0, BC), L, KAPPA, 1.0, BC, dagger=True)
        assert np.vdot(res, res).real < (eps if mode == 1 else 2 * eps), (mode, np.vdot(res, res).real)


def test_momentum_and_link_update_match_oracle_32x32x32x64(lq, orc_all_threads):
    """The gauge legs of an MD step at configs[3]'s size against the oracle: P_update! and U_update! as the library runs them by default (the momentum update and
    the link update behind it in ONE sweep into the second link buffer, lazy_merge = 2; Cayley-Hamilton series for exp) and as two separate passes."""
    orc = orc_all_threads
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    Uh = U.download()
    p = lq.initialize_TA_Gaugefields(U)
    lq.gauss_distribution_(p, 5)
    Ph = p.download()
    beta, eps, dt = 5.7, -0.013, 0.021
    Pref = orc.momentum_add_ta(Ph.copy(), eps, orc.gauge_force(Uh, L, beta), L)
    Uref = orc.link_update(Uh.copy(), Pref, dt, L)
    for merge in (2, 0):
        lat.set_param("lazy_merge", merge)
        lat.set_param("md_reunitarize", 0)          # the oracle's update is the literal one
        U.upload(Uh); p.upload(Ph)
        lq.P_update_(U, p, eps, beta)
        lq.U_update_(U, p, dt)
        assert lat.get_param("lazy_deferred") == (8 if merge else 0)
        assert rel_err(U.download(), Uref) < 1e-13, merge
        assert rel_err(p.download(), Pref) < 1e-13, merge
    lat.set_param("md_reunitarize", 1)
    U.upload(Uh); p.upload(Ph)
    lq.P_update_(U, p, eps, beta)
    lq.U_update_(U, p, dt)
    assert rel_err(U.download(), Uref) < 1e-12 and lq.unitarity_deviation(U) == 0.0       # projected in the same sweep: within rounding of the literal update


def test_fermion_force_sweeps_match_oracle_at_baseline_sizes(lq, orc_all_threads):
    """The outer-product sweep of calc_UdSfdU! at configs[3]'s and configs[4]'s sizes against the oracle (Wilson 32^3x64, staggered 48^3x96), given fields X, Y."""
    orc = orc_all_threads
    for L, kind, name, km in (((32, 32, 32, 64), lq.WILSON, "Wilson", KAPPA), ((48, 48, 48, 96), lq.STAGGERED, "Staggered", 0.05)):
        U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
        lat = U.lattice
        Uh = U.download()
        D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": km, "mass": km, "r": 1.0, "boundarycondition": BC})
        X, Y = lq.Fermionfields(lat, kind), lq.Fermionfields(lat, kind)
        lq.gauss_distribution_fermion_(X, 7)
        lq.gauss_distribution_fermion_(Y, 8)
        G = lq.Gaugefields(lat)
        lq.fermion_force_(G, D, X, Y)
        ref = orc.fermion_force(orc.WILSON if kind == lq.WILSON else orc.STAGGERED, Uh, X.download(), Y.download(), L, km, 1.0, BC)
        assert rel_err(G.download(), ref) < 1e-13, name
        for o in (G, X, Y, D, U):
            o.close()


def test_evenodd_bicgstab_32x32x32x64_dot_partial_layouts_and_forms(lq):
    """32^3 x 64, where the reductions of the even-odd chain are separate launches (8192 workgroups per hop): the [value][workgroup] layout of the hops' dot partials
    (bicg_dot_soa, default) against the [workgroup][value] one -- same additions, same bits -- for the fp64 chain and its fp32 twin inside the mixed-precision solver, and
    the merged update launch (bicg_fused = 4) against the two-launch form: same iteration count, solution to 1e-10, true residual of the full system below the target."""
    L = (32, 32, 32, 64)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "eps_CG": 1e-16, "MaxCGstep": 3000})
    D.method_CG = "bicgstab_evenodd"
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    got = {}
    for mixed in (0, 1):
        lat.set_param("bicg_mixed", mixed)
        for fused, soa in ((4, 1), (4, 0), (2, 1)):
            lat.set_param("bicg_fused", fused)
            lat.set_param("bicg_dot_soa", soa)
            x = b.similar()
            it, rr = lq.solve_DinvX_(x, D, b, return_info=True)
            r = b.similar()
            lq.mul_(r, D, x)
            lq.add_fermion_(r, -1.0, b)
            assert rr < 1e-16 and lq.dot(r, r).real < 1e-15, (mixed, fused, soa)
            got[(mixed, fused, soa)] = (it, x.download())
            x.close(); r.close()
        assert got[(mixed, 4, 1)][0] == got[(mixed, 4, 0)][0] and np.array_equal(got[(mixed, 4, 1)][1], got[(mixed, 4, 0)][1]), mixed
        assert abs(got[(mixed, 4, 1)][0] - got[(mixed, 2, 1)][0]) <= 1 and rel_err(got[(mixed, 4, 1)][1], got[(mixed, 2, 1)][1]) < (1e-9 if mixed else 1e-10), mixed
    lat.set_param("bicg_mixed", 0); lat.set_param("bicg_fused", 4); lat.set_param("bicg_dot_soa", 1)
