"""The rational (RHMC) staggered action on the device: lqcd_rational_apply / lqcd_rational_force against the oracle's composition
(multi-shift CG + per-pole force), and the reference's two general-Nf HMC tests (test/runtests.jl:114-130 with test/test_Nf2.toml
and test/test_Nf3.toml: thermalised 4^4 configurations, beta = 5.7, mass = 0.5, dtau = 0.05, 20 MD steps, 10 trajectories; final
plaquette within 10 % of test/debugplaqdata.txt lines 9 and 10)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu

BETA, MASS = 5.7, 0.5
BC = (1, 1, 1, -1)
REF_PLAQ = {2: 0.56287171870089, 3: 0.5595757232711884}     # /root/reference/test/debugplaqdata.txt:9,10 (runtests.jl:116,125)


@pytest.mark.parametrize("nf", [2, 3, 1])
def test_rational_action_and_force_match_oracle(lq, orc, nf):
    assert lq.lib.device_count() > 0
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 821)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS, "boundarycondition": BC, "eps_CG": 1e-22})
    fa = lq.FermiAction(D, {"Nf": nf})
    assert fa.rational and len(fa.rhmc_MD[1]) < len(fa.rhmc_action[1])
    phih = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 822)
    phi = lq.Fermionfields(lat, lq.STAGGERED).upload(phih)
    a0, res, poles = fa.rhmc_action
    yo, _ = orc.rational_apply(orc.STAGGERED, Uh, phih, L, MASS, a0, res, poles, 1.0, BC)
    S = lq.evaluate_FermiAction(fa, U, phi)
    assert abs(S - np.vdot(phih, yo).real) < 1e-10 * abs(S)
    assert rel_err(fa._temporary_fermionfields[0].download(), yo) < 1e-10
    G = lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G, fa, U, phi)
    Go = orc.rational_force(orc.STAGGERED, Uh, phih, L, MASS, fa.rhmc_MD[1], fa.rhmc_MD[2], 1.0, BC)
    assert rel_err(G.download(), Go) < 1e-10
    # heat bath: S_f(phi = (D'D)^(Nf/16) xi) = xi' xi
    xi = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_sampling_in_action_(xi, U, fa, 823)
    lq.sample_pseudofermions_(phi, U, fa, xi)
    assert abs(lq.evaluate_FermiAction(fa, U, phi) / lq.dot(xi, xi).real - 1.0) < 1e-9


@pytest.mark.parametrize("mode", [1, 2])
def test_rational_action_with_per_pole_mixed_precision_solves(lq, orc, mode):
    """Tunable mixed_action_solver = 1: every pole is a mixed-precision solve with the staggered operator of mass sqrt(m^2 + pole);
    = 2: all poles through the mixed-precision multi-shift CG (one fp32 pass + fp64 defect correction per pole)
    (BASELINE.json configs[4]: RHMC with an fp32 inner / fp64 outer CG); same action and force as the fp64 multi-shift CG."""
    L = (6, 6, 4, 2)          # a partially filled chunk
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 831)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.1, "boundarycondition": BC, "eps_CG": 1e-20})
    fa = lq.FermiAction(D, {"Nf": 3})
    phih = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 832)
    phi = lq.Fermionfields(lat, lq.STAGGERED).upload(phih)
    G = lq.Gaugefields(lat)
    S0 = lq.evaluate_FermiAction(fa, U, phi)
    lq.calc_UdSfdU_(G, fa, U, phi)
    G0 = G.download()
    lat.set_param("mixed_action_solver", mode)
    S1 = lq.evaluate_FermiAction(fa, U, phi)
    lq.calc_UdSfdU_(G, fa, U, phi)
    lat.set_param("mixed_action_solver", 0)
    assert abs(S1 - S0) < 1e-9 * abs(S0) and rel_err(G.download(), G0) < 1e-8
    a0, res, poles = fa.rhmc_action
    yo, _ = orc.rational_apply(orc.STAGGERED, Uh, phih, L, 0.1, a0, res, poles, 1.0, BC)
    assert abs(S1 - np.vdot(phih, yo).real) < 1e-9 * abs(S1)
    # new links in the same handle: the cached fp32 copies must follow
    Uh2 = orc.hot_gauge(L, 833)
    U.upload(Uh2)
    lat.set_param("mixed_action_solver", mode)
    S2 = lq.evaluate_FermiAction(fa, U, phi)
    lat.set_param("mixed_action_solver", 0)
    yo2, _ = orc.rational_apply(orc.STAGGERED, Uh2, phih, L, 0.1, a0, res, poles, 1.0, BC)
    assert abs(S2 - np.vdot(phih, yo2).real) < 1e-9 * abs(S2)


def test_fermion_force_acc_scales_and_accumulates(lq, orc):
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 824)
    U = lq.Gaugefields(lat).upload(Uh)
    for kind, name, km in ((lq.WILSON, "Wilson", 0.141139), (lq.STAGGERED, "Staggered", MASS)):
        D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": km, "mass": km, "boundarycondition": BC})
        Xh, Yh = (orc.gaussian_spinor(lat.fermion_shape(kind), s) for s in (825, 826))
        X, Y = lq.Fermionfields(lat, kind).upload(Xh), lq.Fermionfields(lat, kind).upload(Yh)
        G = lq.Gaugefields(lat)
        lq.fermion_force_(G, D, X, Y)
        g1 = G.download()
        assert rel_err(g1, orc.fermion_force(kind, Uh, Xh, Yh, L, km, 1.0, BC)) < 1e-13
        lq.fermion_force_(G, D, X, Y, scale=0.25, accumulate=True)
        assert rel_err(G.download(), 1.25 * g1) < 1e-14
        lq.fermion_force_(G, D, X, Y, scale=-2.0)
        assert rel_err(G.download(), -2.0 * g1) < 1e-14


@pytest.mark.parametrize("nf", [2, 3])
def test_hmc_repeats_the_reference_general_nf_tests_on_device(lq, orc, nf):
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "staggered_nf%d_4x4x4x4.ildg" % nf), L)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    start = lq.calculate_Plaquette(U)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS, "boundarycondition": BC, "eps_CG": 1e-19})
    fa = lq.FermiAction(D, {"Nf": nf})
    p, G, Uold = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
    xi, phi = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
    dtau, mdsteps = 0.05, 20
    rng = np.random.default_rng(131 + nf)
    dHs, acc = [], 0
    for traj in range(10):
        lq.substitute_U_(Uold, U)
        lq.gauss_distribution_(p, 700 + traj)
        lq.gauss_sampling_in_action_(xi, U, fa, 800 + traj)
        lq.sample_pseudofermions_(phi, U, fa, xi)
        Hold = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.evaluate_FermiAction(fa, U, phi)
        for _ in range(mdsteps):                           # runMD_QPQ! (standardMD.jl:125-139)
            lq.U_update_(U, p, 0.5 * dtau)
            lq.gauge_force_(G, U, BETA)
            lq.Traceless_antihermitian_add_(p, dtau, G)
            lq.calc_UdSfdU_(G, fa, U, phi)
            lq.Traceless_antihermitian_add_(p, dtau, G)
            lq.U_update_(U, p, 0.5 * dtau)
        dH = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.evaluate_FermiAction(fa, U, phi) - Hold
        dHs.append(dH)
        if np.exp(-dH) >= rng.random():
            acc += 1
        else:
            lq.substitute_U_(U, Uold)
    plaq = lq.calculate_Plaquette(U)
    print("staggered Nf = %d RHMC: dH =" % nf, ["%.3f" % d for d in dHs], "accepted", acc, "/ 10, plaquette %.6f (start %.6f)" % (plaq, start))
    assert abs(plaq - REF_PLAQ[nf]) / REF_PLAQ[nf] < 0.1
    assert acc >= 5 and np.abs(dHs).max() < 3.0
    assert abs(plaq - start) > 1e-6 and orc.unitarity_dev(U.download(), L) < 1e-9


@pytest.mark.parametrize("dirac,csw", [("Wilson", 0.0), ("WilsonClover", 1.0)])
def test_wilson_one_flavour_rational_action(lq, orc, dirac, csw):
    """Wilson / Wilson-clover Nf = 1 (the odd flavour of a 2+1 run): S_f = phi^+ (D^+D)^(-1/2) phi on the interval estimated by the
    device Lanczos; action and force against the oracle's composition, heat-bath identity, and the interval encloses the spectrum."""
    L, kappa = (4, 4, 4, 4), 0.141139
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 851)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": dirac, "κ": kappa, "Clover_coefficient": csw, "boundarycondition": BC, "eps_CG": 1e-22})
    tmin, tmax = lq.estimate_spectrum(lq.DdagD_operator(D), steps=80)
    fa = lq.FermiAction(D, {"Nf": 1})
    lo, hi = fa.spectral_interval
    assert fa.rational and abs(fa.alpha - 0.5) < 1e-15 and lo < tmin < tmax < hi
    if csw == 0.0:      # exact extreme eigenvalues of the 3072 x 3072 matrix are affordable here through the oracle's operator
        import scipy.sparse.linalg as sla
        shape = lat.fermion_shape(lq.WILSON)
        n = int(np.prod(shape))
        mv = lambda v: orc.wilson_D(Uh, orc.wilson_D(Uh, np.ascontiguousarray(v.reshape(shape)), L, kappa, 1.0, BC), L, kappa, 1.0, BC, True).reshape(n)
        op = sla.LinearOperator((n, n), matvec=mv, dtype=np.complex128)
        emax = sla.eigsh(op, k=1, which="LA", return_eigenvectors=False, tol=1e-8)[0]
        emin = sla.eigsh(op, k=1, which="SA", return_eigenvectors=False, tol=1e-8)[0]
        assert lo < emin and emax < hi and tmin < 1.3 * emin and tmax > 0.95 * emax
    phih = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 852)
    phi = lq.Fermionfields(lat, lq.WILSON).upload(phih)
    S = lq.evaluate_FermiAction(fa, U, phi)
    G = lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G, fa, U, phi)
    if csw == 0.0:
        a0, res, poles = fa.rhmc_action
        yo, _ = orc.rational_apply(orc.WILSON, Uh, phih, L, kappa, a0, res, poles, 1.0, BC)
        assert abs(S - np.vdot(phih, yo).real) < 1e-10 * abs(S)
        Go = orc.rational_force(orc.WILSON, Uh, phih, L, kappa, fa.rhmc_MD[1], fa.rhmc_MD[2], 1.0, BC)
        assert rel_err(G.download(), Go) < 1e-9
    # heat bath: S_f((D'D)^(1/4) xi) = xi' xi   (both operators)
    xi = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_sampling_in_action_(xi, U, fa, 853)
    lq.sample_pseudofermions_(phi, U, fa, xi)
    assert abs(lq.evaluate_FermiAction(fa, U, phi) / lq.dot(xi, xi).real - 1.0) < 1e-8
    # the force is the derivative of the device action (central differences along one generator on one link)
    lq.calc_UdSfdU_(G, fa, U, phi)
    Gh = G.download()
    from scipy.linalg import expm
    rng = np.random.default_rng(854)
    T = sum(c * g for c, g in zip(rng.normal(size=8), orc.GELLMANN))
    mu, t, z, y, x = 2, 1, 3, 0, 2
    vals = []
    fa_acc = lq.FermiAction(D, {"Nf": 1, "rhmc_lambda_min": lo, "rhmc_lambda_max": hi, "rhmc_tol_action": 1e-8})   # the MD fit as the action
    fa_acc.rhmc_action = fa.rhmc_MD
    U2 = lq.Gaugefields(lat)
    for sgn in (+1, -1):
        Up = Uh.copy()
        Up[mu, t, z, y, x] = (expm(1j * sgn * 1e-4 * T) @ Uh[mu, t, z, y, x].T).T
        U2.upload(Up)
        vals.append(lq.evaluate_FermiAction(fa_acc, U2, phi))
    fa.D(U)
    fd = (vals[0] - vals[1]) / 2e-4
    an = -2.0 * np.trace(T @ Gh[mu, t, z, y, x].T).imag
    assert abs(fd - an) < 2e-6 * max(1.0, abs(an)), (fd, an)


def test_two_plus_one_flavour_wilson_clover_trajectory(lq, orc):
    """2+1 flavours of Wilson-clover quarks: a 2-flavour pseudofermion (S = eta'(D'D)^-1 eta, light kappa) and a rational 1-flavour one
    (S = phi'(D'D)^(-1/2) phi, heavier kappa) in one leapfrog trajectory -- second-order energy conservation and reversibility."""
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    par = {"Dirac_operator": "WilsonClover", "Clover_coefficient": 1.0, "boundarycondition": BC, "eps_CG": 1e-20}
    Dl = lq.Dirac_operator(U, None, dict(par, **{"κ": 0.141139}))
    Ds = lq.Dirac_operator(U, None, dict(par, **{"κ": 0.13}))
    fl, fs = lq.FermiAction(Dl), lq.FermiAction(Ds, {"Nf": 1, "rhmc_tol_MD": 1e-10})
    p, G = lq.Gaugefields(lat), lq.Gaugefields(lat)
    xil, xis = lq.Fermionfields(lat, lq.WILSON), lq.Fermionfields(lat, lq.WILSON)
    etal, etas = xil.similar(), xis.similar()

    def H():
        return (lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.evaluate_FermiAction(fl, U, etal)
                + lq.evaluate_FermiAction(fs, U, etas))

    def leapfrog(dtau, n):
        for _ in range(n):
            lq.U_update_(U, p, 0.5 * dtau)
            lq.P_update_(U, p, dtau, BETA)
            for fa, eta in ((fl, etal), (fs, etas)):
                lq.calc_UdSfdU_(G, fa, U, eta)
                lq.Traceless_antihermitian_add_(p, dtau, G)
            lq.U_update_(U, p, 0.5 * dtau)

    dH = []
    for n in (8, 16):
        U.upload(Uh)
        lq.gauss_distribution_(p, 861)
        lq.gauss_sampling_in_action_(xil, U, fl, 862)
        lq.gauss_sampling_in_action_(xis, U, fs, 863)
        lq.sample_pseudofermions_(etal, U, fl, xil)
        lq.sample_pseudofermions_(etas, U, fs, xis)
        H0 = H()
        assert abs(lq.evaluate_FermiAction(fs, U, etas) / lq.dot(xis, xis).real - 1.0) < 1e-8
        leapfrog(0.4 / n, n)
        dH.append(H() - H0)
    assert abs(dH[1]) < abs(dH[0]) < 3.0 and 3.0 < abs(dH[0] / dH[1]) < 5.0, dH
    P = p.download()
    p.upload(-P)
    leapfrog(0.4 / 16, 16)
    assert np.abs(U.download() - Uh).max() < 1e-8


def test_wilson_rational_interval_follows_the_links(lq, orc):
    """the fit interval of the Wilson Nf = 1 action is re-checked on the current links at every heat bath / action
    evaluation: an estimated interval is refitted when the spectrum has left it, a caller-fixed one raises."""
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    U_free = lq.Initialize_Gaugefields(3, 0, *L, condition="cold", lattice=lat)          # free field: lambda_min = (1 - 8 kappa)^2-ish, far above ...
    U_hot = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 871))                            # ... the hot-start spectrum's lower edge
    D = lq.Dirac_operator(U_hot, None, {"Dirac_operator": "Wilson", "κ": 0.125, "boundarycondition": BC, "eps_CG": 1e-20})
    t_hot = lq.estimate_spectrum(lq.DdagD_operator(D))
    t_free = lq.estimate_spectrum(lq.DdagD_operator(D(U_free)))
    D(U_hot)
    assert t_free[0] < 0.3 * t_hot[0] or t_free[1] > 1.3 * t_hot[1]                      # the two spectra really differ
    fa = lq.FermiAction(D, {"Nf": 1})
    lo0, hi0 = fa.spectral_interval
    phi = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(phi, 872)
    lq.evaluate_FermiAction(fa, U_hot, phi)
    assert fa.interval_refits == 0 and fa.spectral_interval == (lo0, hi0)
    S = lq.evaluate_FermiAction(fa, U_free, phi)                                         # spectrum left the interval: refit, then evaluate
    lo1, hi1 = fa.spectral_interval
    assert fa.interval_refits == 1 and lo1 <= 0.6 * t_free[0] and hi1 >= 1.1 * t_free[1]
    X = phi.similar()                                                                    # S_f = phi^+ (D^+D)^(-1/2) phi by an independent fit on the new interval
    lq.apply_inverse_power_(X, lq.DdagD_operator(D(U_free)), phi, 0.5, lo1, hi1, tol=1e-11)
    assert abs(lq.dot(phi, X).real / S - 1.0) < 1e-8
    fixed = lq.FermiAction(D(U_hot), {"Nf": 1, "rhmc_lambda_min": lo0, "rhmc_lambda_max": hi0})
    lq.evaluate_FermiAction(fixed, U_hot, phi)
    with pytest.raises(lq.LQCDError):
        lq.evaluate_FermiAction(fixed, U_free, phi)


@pytest.mark.parametrize("name,km,nf,alpha", [("Staggered", MASS, 2, 2 / 8), ("Staggered", MASS, 3, 3 / 8), ("Staggered", MASS, 1, 1 / 8), ("Wilson", 0.12, 1, 0.5)])
def test_action_handle_decides_and_fits_below_the_c_abi(lq, orc, name, km, nf, alpha):
    """FermiAction(D, Dict("Nf" => nf)) (universe.jl:106-110,138; test/test_Nf2.toml:8, test/test_Nf3.toml:8): the library alone -- no host-side
    numerics of the binding -- classifies the action, chooses the interval and produces coefficients that verify to their tolerance on it with
    the right signs; the three fits relate as x^(-alpha) (tight, loose) and x^(alpha/2 - 1)."""
    import ctypes as C
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 861))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": km, "mass": km, "boundarycondition": BC, "eps_CG": 1e-20})
    fa = lq.FermiAction(D, {"Nf": nf})
    assert fa.rational and not fa.evensite and abs(fa.alpha - alpha) < 1e-15 and fa.Nf == nf
    lo, hi = fa.spectral_interval
    if name == "Staggered":
        assert abs(lo / km ** 2 - 1) < 1e-8 and abs(hi / (km ** 2 + 16) - 1) < 1e-8
    else:
        # the lower edge is CERTIFIED: the Lanczos run continues until the residual bound |beta_k s_k| of the smallest Ritz value is below
        # 10 % of it, and the fit starts at half of (Ritz value - bound).  Checked against the exact spectrum of the oracle's dense D^+D.
        w = np.linalg.eigvalsh(orc.dense_DdagD(orc.WILSON, orc.hot_gauge(L, 861), L, km, 1.0, BC))
        bound, used = fa._get("ritz_bound"), int(fa._get("lanczos_steps_used"))
        theta = 2.0 * lo + bound                                    # the smallest Ritz value the edge was made from
        assert used >= 60 and 0.0 <= bound <= 0.1 * theta
        assert w[0] * (1 - 1e-9) <= theta and np.abs(w - theta).min() <= bound * (1 + 1e-6) + 1e-12      # a Ritz value lies above lambda_min, an eigenvalue within its bound
        assert lo <= 0.5 * w[0] * (1 + 1e-9) and 1.2 * w[-1] * (1 - 1e-6) <= hi <= 1.2 * w[-1] * 1.1
        tmin, tmax = lq.estimate_spectrum(lq.DdagD_operator(D))     # the plain 60-step estimate (no certificate) brackets from inside
        assert tmin >= w[0] * (1 - 1e-9) and tmax <= w[-1] * (1 + 1e-9)
    x = np.exp(np.linspace(np.log(lo), np.log(hi), 9001))
    for which, (power, tol) in enumerate(((alpha, 1e-12), (alpha, 1e-8), (1 - alpha / 2, 1e-12))):
        a0, res, poles = (fa.rhmc_action, fa.rhmc_MD, fa.rhmc_sampling)[which]
        err = C.c_double(0)
        n = C.c_int(0)
        lq.lib.check(lq.lib.lib().lqcd_action_coefficients(fa._fa, which, None, None, None, 0, C.byref(n), C.byref(err)))
        assert n.value == len(poles) and a0 >= 0 and (res > 0).all() and (poles > 0).all()
        measured = np.abs(lq.rational.evaluate(a0, res, poles, x) * x ** power - 1.0).max()
        assert measured <= max(tol, 10 * err.value) and err.value <= 1e-10 * (1 if which != 1 else 1e3), (which, measured, err.value)
    assert len(fa.rhmc_MD[1]) < len(fa.rhmc_action[1])
    # the same coefficients from the stand-alone export on the same interval
    a0, res, poles, _ = lq.rational.inverse_power_partial_fractions(alpha, lo, hi, 1e-12)
    assert len(poles) == len(fa.rhmc_action[1]) and np.allclose(poles, fa.rhmc_action[2], rtol=1e-12) and np.allclose(res, fa.rhmc_action[1], rtol=1e-12)
    fa.close()


def test_action_handle_exact_actions_and_refusals(lq, orc):
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 862))
    Ds = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS, "boundarycondition": BC})
    Dw = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.12, "boundarycondition": BC})
    for D, nf, even in ((Ds, 4, True), (Ds, 8, False), (Dw, 2, False)):
        fa = lq.FermiAction(D, {"Nf": nf})
        assert not fa.rational and fa.evensite == even and fa.Nf == nf
        fa.close()
    assert lq.FermiAction(Ds).Nf == 4 and lq.FermiAction(Dw).Nf == 2              # the reference's defaults when the Dict carries no Nf
    for D, nf in ((Ds, 9), (Ds, 8.5), (Dw, 3)):
        with pytest.raises(lq.LQCDError, match="outside"):
            lq.FermiAction(D, {"Nf": nf})
    with pytest.raises(lq.LQCDError, match="rhmc_lambda"):
        lq.FermiAction(Ds, {"Nf": 2, "rhmc_lambda_min": 3.0, "rhmc_lambda_max": 1.0})
    # an action follows later changes of the operator's stopping rule (D.eps_CG is a plain attribute of the binding)
    fa = lq.FermiAction(Ds, {"Nf": 8})
    eta = lq.Fermionfields(lat, lq.STAGGERED)
    lq.gauss_distribution_fermion_(eta, 863)
    Ds.MaxCGstep = 2
    with pytest.raises(lq.NotConverged):
        lq.evaluate_FermiAction(fa, U, eta)
    Ds.MaxCGstep = 3000
    assert lq.evaluate_FermiAction(fa, U, eta) > 0
