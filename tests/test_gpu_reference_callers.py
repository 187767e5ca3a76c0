"""The reference's UNCHANGED callers, transliterated line by line on top of the per-direction interface they really use
(U[mu], p[mu], one temporary link field at a time), must reproduce the fused four-direction trajectory of tests/test_gpu_md.py.

Transliterated (Julia `!` -> `_`, `μ = 1:Dim` kept 1-based, everything else verbatim):
  U_update!, P_update!, P_update_fermion!        /root/reference/src/md/AbstractMD.jl:78-135
  StandardMD, initialize_MD!, runMD_QPQ_sw!      /root/reference/src/md/standardMD.jl:5-166
  update!(::StandardHMC)                         /root/reference/src/updates/standardHMC.jl:41-91
  the construction of gauge_action / fermi_action  /root/reference/src/system/universe.jl:88-138
The only liberties: the random numbers (the device generators are counter based and take a seed; the accept test draws from
numpy) -- the reference's own RNG stream is not reproducible outside Julia either."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from test_gpu_md import BC, BETA, KAPPA, DeviceHMC

pytestmark = pytest.mark.gpu
Dim = 4


class StandardMD:
    """struct StandardMD + its constructor (standardMD.jl:5-80)."""

    def __init__(self, lq, U, gauge_action, quench, dtau, MDsteps, fermi_action=None, QPQ=True, SextonWeingargten=False, Nsw=2):
        self.lq = lq
        self.p = lq.initialize_TA_Gaugefields(U)      # p = initialize_TA_Gaugefields(U)
        if quench:
            self.eta = self.xi = None
            if SextonWeingargten:
                raise RuntimeError("The quench update does not need the SextonWeingargten method. Put SextonWeingargten = false")
        elif fermi_action is None:
            self.eta = self.xi = None
        else:
            self.eta = fermi_action._temporary_fermionfields[0].similar()    # η = similar(fermi_action._temporary_fermionfields[1])
            self.xi = self.eta.similar()                                      # ξ = similar(η)
        assert Nsw % 2 == 0, f"Nsw should be even number! now Nsw = {Nsw}"
        self.gauge_action, self.quench, self.dtau, self.MDsteps = gauge_action, quench, dtau, MDsteps
        self.QPQ, self.fermi_action, self.SextonWeingargten, self.Nsw = QPQ, fermi_action, SextonWeingargten, Nsw
        self.seed = 0


def U_update_(U, p, eps, md):                                   # AbstractMD.jl:78-98
    lq = md.lq
    temps = lq.get_temporary_gaugefields(md.gauge_action)
    temp1, it_temp1 = lq.get_temp(temps)
    temp2, it_temp2 = lq.get_temp(temps)
    expU, it_expU = lq.get_temp(temps)
    W, it_W = lq.get_temp(temps)
    for mu in range(1, Dim + 1):
        lq.exptU_(expU, eps * md.dtau, p[mu], [temp1, temp2])
        lq.mul_(W, expU, U[mu])
        lq.substitute_U_(U[mu], W)
    lq.unused_(temps, it_temp1)
    lq.unused_(temps, it_temp2)
    lq.unused_(temps, it_expU)
    lq.unused_(temps, it_W)


def P_update_(U, p, eps, md):                                   # AbstractMD.jl:100-118   p -> p + factor*U*dSdUμ
    lq = md.lq
    NC = U[1].NC
    temps = lq.get_temporary_gaugefields(md.gauge_action)
    dSdUmu, its_dSdUmu = lq.get_temp(temps)
    factor = -eps * md.dtau / NC
    temp1, it_temp1 = lq.get_temp(temps)
    for mu in range(1, Dim + 1):
        lq.calc_dSdUmu_(dSdUmu, md.gauge_action, mu, U)
        lq.mul_(temp1, U[mu], dSdUmu)                          # U*dSdUμ
        lq.Traceless_antihermitian_add_(p[mu], factor, temp1)
    lq.unused_(temps, its_dSdUmu)
    lq.unused_(temps, it_temp1)


def P_update_fermion_(U, p, eps, md):                           # AbstractMD.jl:120-135
    lq = md.lq
    temps = lq.get_temporary_gaugefields(md.gauge_action)
    UdSfdUmu, its_UdSfdUmu = lq.get_temp(temps, Dim)
    factor = -eps * md.dtau
    lq.calc_UdSfdU_(UdSfdUmu, md.fermi_action, U, md.eta)
    for mu in range(1, Dim + 1):
        lq.Traceless_antihermitian_add_(p[mu], factor, UdSfdUmu[mu - 1])
    lq.unused_(temps, its_UdSfdUmu)


def initialize_MD_(U, md):                                      # standardMD.jl:82-101
    lq = md.lq
    md.seed += 3
    lq.gauss_distribution_(md.p, md.seed)                       # gauss_distribution!(md.p)  #initial momentum
    if not md.quench:
        lq.gauss_sampling_in_action_(md.xi, U, md.fermi_action, md.seed + 1)
        lq.sample_pseudofermions_(md.eta, U, md.fermi_action, md.xi)


def runMD_QPQ_sw_(U, md):                                       # standardMD.jl:146-166
    p = md.p
    for itrj in range(md.MDsteps):
        for isw in range(md.Nsw // 2):
            U_update_(U, p, 0.5 / md.Nsw, md)
            P_update_(U, p, 1.0 / md.Nsw, md)
            U_update_(U, p, 0.5 / md.Nsw, md)
        if not md.quench:
            P_update_fermion_(U, p, 1.0, md)
        for isw in range(md.Nsw // 2):
            U_update_(U, p, 0.5 / md.Nsw, md)
            P_update_(U, p, 1.0 / md.Nsw, md)
            U_update_(U, p, 0.5 / md.Nsw, md)


def runMD_QPQ_(U, md):                                          # standardMD.jl:127-144
    p = md.p
    for itrj in range(md.MDsteps):
        U_update_(U, p, 0.5, md)
        P_update_(U, p, 1.0, md)
        if not md.quench:
            P_update_fermion_(U, p, 1.0, md)
        U_update_(U, p, 0.5, md)


def runMD_(U, md):                                              # standardMD.jl:103-125
    if md.QPQ:
        if md.SextonWeingargten:
            runMD_QPQ_sw_(U, md)
        else:
            runMD_QPQ_(U, md)
    else:
        raise RuntimeError("PQP update is not transliterated")


class StandardHMC:                                              # standardHMC.jl:1-38
    def __init__(self, lq, U, gauge_action, quench, dtau, MDsteps, fermi_action, SextonWeingargten=False, QPQ=True, Nsw=2, seed=0):
        self.md = StandardMD(lq, U, gauge_action, quench, dtau, MDsteps, fermi_action, QPQ=QPQ, SextonWeingargten=SextonWeingargten, Nsw=Nsw)
        self.md.seed = seed
        self.Uold = U.similar()
        self.rng = np.random.default_rng(seed)
        self.dH = []


def update_(updatemethod, U):                                   # standardHMC.jl:41-91
    md = updatemethod.md
    lq = md.lq
    NC = U[1].NC
    Uold = updatemethod.Uold
    lq.substitute_U_(Uold, U)                                   # previous configuration
    initialize_MD_(U, md)
    Sp = md.p * md.p / 2
    Sg = -lq.evaluate_GaugeAction(md.gauge_action, U) / NC
    Sold = Sp + Sg
    if not md.quench:
        Sfold = lq.dot(md.xi, md.xi).real
        Sold += Sfold
    runMD_(U, md)
    Sp = md.p * md.p / 2
    Sg = -lq.evaluate_GaugeAction(md.gauge_action, U) / NC
    Snew = Sp + Sg
    if not md.quench:
        Sfnew = lq.evaluate_FermiAction(md.fermi_action, U, md.eta)
        Snew += Sfnew
    updatemethod.dH.append(Snew - Sold)
    accept = np.exp(Sold - Snew) >= updatemethod.rng.random()
    if not accept:
        lq.substitute_U_(U, Uold)                               # back to previous configuration
    return accept


def _universe(lq, U, kappa, beta):
    """universe.jl:88-138: gauge_action = GaugeAction(U); push!(gauge_action, beta/2, plaqloop + plaqloop'); D; FermiAction(D, Dict())."""
    gauge_action = lq.GaugeAction(U)
    plaqloop = lq.make_loops_fromname("plaquette", Dim=Dim)
    plaqloop = plaqloop + lq.make_loops_fromname("plaquette", Dim=Dim, adjoint=True)      # append!(plaqloop, plaqloop')
    gauge_action.push_(beta / 2, plaqloop)
    x = lq.Initialize_pseudofermion_fields(U[1], "Wilson", nowing=True)
    params = {"Dirac_operator": "Wilson", "κ": kappa, "r": 1.0, "faster version": True, "eps_CG": 1e-19, "verbose_level": 2,
              "MaxCGstep": 3000, "boundarycondition": BC}
    D = lq.Dirac_operator(U, x, params)
    fermi_action = lq.FermiAction(D, {})
    return gauge_action, fermi_action


def _fixture(lq):
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    return L, Uh


def test_single_direction_entry_points_match_the_fused_ones(lq, orc):
    """P_update! / U_update! written with U[mu], p[mu] and temporaries (5 single-direction C entry points) = the fused kernels."""
    L = (8, 4, 6, 4)
    lat = lq.Lattice(L)
    Uh, Ph = orc.hot_gauge(L, 31), orc.gaussian_momenta(L, 32)
    U1, P1 = lq.Gaugefields(lat).upload(Uh), lq.Gaugefields(lat).upload(Ph)
    U2, P2 = lq.Gaugefields(lat).upload(Uh), lq.Gaugefields(lat).upload(Ph)
    ga = lq.GaugeAction(U1)
    ga.push_(BETA / 2, lq.make_loops_fromname("plaquette") + lq.make_loops_fromname("plaquette", adjoint=True))

    class MD:
        pass
    md = MD()
    md.lq, md.gauge_action, md.dtau = lq, ga, 0.05
    P_update_(U1, P1, 0.3, md)
    lq.P_update_(U2, P2, 0.3 * 0.05, BETA)
    assert rel_err(P1.download(), P2.download()) < 1e-14
    U_update_(U1, P1, 0.7, md)
    lq.U_update_(U2, P2, 0.7 * 0.05)
    assert rel_err(U1.download(), U2.download()) < 1e-14
    # staple alone against the oracle's force:  G = -(beta/6) U * A  and  dSdUmu = (beta/2) A
    G = lq.Gaugefields(lat)
    lq.gauge_force_(G, U2, BETA)
    Gh, U2h = G.download(), U2.download()
    temps = lq.get_temporary_gaugefields(ga)
    dS, it = lq.get_temp(temps)
    T, it2 = lq.get_temp(temps)
    for mu in range(1, 5):
        lq.calc_dSdUmu_(dS, ga, mu, U2)
        lq.mul_(T, U2[mu], dS)
        assert rel_err(-T.download() / 3.0, Gh[mu - 1]) < 1e-13
    lq.unused_(temps, [it, it2])
    assert abs(-lq.evaluate_GaugeAction(ga, U2) / 3 - lq.evaluate_GaugeAction(U2, BETA)) < 1e-9
    assert abs(P1 * P1 / 2 - lq.momentum_action(P1)) < 1e-9


@pytest.mark.parametrize("reunit,lazy", [(1, 1), (0, 1), (0, 0)])
def test_transliterated_reference_callers_reproduce_the_fused_trajectory(lq, reunit, lazy):
    """update!(::StandardHMC) exactly as the reference wrote it, on U[mu] / p[mu], against DeviceHMC (fused kernels): same seeds,
    same momenta and noise, so the trajectories must agree to rounding (1e-12 on the links, 1e-9 on dH).  md_reunitarize = 0 is the reference's
    LITERAL link update exp(t p) U (no projection anywhere), lazy_links = 0 its literal call sequence (every generic its own kernel): the defaults
    (projection of on-group links inside the update pass, fused triples) are optimisations of THIS path and must not be the only one covered."""
    L, Uh = _fixture(lq)
    dtau, mdsteps, nsw, seed = 0.05, 20, 10, 1234
    lata, latb = lq.Lattice(L), lq.Lattice(L)
    for lat in (lata, latb):
        lat.set_param("md_reunitarize", reunit)
    latb.set_param("lazy_links", lazy)
    Ua = lq.Gaugefields(lata).upload(Uh)
    Ub = lq.Gaugefields(latb).upload(Uh)
    fused = DeviceHMC(lq, Ua, KAPPA, BETA, dtau, mdsteps, nsw, seed)
    gauge_action, fermi_action = _universe(lq, Ub, KAPPA, BETA)
    hmc = StandardHMC(lq, Ub, gauge_action, False, dtau, mdsteps, fermi_action, SextonWeingargten=True, Nsw=nsw, seed=seed)
    for traj in range(2):
        acc_a = fused.update()
        acc_b = update_(hmc, Ub)
        assert acc_a == acc_b
        assert abs(fused.dH[-1] - hmc.dH[-1]) < 1e-8, (fused.dH[-1], hmc.dH[-1])
        assert rel_err(Ub.download(), Ua.download()) < 1e-12
    assert abs(lq.calculate_Plaquette(Ua) - lq.calculate_Plaquette(Ub)) < 1e-13


def test_lazy_per_direction_triples_equal_the_eager_calls(lq, orc):
    """The bindings evaluate the reference's per-direction call triples lazily -- exptU! -> mul! -> substitute_U! becomes ONE lqcd_link_exp_mul,
    calc_dSdUmu! -> mul! -> Traceless_antihermitian_add! ONE lqcd_link_add_ta_staple -- with the callers unchanged.  Same links and momenta as the
    eager single-direction calls (rounding of the fused passes: 1e-14), also on a configuration that is not on the group (no projection there),
    and a temporary that IS read in the middle of a triple holds what the eager call would have put there."""
    L = (4, 4, 6, 8)
    Uh = orc.hot_gauge(L, 31)
    res = {}
    for lazy in (True, False):
        lat = lq.Lattice(L)
        lat.lazy_links = lazy
        U = lq.Gaugefields(lat).upload(Uh)
        ga = lq.GaugeAction(U)
        pl = lq.make_loops_fromname("plaquette", Dim=Dim)
        ga.push_(5.7 / 2, pl + lq.make_loops_fromname("plaquette", Dim=Dim, adjoint=True))
        md = StandardMD(lq, U, ga, True, 0.05, 20)
        lq.gauss_distribution_(md.p, 33)
        for _ in range(3):
            U_update_(U, md.p, 0.5, md)
            P_update_(U, md.p, 1.0, md)
            U_update_(U, md.p, 0.5, md)
        assert lat._lazy is None
        res[lazy] = (U.download(), md.p.download(), lq.unitarity_deviation(U))
        assert not lat._done
    assert np.abs(res[True][0] - res[False][0]).max() < 1e-13 and np.abs(res[True][1] - res[False][1]).max() < 1e-12
    assert res[True][2] == 0.0 and res[False][2] > 0.0          # the fused in-place update projects links that are on the group (md_reunitarize)
    # a triple that is interrupted: the temporary is materialised exactly as the eager call leaves it
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    p = lq.initialize_TA_Gaugefields(U)
    lq.gauss_distribution_(p, 34)
    tmp, tmp2 = lq.Gaugefields(lat), lq.Gaugefields(lat)
    lq.exptU_(tmp[1], 0.3, p[2])
    assert lat._lazy is not None
    lazy_val = tmp[1].download()                      # reading flushes
    assert lat._lazy is None
    lat.lazy_links = False
    lq.exptU_(tmp2[1], 0.3, p[2])
    assert np.array_equal(lazy_val, tmp2[1].download())
    lat.lazy_links = True
    lq.exptU_(tmp[1], 0.3, p[2])
    lq.mul_(tmp[2], tmp[1], U[3])
    lq.substitute_U_(tmp2[4], tmp[2])                 # not the in-place pattern: materialised, then copied
    # three directions of an update, then something else: the deferred ones run one by one and give what four single calls give
    Ua, Ub = lq.Gaugefields(lat), lq.Gaugefields(lat)
    lq.substitute_U_(Ua, U); lq.substitute_U_(Ub, U)
    for mu in (1, 2, 4):
        lq.exptU_(tmp[1], 0.2, p[mu]); lq.mul_(tmp[2], tmp[1], Ua[mu]); lq.substitute_U_(Ua[mu], tmp[2])
    assert len(lat._done) == 3
    got = Ua.download()
    assert not lat._done
    lat.lazy_links = False
    for mu in (1, 2, 4):
        lq.exptU_(tmp[1], 0.2, p[mu]); lq.mul_(tmp[2], tmp[1], Ub[mu]); lq.substitute_U_(Ub[mu], tmp[2])
    assert np.abs(got - Ub.download()).max() < 1e-13
    lat.lazy_links = True
    lat.lazy_links = False
    lq.exptU_(tmp[3], 0.3, p[2])
    lq.mul_(tmp[4], tmp[3], U[3])
    assert np.array_equal(tmp2[4].download(), tmp[4].download())


def test_operator_application_sees_deferred_link_updates(lq, orc):
    """Three directions of a per-direction link update are deferred (waiting for a fourth that would make them one fused call); applying the Dirac
    operator asks for its handle, which runs them first: the result is the one of the eager calls."""
    L = (4, 4, 4, 8)
    Uh = orc.hot_gauge(L, 71)
    out = []
    for lazy in (True, False):
        lat = lq.Lattice(L)
        lat.lazy_links = lazy
        U = lq.Gaugefields(lat).upload(Uh)
        p = lq.initialize_TA_Gaugefields(U)
        lq.gauss_distribution_(p, 72)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.12})
        x = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(x, 73)
        y = x.similar()
        tmp = lq.Gaugefields(lat)
        for mu in (1, 3, 4):
            lq.exptU_(tmp[1], 0.1, p[mu]); lq.mul_(tmp[2], tmp[1], U[mu]); lq.substitute_U_(U[mu], tmp[2])
        assert len(lat._done) == (3 if lazy else 0)
        lq.mul_(y, D, x)
        assert not lat._done
        out.append(y.download())
    assert np.abs(out[0] - out[1]).max() < 1e-13 * np.abs(out[1]).max()


def test_back_to_back_link_updates_merge(lq, orc):
    """runMD_QPQ_sw! (standardMD.jl:146-166) ends one Sexton-Weingarten block with U_update!(0.5/N) and starts the next with the same call, the momenta
    untouched in between: the library lets a complete link update wait and adds the next step to it (tunable lazy_merge) -- exp(b P) exp(a P) = exp((a + b) P).
    Same links and momenta as the eager sequence; whatever reads U or writes P in between runs the waiting update first."""
    L = (4, 4, 4, 8)
    Uh = orc.hot_gauge(L, 81)
    beta, dtau, nsw = 5.7, 0.05, 4
    res = {}
    for merge in (2, 1, 0):
        lat = lq.Lattice(L)
        lat.set_param("lazy_merge", merge)
        U = lq.Gaugefields(lat).upload(Uh)
        p = lq.initialize_TA_Gaugefields(U)
        lq.gauss_distribution_(p, 82)
        seen = []
        for _ in range(2):
            for _ in range(nsw // 2):
                lq.U_update_(U, p, 0.5 / nsw * dtau)
                seen.append(lat.get_param("lazy_deferred"))
                lq.P_update_(U, p, -dtau / nsw, beta)
                seen.append(lat.get_param("lazy_deferred"))
                lq.U_update_(U, p, 0.5 / nsw * dtau)
                seen.append(lat.get_param("lazy_deferred"))
        # 2: the momentum update waits too, and runs with the (merged) link update behind it as one sweep; 1: only link updates wait; 0: nothing waits
        assert seen == {2: [4, 4, 8] + [8, 4, 8] * (nsw - 1), 1: [4, 0, 4] * nsw, 0: [0, 0, 0] * nsw}[merge]
        lq.calculate_Plaquette(U)      # reads the links: what waits runs
        assert lat.get_param("lazy_deferred") == 0
        res[merge] = (U.download(), p.download(), lq.unitarity_deviation(U))
    for merge in (2, 1):
        assert np.abs(res[merge][0] - res[0][0]).max() < 1e-13 and np.abs(res[merge][1] - res[0][1]).max() < 1e-12
        assert res[merge][2] == 0.0      # projected in the same sweep (md_reunitarize), as the separate link update does
    # a step forward and the same step back before anything looks: the links do not move at all
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    p = lq.initialize_TA_Gaugefields(U)
    lq.gauss_distribution_(p, 83)
    lat.set_param("md_reunitarize", 0)
    lq.U_update_(U, p, 0.37)
    lq.U_update_(U, p, -0.37)
    assert np.array_equal(U.download(), Uh)
    # the momentum field changes between two updates: no merge across that
    out = []
    for merge in (2, 0):
        lat = lq.Lattice(L)
        lat.set_param("lazy_merge", merge)
        U = lq.Gaugefields(lat).upload(Uh)
        p = lq.initialize_TA_Gaugefields(U)
        lq.gauss_distribution_(p, 84)
        lq.U_update_(U, p, 0.1)
        lq.gauss_distribution_(p, 85)
        lq.U_update_(U, p, 0.1)
        out.append(U.download())
    assert np.abs(out[0] - out[1]).max() < 1e-14


def test_fused_momentum_and_link_update_on_reference_format_links(lq, orc):
    """The one-sweep P_update! + U_update! (lazy_merge = 2) on links that are NOT on the group to 1e-13 (what the reference's text files hold): nothing is
    projected, all three rows are read, and the result is the one of the two separate passes; then the operator built on the field sees the swapped buffer."""
    L = (4, 4, 8, 8)
    Uh = orc.hot_gauge(L, 91)
    rng = np.random.default_rng(92)
    Uh = Uh + 1e-10 * (rng.standard_normal(Uh.shape) + 1j * rng.standard_normal(Uh.shape))
    out = []
    for merge in (2, 0):
        lat = lq.Lattice(L)
        lat.set_param("lazy_merge", merge)
        U = lq.Gaugefields(lat).upload(Uh)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.12})
        x = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(x, 93)
        y = x.similar()
        lq.mul_(y, D, x)                         # the operator has seen the field's first buffer
        p = lq.initialize_TA_Gaugefields(U)
        lq.gauss_distribution_(p, 94)
        for _ in range(3):
            lq.P_update_(U, p, -0.01, 5.7)
            lq.U_update_(U, p, 0.02)
        lq.mul_(y, D, x)
        out.append((U.download(), p.download(), y.download(), lq.unitarity_deviation(U)))
    assert np.abs(out[0][0] - out[1][0]).max() < 1e-13 and np.abs(out[0][1] - out[1][1]).max() < 1e-12
    assert np.abs(out[0][2] - out[1][2]).max() < 1e-12 * np.abs(out[1][2]).max()
    assert out[0][3] > 1e-11 and abs(out[0][3] - out[1][3]) < 1e-13      # left off the group exactly as the literal update leaves them
    yo = orc.wilson_D(out[1][0], x.download(), L, 0.12, 1.0, (1, 1, 1, -1))
    assert np.abs(out[0][2] - yo).max() < 1e-13 * np.abs(yo).max()
