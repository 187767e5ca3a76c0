"""The reference's UNCHANGED callers on the device-backed field types.  Their call sequences are DATA here -- tests/golden/ref_exec_traces.json: what runs of
U_update!, P_update!, both P_update_fermion! methods, initialize_MD!, runMD! (QPQ / QPQ_sw / PQP) and update! against a recording binding EMITTED in the build
container (tests/refgen/record_traces.py executes the reference's functions; loops unrolled, branches on parameters resolved, locals replaced by slots) -- and
tests/ref_trace.py replays them against the binding, one binding function per entry (U[mu], p[mu], one temporary link field at a time, exactly the
interface the callers use).  What comes out is compared with the CPU ORACLE's trajectory from the same momenta and pseudofermion (tests/oracle_md.py on
oracle/oracle.py's primitives), not with another device path.
Reference sites: /root/reference/src/md/AbstractMD.jl:78-135, src/md/standardMD.jl:82-227, src/updates/standardHMC.jl:41-91, src/system/universe.jl:88-138."""
import os

import numpy as np
import pytest

import oracle_md
from conftest import GOLDEN, rel_err
from ref_trace import Raised, Replay, standard_hmc, standard_md
from test_gpu_md import BC, BETA, KAPPA

pytestmark = pytest.mark.gpu
Dim = 4
L4 = (4, 4, 4, 4)


def plaquette_action(lq, U, beta):
    """The gauge action the reference's runs build (universe.jl:88-96): the plaquette loops and their adjoints with coefficient beta / 2."""
    ga = lq.GaugeAction(U)
    ga.push_(beta / 2, lq.make_loops_fromname("plaquette", Dim=Dim) + lq.make_loops_fromname("plaquette", Dim=Dim, adjoint=True))
    return ga


def wilson_action(lq, U, kappa, eps=1e-19):
    """Two flavours of Wilson fermions with the parameter dictionary of universe.jl:103-138."""
    x = lq.Initialize_pseudofermion_fields(U[1], "Wilson", nowing=True)
    D = lq.Dirac_operator(U, x, {"Dirac_operator": "Wilson", "κ": kappa, "r": 1.0, "faster version": True, "eps_CG": eps, "verbose_level": 2,
                                 "MaxCGstep": 3000, "boundarycondition": BC})
    return lq.FermiAction(D, {})


def _fixture(lq):
    return lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L4)


def test_replayed_momentum_and_link_updates_match_the_oracle(lq, orc):
    """P_update! / U_update! replayed from the trace (per direction: calc_dSdUμ! -> mul! -> Traceless_antihermitian_add!, exptU! -> mul! -> substitute_U!)
    against the oracle's gauge_force / momentum_add_ta / link_update, and against the fused four-direction entry points."""
    L = (8, 4, 6, 4)
    lat = lq.Lattice(L)
    Uh, Ph = orc.hot_gauge(L, 31), orc.gaussian_momenta(L, 32)
    U1, P1 = lq.Gaugefields(lat).upload(Uh), lq.Gaugefields(lat).upload(Ph)
    U2, P2 = lq.Gaugefields(lat).upload(Uh), lq.Gaugefields(lat).upload(Ph)
    ga = plaquette_action(lq, U1, BETA)
    md = standard_md(lq, U1, ga, 0.05, 20)
    rp = Replay(lq)
    rp.call("P_update!", U1, P1, 0.3, md)
    assert [n for n in rp.log if n.endswith("!")].count("calc_dSdUμ!") == Dim
    Po = orc.momentum_add_ta(Ph.copy(), 0.3 * 0.05, orc.gauge_force(Uh, L, BETA), L)
    assert rel_err(P1.download(), Po) < 1e-13
    lq.P_update_(U2, P2, 0.3 * 0.05, BETA)
    assert rel_err(P1.download(), P2.download()) < 1e-14
    rp.call("U_update!", U1, P1, 0.7, md)
    Uo = orc.link_update(Uh.copy(), Po, 0.7 * 0.05, L)
    assert rel_err(U1.download(), Uo) < 1e-13
    lq.U_update_(U2, P2, 0.7 * 0.05)
    assert rel_err(U1.download(), U2.download()) < 1e-14
    # the staple generic on its own:  U[mu] * dSdUmu = -3 G_mu with the oracle's G = -(beta/6) U A
    Go = orc.gauge_force(Uo, L, BETA)
    temps = lq.get_temporary_gaugefields(ga)
    dS, it = lq.get_temp(temps)
    T, it2 = lq.get_temp(temps)
    for mu in range(1, 5):
        lq.calc_dSdUmu_(dS, ga, mu, U1)
        lq.mul_(T, U1[mu], dS)
        assert rel_err(-T.download() / 3.0, Go[mu - 1]) < 1e-13
    lq.unused_(temps, [it, it2])
    assert abs(-lq.evaluate_GaugeAction(ga, U1) / 3 - orc.gauge_action(Uo, L, BETA)) < 1e-9
    assert abs(P1 * P1 / 2 - orc.momentum_action(Po, L)) < 1e-9


SCHEMES = {"QPQ_sw": dict(QPQ=True, SextonWeingargten=True, Nsw=10), "QPQ": dict(QPQ=True, SextonWeingargten=False), "PQP": dict(QPQ=False, SextonWeingargten=False)}


@pytest.mark.parametrize("scheme,quench,reunit,lazy", [("QPQ_sw", False, 1, 1), ("QPQ_sw", False, 0, 1), ("QPQ_sw", False, 0, 0), ("QPQ", False, 1, 1), ("PQP", False, 1, 1),
                                                       ("QPQ", True, 1, 1), ("PQP", True, 0, 0)])
def test_replayed_update_reproduces_the_oracle_trajectory(lq, orc, scheme, quench, reunit, lazy):
    """update!(::StandardHMC) replayed entry by entry from the trace its run emitted, started from its thermalised Wilson configuration (test/test_wilson.toml: beta 5.7,
    kappa 0.141139, dtau 0.05, 20 MD steps, Sexton-Weingarten N = 10), against the oracle's trajectory from the momenta and the pseudofermion the device drew:
    links to 1e-9, momenta to 1e-8, dH to 1e-6, the same accept decision.  md_reunitarize = 0 is the reference's literal link update exp(t p) U (no projection),
    lazy_links = 0 its literal call sequence (every generic its own kernel): the defaults are optimisations of THIS path and must not be the only one covered.
    All three integrators of runMD! (standardMD.jl:103-190), quenched and dynamical."""
    Uh = _fixture(lq)
    dtau, mdsteps = 0.05, 20
    lat = lq.Lattice(L4)
    lat.set_param("md_reunitarize", reunit)
    lat.set_param("lazy_links", lazy)
    U = lq.Gaugefields(lat).upload(Uh)
    ga = plaquette_action(lq, U, BETA)
    fa = None if quench else wilson_action(lq, U, KAPPA)
    md = standard_md(lq, U, ga, dtau, mdsteps, fermi_action=fa, **SCHEMES[scheme])
    hmc = standard_hmc(lq, U, md)
    seen = {}

    def drawn(rp):       # the random fields the device drew: the oracle starts from the same ones
        seen["P"] = md["p"].download()
        seen["eta"] = None if quench else md["η"].download()
        seen["xi2"] = 0.0 if quench else lq.dot(md["ξ"], md["ξ"]).real

    def evolved(rp):
        seen["U1"], seen["P1"] = U.download(), md["p"].download()

    rp = Replay(lq, seed=1234, hooks={("after", "initialize_MD!"): drawn, ("after", "runMD!"): evolved,
                                      ("after", "update!"): lambda rp: seen.update(dH=rp.watch("H_new") - rp.watch("H_old"))})
    for traj in range(1 if scheme != "QPQ_sw" or reunit == 0 else 2):
        U0 = U.download()
        accepted = rp.call("update!", hmc, U)
        ferm = None if quench else (orc.WILSON, KAPPA, BC, seen["eta"], 1e-19)
        Uo, Po = oracle_md.trajectory(orc, U0, seen["P"], L4, BETA, dtau, mdsteps, scheme, 10, ferm)
        assert rel_err(seen["U1"], Uo) < 1e-9 and rel_err(seen["P1"], Po) < 1e-8, (scheme, traj)
        Hold = orc.momentum_action(seen["P"], L4) + orc.gauge_action(U0, L4, BETA) + seen["xi2"]
        dH = oracle_md.hamiltonian(orc, Uo, Po, L4, BETA, ferm) - Hold
        assert abs(dH) < 2.0 and abs(seen["dH"] - dH) < 1e-6, (seen["dH"], dH)      # the device's Snew - Sold is the oracle's
        # the accept decision the replay took is the one this dH and the same uniform deviate give; a rejected trajectory restores the links bit for bit
        u = np.random.default_rng(1234).random(traj + 1)[-1]
        assert accepted == bool(np.exp(-dH) >= u) or abs(np.exp(-dH) - u) < 1e-5
        assert np.array_equal(U.download(), seen["U1"] if accepted else U0)
    if not quench:
        assert rp.log.count("calc_UdSfdU!") == mdsteps * (2 if scheme == "PQP" else 1) * (traj + 1)


def test_runMD_refuses_what_the_reference_refuses(lq, orc):
    """runMD! (standardMD.jl:103-124): PQP with Sexton-Weingarten raises in the reference -- and in the replay, before any field is touched."""
    lat = lq.Lattice(L4)
    U = lq.Gaugefields(lat).upload(_fixture(lq))
    md = standard_md(lq, U, plaquette_action(lq, U, BETA), 0.05, 2, QPQ=False, SextonWeingargten=True)
    before = U.download()
    with pytest.raises(Raised):
        Replay(lq).call("runMD!", U, md)
    assert np.array_equal(U.download(), before)


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
        md = standard_md(lq, U, plaquette_action(lq, U, 5.7), 0.05, 20)
        lq.gauss_distribution_(md.p, 33)
        rp = Replay(lq)
        for _ in range(3):
            rp.call("U_update!", U, md.p, 0.5, md)
            rp.call("P_update!", U, md.p, 1.0, md)
            rp.call("U_update!", U, md.p, 0.5, md)
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
