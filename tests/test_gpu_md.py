"""GPU parity of the gauge side of the MD step against the oracle, and an end-to-end HMC run on the
device that repeats the reference's own test (test/runtests.jl:88-99 with test/test_wilson.toml): start from the reference's
thermalised 4^4 configuration, beta = 5.7, kappa = 0.141139, dtau = 0.05, 20 MD steps, Sexton-Weingarten with N = 10, ten
trajectories; the final plaquette must lie within 10 % of the reference's recorded value (test/debugplaqdata.txt line 7)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err
from oracle_md import stages

pytestmark = pytest.mark.gpu

KAPPA, BETA = 0.141139, 5.7
BC = (1, 1, 1, -1)
REF_PLAQ_WILSON_HMC = 0.5784043949012552          # /root/reference/test/debugplaqdata.txt:7 (plaqvalues[7], runtests.jl:94)


@pytest.fixture(scope="module")
def gpu(lq):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return lq


@pytest.mark.parametrize("L", [(4, 4, 4, 4), (8, 4, 6, 2)])
def test_md_kernels_match_oracle(gpu, orc, L):
    lq = gpu
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 301)
    Ph = orc.gaussian_momenta(L, 302)
    U, P, G = lq.Gaugefields(lat).upload(Uh), lq.Gaugefields(lat).upload(Ph), lq.Gaugefields(lat)
    assert abs(lq.evaluate_GaugeAction(U, BETA) - orc.gauge_action(Uh, L, BETA)) < 1e-10
    assert abs(lq.momentum_action(P) - orc.momentum_action(Ph, L)) < 1e-10 * orc.momentum_action(Ph, L)
    lq.gauge_force_(G, U, BETA)
    Gh = orc.gauge_force(Uh, L, BETA)
    assert rel_err(G.download(), Gh) < 1e-13
    P2 = lq.Gaugefields(lat).upload(Ph)
    lq.Traceless_antihermitian_add_(P, -0.37, G)
    orc.momentum_add_ta(Ph, -0.37, Gh, L)
    assert rel_err(P.download(), Ph) < 1e-13
    lq.P_update_(U, P2, -0.37, BETA)                  # fused force + projection + accumulate
    assert rel_err(P2.download(), Ph) < 1e-13
    lq.U_update_(U, P, 0.11)
    orc.link_update(Uh, Ph, 0.11, L)
    assert rel_err(U.download(), Uh) < 1e-13
    assert orc.unitarity_dev(U.download(), L) < 1e-14
    # Uold <- U
    Uold = lq.Gaugefields(lat)
    lq.substitute_U_(Uold, U)
    assert np.array_equal(Uold.download(), U.download())
    with pytest.raises(lq.LQCDError):
        lq.gauge_force_(U, U, BETA)


def test_momentum_sampling(gpu, orc):
    lq = gpu
    L = (8, 8, 8, 8)
    lat = lq.Lattice(L)
    P = lq.Gaugefields(lat)
    lq.gauss_distribution_(P, 7)
    Ph = np.swapaxes(P.download(), -1, -2)           # [.., a, b]
    assert np.abs(Ph + Ph.conj().swapaxes(-1, -2)).max() < 1e-15                       # anti-Hermitian
    assert np.abs(np.trace(Ph, axis1=-2, axis2=-1)).max() < 1e-15                      # traceless
    ndof = 4 * 8 ** 4 * 8
    assert abs(lq.momentum_action(P) / ndof - 0.5) < 0.01                              # <pi_a^2>/2 = 1/2 per generator
    pi = 2.0 * np.einsum("...ij,aji->...a", Ph, orc.GELLMANN / 2).imag                 # P = i pi_a T_a  =>  pi_a = 2 Im tr(P T_a)
    assert abs(pi.mean()) < 0.01 and abs(pi.std() - 1.0) < 0.01
    P2 = lq.Gaugefields(lat)
    lq.gauss_distribution_(P2, 7)
    assert np.array_equal(P2.download(), P.download())
    lq.gauss_distribution_(P2, 8)
    assert not np.array_equal(P2.download(), P.download())


class DeviceHMC:
    """A Sexton-Weingarten HMC on the fused entry points, every field resident: what the reference's update!(::StandardHMC) with runMD_QPQ_sw!
    (standardHMC.jl:41-91, standardMD.jl:146-166) amounts to.  (The callers' own call sequences: tests/test_gpu_reference_callers.py.)"""

    def __init__(self, lq, U, kappa, beta, dtau, mdsteps, nsw, seed, dirac="Wilson", csw=0.0):
        self.lq, self.U, self.beta, self.dtau, self.mdsteps, self.nsw = lq, U, beta, dtau, mdsteps, nsw
        lat = U.lattice
        self.D = lq.Dirac_operator(U, None, {"Dirac_operator": dirac, "κ": kappa, "Clover_coefficient": csw, "boundarycondition": BC,
                                            "eps_CG": 1e-19})
        self.fa = lq.FermiAction(self.D)
        self.p, self.G, self.Uold = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
        self.xi, self.eta = lq.Fermionfields(lat, lq.WILSON), lq.Fermionfields(lat, lq.WILSON)
        self.rng = np.random.default_rng(seed)
        self.seed = seed
        self.dH, self.accepted = [], []

    def U_update(self, eps):
        self.lq.U_update_(self.U, self.p, eps * self.dtau)

    def P_update(self, eps):
        self.lq.P_update_(self.U, self.p, eps * self.dtau, self.beta)

    def P_update_fermion(self, eps):
        self.lq.calc_UdSfdU_(self.G, self.fa, self.U, self.eta)
        self.lq.Traceless_antihermitian_add_(self.p, eps * self.dtau, self.G)

    def run_md(self):
        legs = {"U": self.U_update, "G": self.P_update, "F": self.P_update_fermion}
        for _ in range(self.mdsteps):
            for leg, coeff in stages("QPQ_sw", self.nsw):      # the integrator table the oracle's trajectory runs on (tests/oracle_md.py)
                legs[leg](coeff)

    def H_new(self):
        return self.lq.momentum_action(self.p) + self.lq.evaluate_GaugeAction(self.U, self.beta) + self.lq.evaluate_FermiAction(self.fa, self.U, self.eta)

    def update(self):
        lq = self.lq
        lq.substitute_U_(self.Uold, self.U)
        self.seed += 3
        lq.gauss_distribution_(self.p, self.seed)                               # initialize_MD!
        lq.gauss_sampling_in_action_(self.xi, self.U, self.fa, self.seed + 1)
        lq.sample_pseudofermions_(self.eta, self.U, self.fa, self.xi)
        Hold = lq.momentum_action(self.p) + lq.evaluate_GaugeAction(self.U, self.beta) + lq.dot(self.xi, self.xi).real
        self.run_md()
        dH = self.H_new() - Hold
        accept = np.exp(-dH) >= self.rng.random()
        if not accept:
            lq.substitute_U_(self.U, self.Uold)
        self.dH.append(dH)
        self.accepted.append(bool(accept))
        return accept


def _fixture(lq):
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    return L, Uh, lq.Gaugefields(lq.Lattice(L)).upload(Uh)


@pytest.mark.parametrize("dirac,csw", [("Wilson", 0.0), ("WilsonClover", 1.0)])
def test_hmc_energy_conservation_and_reversibility_on_device(gpu, orc, dirac, csw):
    """Wilson: the reference's 2-flavour HMC.  WilsonClover (BASELINE.json configs[3]; the reference rejects the operator): the same
    trajectory with the clover term in the action and its derivative in the force -- the integrator stays second order and reversible."""
    lq = gpu
    L, Uh, U = _fixture(lq)
    dH = []
    for mdsteps in (10, 20):
        U.upload(Uh)
        h = DeviceHMC(lq, U, KAPPA, BETA, 0.5 / mdsteps, mdsteps, 10, seed=400, dirac=dirac, csw=csw)
        lq.gauss_distribution_(h.p, 401)
        lq.gauss_distribution_fermion_(h.xi, 402)
        lq.sample_pseudofermions_(h.eta, U, h.fa, h.xi)
        H0 = lq.momentum_action(h.p) + lq.evaluate_GaugeAction(U, BETA) + lq.dot(h.xi, h.xi).real
        assert abs(lq.evaluate_FermiAction(h.fa, U, h.eta) - lq.dot(h.xi, h.xi).real) < 1e-8 * H0    # S_f(eta = D'xi) = xi'xi
        h.run_md()
        dH.append(h.H_new() - H0)
    assert abs(dH[1]) < abs(dH[0]) < 3.0 and 3.0 < abs(dH[0] / dH[1]) < 5.0          # second-order integrator (H ~ 5000)
    # reversibility: flip the momenta, integrate back
    P = h.p.download()
    h.p.upload(-P)
    h.run_md()
    assert np.abs(U.download() - Uh).max() < 1e-9


def test_hmc_repeats_the_reference_wilson_test_on_device(gpu, orc):
    """runtests.jl:88-99: |plaq - plaq_comparison| / plaq_comparison < 0.1 after Nsteps = 10 trajectories of test_wilson.toml."""
    lq = gpu
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        start_plaq = json.load(f)["plaquette"]["confs_HMC_L04040404_beta5.7_Wilson_kappa0.141139"]
    L, Uh, U = _fixture(lq)
    assert abs(lq.calculate_Plaquette(U) - start_plaq) < 1e-13
    h = DeviceHMC(lq, U, KAPPA, BETA, dtau=0.05, mdsteps=20, nsw=10, seed=111)
    for _ in range(10):
        h.update()
    plaq = lq.calculate_Plaquette(U)
    print("HMC: dH =", ["%.3f" % d for d in h.dH], "accepted", sum(h.accepted), "/ 10, plaquette", plaq)
    assert abs(plaq - REF_PLAQ_WILSON_HMC) / REF_PLAQ_WILSON_HMC < 0.1                # the reference's own criterion
    assert sum(h.accepted) >= 6 and np.abs(np.array(h.dH)).max() < 2.0               # and the integrator actually works
    assert abs(plaq - start_plaq) > 1e-6                                             # the configuration did move
    assert orc.unitarity_dev(U.download(), L) < 1e-9                                # exp(dt P) keeps the links in SU(3)


def test_reference_criterion_discriminates_a_wrong_pseudofermion_weight(gpu, orc):
    """Negative control: the same run with the pseudofermion noise drawn with <|xi_i|^2> = 2 (weight exp(-S_f/2) instead of
    exp(-S_f)) drifts to a plaquette of about 0.47 and FAILS the reference's 10 % criterion -- the end-to-end golden value does
    constrain the fermion sector, not only the plumbing."""
    lq = gpu
    L, Uh, U = _fixture(lq)
    h = DeviceHMC(lq, U, KAPPA, BETA, dtau=0.05, mdsteps=20, nsw=10, seed=111)
    good = lq.gauss_sampling_in_action_
    try:
        lq.gauss_sampling_in_action_ = lambda xi, U_, fa, seed=112: lq.gauss_distribution_fermion_(xi, seed)
        for _ in range(10):
            h.update()
    finally:
        lq.gauss_sampling_in_action_ = good
    plaq = lq.calculate_Plaquette(U)
    assert abs(plaq - REF_PLAQ_WILSON_HMC) / REF_PLAQ_WILSON_HMC > 0.1, plaq


def test_hmc_repeats_the_reference_quenched_su3_test_on_device(gpu, orc):
    """runtests.jl:31-38 with test/test01.toml: quenched SU(3) HMC, beta = 5.7, dtau = 1/15, 15 MD steps, 10 trajectories from the
    reference's thermalised configuration; final plaquette within 10 % of debugplaqdata.txt line 2 -- the gauge side of the MD step
    (momenta, staple force fused into the momentum update, link exponential, actions) alone against the reference's golden."""
    lq = gpu
    L, dtau, mdsteps = (4, 4, 4, 4), 1.0 / 15.0, 15
    ref_plaq = 0.55783720583739                                     # /root/reference/test/debugplaqdata.txt:2
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "quenched_su3_4x4x4x4.ildg"), L)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    p, Uold = lq.Gaugefields(lat), lq.Gaugefields(lat)
    start = lq.calculate_Plaquette(U)
    rng = np.random.default_rng(113)
    dHs, acc = [], 0
    for traj in range(10):
        lq.substitute_U_(Uold, U)
        lq.gauss_distribution_(p, 5000 + traj)
        Hold = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA)
        for _ in range(mdsteps):
            lq.U_update_(U, p, 0.5 * dtau)
            lq.P_update_(U, p, dtau, BETA)
            lq.U_update_(U, p, 0.5 * dtau)
        dH = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) - Hold
        dHs.append(dH)
        if np.exp(-dH) >= rng.random():
            acc += 1
        else:
            lq.substitute_U_(U, Uold)
    plaq = lq.calculate_Plaquette(U)
    print("quenched SU(3) HMC: dH =", ["%.3f" % d for d in dHs], "accepted", acc, "/ 10, plaquette %.6f (start %.6f)" % (plaq, start))
    assert abs(plaq - ref_plaq) / ref_plaq < 0.1
    assert acc >= 6 and np.abs(dHs).max() < 2.0 and abs(plaq - start) > 1e-6
    assert orc.unitarity_dev(U.download(), L) < 1e-9


def test_polyakov_loop(lq, orc):
    """The second observable of every trajectory of the reference's runs (Polyakov_loop in every toml under test/): 1/(NC NX NY NZ) sum_x tr prod_t U_4(x, t).
    Cold links give 1; a hot configuration and the reference's own 4^4 Wilson fixture against the numpy restatement; a centre transformation of one
    time slice multiplies the loop by exp(2 pi i / 3) and leaves the plaquette alone."""
    import os
    from conftest import GOLDEN
    L = (4, 6, 4, 8)
    lat = lq.Lattice(L)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="cold", lattice=lat)
    assert abs(lq.calculate_Polyakov_loop(U) - 1.0) < 1e-15
    Uh = orc.hot_gauge(L, 17)
    U.upload(Uh)
    p = lq.calculate_Polyakov_loop(U)
    assert abs(p - orc.polyakov_loop(Uh, L)) < 1e-14
    z = np.exp(2j * np.pi / 3)
    Uz = Uh.copy()
    Uz[3, 5] *= z                                   # every time-like link of the slice t = 5
    U.upload(Uz)
    assert abs(lq.calculate_Polyakov_loop(U) - z * p) < 1e-14
    assert abs(lq.calculate_Plaquette(U) - orc.plaquette(Uh, L)) < 1e-14
    L4 = (4, 4, 4, 4)
    Uf = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L4)
    U4 = lq.Gaugefields(lq.Lattice(L4)).upload(Uf)
    assert abs(lq.calculate_Polyakov_loop(U4) - orc.polyakov_loop(Uf, L4)) < 1e-14
