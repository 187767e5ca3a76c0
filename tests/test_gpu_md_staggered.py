"""The reference's staggered HMC test on the device (test/runtests.jl:101-112 with test/test_staggered.toml): thermalised 4^4
staggered configuration, beta = 5.7, mass = 0.5, 4 tastes, dtau = 0.025, 40 MD steps (plain QPQ leapfrog, standardMD.jl:125-139),
10 trajectories; final plaquette within 10 % of test/debugplaqdata.txt line 8.
Four tastes = the pseudofermion lives on the even sites only: D^+D = m^2 - D_hop^2 is block diagonal in parity, so
S_f = phi_e^+ (D^+D)^-1 phi_e with phi_e = (D^+ xi)_e is the same solve and the same force with the odd half of phi zeroed."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

BETA, MASS = 5.7, 0.5
BC = (1, 1, 1, -1)
REF_PLAQ_STAGGERED_HMC = 0.5734383856968012       # /root/reference/test/debugplaqdata.txt:8 (plaqvalues[8], runtests.jl:107)


def test_hmc_repeats_the_reference_staggered_test_on_device(lq, orc):
    assert lq.lib.device_count() > 0
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "staggered_4x4x4x4.ildg"), L)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    start = lq.calculate_Plaquette(U)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": MASS, "boundarycondition": BC, "eps_CG": 1e-19})
    fa = lq.FermiAction(D, {"Nf": 4})
    p, G, Uold = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
    xi, phi = lq.Fermionfields(lat, lq.STAGGERED), lq.Fermionfields(lat, lq.STAGGERED)
    dtau, mdsteps = 0.025, 40
    rng = np.random.default_rng(111)
    dHs, acc = [], 0
    for traj in range(10):
        lq.substitute_U_(Uold, U)
        lq.gauss_distribution_(p, 500 + traj)
        lq.gauss_sampling_in_action_(xi, U, fa, 600 + traj)
        lq.sample_pseudofermions_(phi, U, fa, xi)          # phi = (D^+ xi) restricted to the even sites (Nf = 4: 4 tastes)
        Hold = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.evaluate_FermiAction(fa, U, phi)
        for _ in range(mdsteps):                           # runMD_QPQ!
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
    print("staggered HMC: dH =", ["%.3f" % d for d in dHs], "accepted", acc, "/ 10, plaquette %.6f (start %.6f)" % (plaq, start))
    assert abs(plaq - REF_PLAQ_STAGGERED_HMC) / REF_PLAQ_STAGGERED_HMC < 0.1
    assert acc >= 6 and np.abs(dHs).max() < 2.0
    assert abs(plaq - start) > 1e-6 and orc.unitarity_dev(U.download(), L) < 1e-9
