"""A pin that does not go through the reference's un-vendored packages at all: the average plaquette of the pure SU(3) Wilson gauge action is one of the best known
numbers of lattice QCD.  A quenched HMC with the library's gauge legs (staple force, momentum heat bath and update, exponential link update, actions, Metropolis step as in
standardHMC.jl:41-91 with quench = true) must land on it -- which fixes the normalisation of beta (S_g = -(beta/3) sum Re tr U_p, the reference's `β/2` on the plaquette
and its adjoint, universe.jl:92-95), of the force and of the kinetic term together.  d<P>/d beta is about 0.15 here: the tolerance below resolves beta to better than 1 %.

Literature (infinite volume; on the 12^4 lattice used here finite-size effects are below the tolerance):
    beta = 5.7:  <P> = 0.5492        beta = 6.0:  <P> = 0.5937
(e.g. the tables of G. S. Bali and K. Schilling, Phys. Rev. D 47 (1993) 661, and S. Necco and R. Sommer, Nucl. Phys. B 622 (2002) 328)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(lq, L, beta, dtau, mdsteps, ntherm, nmeas, seed):
    lat = lq.Lattice(L)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="cold", lattice=lat)
    p, Uold = lq.initialize_TA_Gaugefields(U), lq.Gaugefields(lat)
    rng = np.random.default_rng(seed)
    plaq, acc, dHs = [], 0, []
    for it in range(ntherm + nmeas):
        lq.substitute_U_(Uold, U)
        lq.gauss_distribution_(p, seed + 7 * it + 1)
        H0 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, beta)
        for _ in range(mdsteps):                      # runMD_QPQ! with quench = true (standardMD.jl:127-144)
            lq.U_update_(U, p, 0.5 * dtau)
            lq.P_update_(U, p, dtau, beta)
            lq.U_update_(U, p, 0.5 * dtau)
        dH = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, beta) - H0
        ok = np.exp(-dH) >= rng.random()
        if not ok:
            lq.substitute_U_(U, Uold)
        if it >= ntherm:
            plaq.append(lq.calculate_Plaquette(U))
            acc += bool(ok)
            dHs.append(dH)
    plaq = np.array(plaq)
    nb = 20
    bins = plaq[: len(plaq) // nb * nb].reshape(nb, -1).mean(axis=1)
    return plaq.mean(), bins.std(ddof=1) / np.sqrt(nb), acc / nmeas, np.mean(np.exp(-np.array(dHs)))


@pytest.mark.parametrize("beta,lit", [(5.7, 0.5492), (6.0, 0.5937)])
def test_quenched_plaquette_lands_on_the_literature_value(lq, beta, lit):
    mean, err, acc, expdh = _run(lq, (12, 12, 12, 12), beta, 0.03, 33, 400, 3000, seed=int(100 * beta))
    print("beta %.1f: <P> = %.5f +- %.5f (literature %.4f), acceptance %.2f, <exp(-dH)> = %.3f" % (beta, mean, err, lit, acc, expdh))
    assert err < 3e-4 and acc > 0.6
    assert abs(mean - lit) < 8e-4 + 3 * err, (mean, err, lit)
    assert abs(expdh - 1.0) < 0.1
