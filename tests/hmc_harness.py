"""Device-resident HMC streams for the statistical parity tests (tests/test_gpu_hmc_statistics.py) and scripts/hmc_stats.py.
One trajectory = momentum + pseudofermion refresh, `mdsteps` MD steps of the integrator's stage table (tests/oracle_md.py `stages`, the table the
oracle's trajectory runs on: QPQ leapfrog, or its Sexton-Weingarten form for the actions the reference's test files run with SextonWeingargten = true),
Metropolis test -- on the fused four-direction entry points.  (The reference callers' own per-direction sequences are replayed from their call trace in
tests/test_gpu_reference_callers.py.)"""
import os

import numpy as np

from oracle_md import stages

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BETA, MASS, KAPPA = 5.7, 0.5, 0.141139
BC = (1, 1, 1, -1)
L4 = (4, 4, 4, 4)

# action key -> (fixture, operator parameters, FermiAction parameters, dtau, MD steps, Sexton-Weingarten N) as in the reference's
# test/*.toml files (runtests.jl:31-130)
ACTIONS = {
    "quenched": ("quenched_su3_4x4x4x4.ildg", None, None, 1.0 / 15, 15, 0),
    "wilson_nf2": ("wilson_4x4x4x4.ildg", {"Dirac_operator": "Wilson", "κ": KAPPA}, {}, 0.05, 20, 10),
    "staggered_nf4_evensite": ("staggered_4x4x4x4.ildg", {"Dirac_operator": "Staggered", "mass": MASS}, {"Nf": 4}, 0.025, 40, 0),
    "staggered_nf4_rational": ("staggered_4x4x4x4.ildg", {"Dirac_operator": "Staggered", "mass": MASS}, {"Nf": 4, "force_rational": True}, 0.025, 40, 0),
    "staggered_nf2": ("staggered_nf2_4x4x4x4.ildg", {"Dirac_operator": "Staggered", "mass": MASS}, {"Nf": 2}, 0.05, 20, 0),
    "staggered_nf3": ("staggered_nf3_4x4x4x4.ildg", {"Dirac_operator": "Staggered", "mass": MASS}, {"Nf": 3}, 0.05, 20, 0),
}


def run_stream(lq, action, ntraj, seed, dtau=None, mdsteps=None, params=None):
    """Returns dict(plaq = plaquette after every trajectory, dH, accepted) for `ntraj` trajectories from the reference's thermalised
    fixture of that action.  params: library tunables to set on the lattice context (e.g. {"mixed_action_solver": 2})."""
    fixture, op_par, fa_par, dt0, n0, nsw = ACTIONS[action]
    dtau = dt0 if dtau is None else dtau
    mdsteps = n0 if mdsteps is None else mdsteps
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, fixture), L4)
    lat = lq.Lattice(L4)
    for key, val in (params or {}).items():
        lat.set_param(key, val)
    U = lq.Gaugefields(lat).upload(Uh)
    p, G, Uold = lq.Gaugefields(lat), lq.Gaugefields(lat), lq.Gaugefields(lat)
    fa = xi = phi = None
    if op_par is not None:
        D = lq.Dirac_operator(U, None, dict(op_par, boundarycondition=BC, eps_CG=1e-19))
        fa = lq.FermiAction(D, fa_par)
        kind = D.kind
        xi, phi = lq.Fermionfields(lat, kind), lq.Fermionfields(lat, kind)
    rng = np.random.default_rng(seed)

    def H():
        h = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA)
        return h + (lq.evaluate_FermiAction(fa, U, phi) if fa is not None else 0.0)

    def md_step():
        for leg, coeff in stages("QPQ_sw" if nsw else "QPQ", nsw):
            if leg == "U":
                lq.U_update_(U, p, coeff * dtau)
            elif leg == "G":
                lq.P_update_(U, p, coeff * dtau, BETA)
            elif fa is not None:
                lq.calc_UdSfdU_(G, fa, U, phi)
                lq.Traceless_antihermitian_add_(p, coeff * dtau, G)

    plaq, dHs, acc = [], [], []
    for traj in range(ntraj):
        lq.substitute_U_(Uold, U)
        lq.gauss_distribution_(p, seed * 100003 + 3 * traj)
        if fa is not None:
            lq.gauss_sampling_in_action_(xi, U, fa, seed * 100003 + 3 * traj + 1)
            lq.sample_pseudofermions_(phi, U, fa, xi)
        H0 = H()
        for _ in range(mdsteps):
            md_step()
        dH = H() - H0
        ok = bool(np.exp(-dH) >= rng.random())
        if not ok:
            lq.substitute_U_(U, Uold)
        plaq.append(lq.calculate_Plaquette(U))
        dHs.append(dH)
        acc.append(ok)
    for f in (p, G, Uold, xi, phi):
        if f is not None:
            f.close()
    return {"plaq": np.array(plaq), "dH": np.array(dHs), "accepted": np.array(acc)}


def binned(x, nbin=10):
    """mean and error of the mean from `nbin` bins (the bins absorb the autocorrelation of consecutive trajectories)."""
    x = np.asarray(x, dtype=float)
    n = (len(x) // nbin) * nbin
    b = x[len(x) - n:].reshape(nbin, -1).mean(axis=1)
    return float(b.mean()), float(b.std(ddof=1) / np.sqrt(nbin))
