"""Stout smearing of the links the fermion action sees (universe.jl:147-171; standardMD.jl:82-101, 192-227; standardHMC.jl:67-68): the device layer and its
back-propagation against the numpy restatement (oracle.stout_*, itself checked by finite differences in tests/test_cpu_stout_restatement.py), the fermion force
through the smearing against finite differences of the device action, and energy conservation of the reference's unchanged callers -- their CovNeuralnet
methods replayed from the call trace (tests/golden/ref_exec_traces.json, tests/ref_trace.py: the P_update_fermion! method that dispatches on TC <: CovNeuralnet,
the smeared branch of initialize_MD!, update! with md.cov_neural_net set), as tests/test_gpu_reference_callers.py replays the others."""
import numpy as np
import pytest
import scipy.linalg as sla

from conftest import rel_err
from ref_trace import Replay, standard_hmc, standard_md
from test_gpu_reference_callers import plaquette_action, wilson_action

pytestmark = pytest.mark.gpu
Dim = 4
BC = (1, 1, 1, -1)


@pytest.mark.parametrize("L,rho", [((4, 4, 4, 8), 0.1), ((8, 4, 4, 4), 0.15)])
def test_smearing_and_backpropagation_match_the_oracle(lq, orc, L, rho):
    Uh = orc.hot_gauge(L, 7)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    nn = lq.CovNeuralnet(U)
    nn.push_(lq.STOUT_Layer(["plaquette"], [rho], U))
    Uout, multi, _ = lq.calc_smearedU(U, nn)
    ref = orc.stout_smear(Uh, L, rho)
    assert rel_err(Uout.download(), ref) < 1e-13
    assert lq.unitarity_deviation(Uout) < 1e-13 and lq.calculate_Plaquette(Uout) > lq.calculate_Plaquette(U) + 0.05
    rng = np.random.default_rng(8)
    Gs = rng.standard_normal(orc.gauge_shape(L)) + 1j * rng.standard_normal(orc.gauge_shape(L))
    Gd, G = lq.Gaugefields(lat).upload(Gs), lq.Gaugefields(lat)
    lq.lib.check(lq.lib.lib().lqcd_stout_backprop(G._h, Gd._h, U._h, __import__("ctypes").c_double(rho)))
    assert rel_err(G.download(), orc.stout_backprop(Gs, Uh, L, rho)) < 1e-12
    assert np.array_equal(Gd.download(), Gs)                          # out of place: the input is left alone
    lq.lib.check(lq.lib.lib().lqcd_stout_backprop(Gd._h, Gd._h, U._h, __import__("ctypes").c_double(rho)))
    assert np.array_equal(Gd.download(), G.download())                # in place: the same bits
    with pytest.raises(lq.LQCDError):
        lq.STOUT_Layer(["plaquette", "rectangular"], [0.1, 0.05], U)


def stout_net(lq, U, rhos):
    """The stack of stout layers between the links of the MD and the links the fermion action sees (universe.jl:147-171): one plaquette layer per rho."""
    nn = lq.CovNeuralnet(U)
    for rho in rhos:
        nn.push_(lq.STOUT_Layer(["plaquette"], [rho], U))
    return nn


def smeared_md(lq, U, kappa, rhos, dtau, steps, eps=1e-19):
    fa = wilson_action(lq, U, kappa, eps)
    return standard_md(lq, U, plaquette_action(lq, U, 5.7), dtau, steps, fermi_action=fa, cov_neural_net=stout_net(lq, U, rhos))


@pytest.mark.parametrize("rhos", [(0.1,), (0.08, 0.12)])
def test_fermion_force_through_the_smearing_is_the_derivative_of_the_action(lq, orc, rhos):
    L = (4, 4, 4, 4)
    Uh = orc.hot_gauge(L, 9)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    md = smeared_md(lq, U, 0.12, rhos, 0.05, 1, eps=1e-24)
    fa, nn = md.fermi_action, md.cov_neural_net
    rp = Replay(lq, seed=0)
    rp.call("initialize_MD!", U, md)                                  # the smeared branch: heat bath on calc_smearedU(U, nn)
    assert rp.log.count("calc_smearedU") == 1
    p0 = md.p.download()
    rp.call("P_update_fermion!", U, md.p, 1.0, md)                    # dispatches to the TC <: CovNeuralnet method (standardMD.jl:192-227)
    assert rp.log.count("back_prop") == 1 and rp.log.count("mul!") == 2 * Dim
    dp = (md.p.download() - p0) / (-md["Δτ"])                         # = TA(U dS/dU) in the reference's sign (factor = -eps dtau)
    rng = np.random.default_rng(10)

    def S(V):
        U2 = lq.Gaugefields(lat).upload(V)
        Uout, _, _ = lq.calc_smearedU(U2, nn)
        return lq.evaluate_FermiAction(fa, Uout, md["η"])

    for _ in range(3):
        idx = tuple(int(rng.integers(n)) for n in (4, L[3], L[2], L[1], L[0]))
        T = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
        T = T + T.conj().T
        T -= np.trace(T) / 3 * np.eye(3)
        h, vals = 1e-4, []
        for e in (h, -h):
            V = Uh.copy()
            V[idx] = V[idx] @ sla.expm(1j * e * T).T
            vals.append(S(V))
        fd = (vals[0] - vals[1]) / (2 * h)
        # dS/d eps = -2 Im tr(T G) with G the C ABI's field = -(the reference's U dS/dU); dp = TA(reference field): for traceless Hermitian T, tr(T X) sees only TA(X)
        an = 2.0 * np.imag(np.trace(T @ dp[idx].T))
        assert abs(fd - an) < 5e-6 * max(1.0, abs(fd)), (fd, an)


def test_hmc_with_stout_smeared_fermions_conserves_energy(lq, orc):
    L = (4, 4, 4, 4)
    Uh = orc.hot_gauge(L, 11)
    dH = {}
    for dtau, steps in ((0.02, 10), (0.01, 20)):
        lat = lq.Lattice(L)
        U = lq.Gaugefields(lat).upload(Uh)
        md = smeared_md(lq, U, 0.12, (0.1,), dtau, steps)
        out = {}
        rp = Replay(lq, seed=3, hooks={("after", "update!"): lambda r: out.update(dH=r.watch("H_new") - r.watch("H_old"))})
        rp.call("update!", standard_hmc(lq, U, md), U)
        assert rp.log.count("calc_smearedU") == 2 + steps and rp.log.count("back_prop") == steps      # heat bath, every fermion kick, final action
        dH[dtau] = out["dH"]
    assert abs(dH[0.02]) < 0.5 and abs(dH[0.01]) < 0.3 * abs(dH[0.02]) + 1e-3, dH      # second-order integrator: dH ~ dtau^2


def test_rccl_self_partition_stout(lq, orc):
    """The stout layer and its back-propagation on a partitioned lattice (LQCD_FORCE_PARTITION + world-size-1 RCCL communicators): the smearing through the
    staple sweep's ghost links and staple faces, the back-propagation's 24-loop gather from the halo-extended block of links and N matrices (it reaches the
    corner n + mu - nu of the neighbouring ranks) -- both against the numpy restatement."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, ctypes as C, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, rho = (8, 4, 6, 8), 0.11
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        Uh = orc.hot_gauge(L, 111)
        U = lq.Gaugefields(lat).upload(Uh)
        nn = lq.CovNeuralnet(U)
        nn.push_(lq.STOUT_Layer(["plaquette"], [rho], U))
        Uout, multi, _ = lq.calc_smearedU(U, nn)
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        assert rel(Uout.download(), orc.stout_smear(Uh, L, rho)) < 1e-13
        rng = np.random.default_rng(8)
        Gs = rng.standard_normal(orc.gauge_shape(L)) + 1j * rng.standard_normal(orc.gauge_shape(L))
        Gd, G = lq.Gaugefields(lat).upload(Gs), lq.Gaugefields(lat)
        lq.lib.check(lq.lib.lib().lqcd_stout_backprop(G._h, Gd._h, U._h, C.c_double(rho)))
        assert rel(G.download(), orc.stout_backprop(Gs, Uh, L, rho)) < 1e-12
        print("RCCL_SELF_STOUT_OK")
    """)
    for mask in ("8", "14", "15"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "RCCL_SELF_STOUT_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])
