"""Dirac_operator = "Domainwall" (universe.jl:116-128; test/test_domainwallhmc.toml; the fifth HMC fermion test of test/runtests.jl:132-137) against the
oracle's restatement (oracle/oracle.py domainwall_*: numpy over the C Wilson operator, itself checked by identities in tests/test_oracle_domainwall.py), and
the reference's test case -- Domainwall_m = 1 = the Pauli-Villars mass -- through the reference's callers replayed from their call trace."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, rel_err

pytestmark = pytest.mark.gpu
BC = (1, 1, 1, -1)


def _setup(lq, orc, L, L5, M, mass, seed=11, bc=BC, **kw):
    Uh = orc.hot_gauge(L, seed)
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    x = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=L5, nowing=True)
    params = {"Dirac_operator": "Domainwall", "mass": mass, "L5": L5, "M": M, "eps_CG": 1e-19, "MaxCGstep": 3000, "boundarycondition": bc}
    params.update(kw)
    D = lq.Dirac_operator(U, x, params)
    return Uh, lat, U, x, D


def _rand5(orc, L, L5, seed):
    rng = np.random.default_rng(seed)
    shp = (L5,) + orc.wilson_shape(L)
    return rng.standard_normal(shp) + 1j * rng.standard_normal(shp)


@pytest.mark.parametrize("L,L5,M,mass,bc", [((4, 4, 4, 8), 4, -1.0, 0.25, BC), ((8, 4, 4, 4), 6, -1.4, 0.05, (1, 1, 1, 1)), ((4, 4, 4, 4), 2, -1.0, 1.0, (1, -1, 1, -1)),
                                            ((16, 8, 8, 8), 3, -1.8, 0.1, BC)])
def test_mul_matches_the_oracle(lq, orc, L, L5, M, mass, bc):
    Uh, lat, U, x, D = _setup(lq, orc, L, L5, M, mass, bc=bc)
    ph = _rand5(orc, L, L5, 3)
    x.upload(ph)
    assert np.array_equal(x.download(), ph)                          # slices go up and come down where they belong
    assert np.array_equal(x.w[L5 - 1].download(), ph[L5 - 1])        # x.w[i5]: a Wilson field aliasing one slice
    y = x.similar()
    for dagger in (False, True):
        lq.mul_(y, D.adjoint() if dagger else D, x)
        ref = orc.domainwall_D(Uh, ph, L, M, mass, bc, dagger=dagger)
        assert rel_err(y.download(), ref) < 1e-13
    lq.mul_(y, lq.DdagD_operator(D), x)
    ref = orc.domainwall_D(Uh, orc.domainwall_D(Uh, ph, L, M, mass, bc), L, M, mass, bc, dagger=True)
    assert rel_err(y.download(), ref) < 1e-13
    # BLAS-1 takes the five-dimensional field as one array
    assert abs(lq.dot(x, x) - np.vdot(ph, ph)) < 1e-11 * abs(np.vdot(ph, ph))


def test_five_dimensional_launch_equals_the_slice_by_slice_form(lq, orc):
    """Where the scalar-addressing Wilson kernel applies (z-planes of whole chunks, links on the group) an application is ONE launch over all slices with the
    fifth-direction hops in its epilogue (tunable dw_batched, read-only dw_active); elsewhere L5 Wilson launches + one pass.  Same result, both against the oracle."""
    L, L5, M, mass = (16, 8, 8, 16), 5, -1.4, 0.07
    Uh, lat, U, x, D = _setup(lq, orc, L, L5, M, mass)
    ph = _rand5(orc, L, L5, 9)
    x.upload(ph)
    y = x.similar()
    got = {}
    for batched in (1, 0):
        lat.set_param("dw_batched", batched)
        for dagger in (False, True):
            lq.mul_(y, D.adjoint() if dagger else D, x)
            assert lat.get_param("dw_active") == batched
            got[(batched, dagger)] = y.download()
            assert rel_err(got[(batched, dagger)], orc.domainwall_D(Uh, ph, L, M, mass, BC, dagger=dagger)) < 1e-13
    for dagger in (False, True):
        assert np.abs(got[(1, dagger)] - got[(0, dagger)]).max() < 1e-14 * np.abs(got[(0, dagger)]).max()
    # links that are not on the group to 1e-14 (the reference's text configurations): the five-dimensional launch reads 12-real links only and steps aside
    rng = np.random.default_rng(10)
    U.upload(Uh + 1e-10 * (rng.standard_normal(Uh.shape) + 1j * rng.standard_normal(Uh.shape)))
    lat.set_param("dw_batched", 1)
    lq.mul_(y, D, x)
    assert lat.get_param("dw_active") == 0
    assert rel_err(y.download(), orc.domainwall_D(U.download(), ph, L, M, mass, BC)) < 1e-13


def test_cg_solution_matches_the_oracle(lq, orc):
    L, L5, M, mass = (4, 4, 4, 8), 4, -1.0, 0.1
    Uh, lat, U, b, D = _setup(lq, orc, L, L5, M, mass)
    bh = _rand5(orc, L, L5, 4)
    b.upload(bh)
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    xo, ito, rro = orc.domainwall_cg(Uh, bh, L, M, mass, BC, eps=1e-19)
    assert rr < 1e-19 and abs(it - ito) <= 2
    xh = x.download()
    assert rel_err(xh, xo) < 1e-9
    res = bh - orc.domainwall_D(Uh, orc.domainwall_D(Uh, xh, L, M, mass, BC), L, M, mass, BC, dagger=True)
    assert np.vdot(res, res).real < 4e-19                              # the true residual of the device solution, by the oracle's operator


@pytest.mark.parametrize("L,L5", [((16, 8, 8, 8), 4), ((16, 16, 16, 16), 4)])      # 512 and 8192 partials per reduction (the one-wave and the 1024-thread form of reduce_final)
def test_fused_cg_on_the_five_dimensional_launch(lq, orc, L, L5):
    """dw_fused_cg: where all slices go through one launch the CG runs the fused iteration of the four-dimensional solver (|D p|^2 and the update r -= alpha D^+ t in the
    operator's epilogues, one partial per chunk and slice).  Against the generic 11-pass loop (same solution, same count) and the oracle's operator (true residual < eps)."""
    M, mass = -1.0, 0.1
    Uh, lat, U, b, D = _setup(lq, orc, L, L5, M, mass)
    bh = _rand5(orc, L, L5, 4)
    b.upload(bh)
    got = {}
    for fused in (1, 0):
        lat.set_param("dw_fused_cg", fused)
        x = b.similar()
        it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
        assert lat.get_param("dw_active") == 1 and rr < 1e-19
        got[fused] = (it, x.download())
    lat.set_param("dw_fused_cg", 1)
    assert abs(got[1][0] - got[0][0]) <= 1 and rel_err(got[1][1], got[0][1]) < 1e-10
    orc.set_threads(os.cpu_count() or 1)      # (the oracle's own CG at this size takes two minutes: its operator certifies the solution instead; the generic loop is
    try:                                      #  compared with the oracle's CG in test_cg_solution_matches_the_oracle)
        res = bh - orc.domainwall_D(Uh, orc.domainwall_D(Uh, got[1][1], L, M, mass, BC), L, M, mass, BC, dagger=True)
    finally:
        orc.set_threads(1)
    assert np.vdot(res, res).real < 4e-19


def test_action_heat_bath_and_force(lq, orc):
    L, L5, M, mass = (4, 4, 4, 4), 4, -1.0, 0.2
    Uh, lat, U, phi, D = _setup(lq, orc, L, L5, M, mass, eps_CG=1e-22)
    fa = lq.FermiAction(D, {})
    xi = phi.similar()
    lq.gauss_sampling_in_action_(xi, U, fa, 21)
    xih = xi.download()
    assert 0.9 < np.vdot(xih, xih).real / xih.size < 1.1               # <|xi_i|^2> = 1 on every slice
    assert abs(np.vdot(xih[0], xih[1])) < 0.2 * np.vdot(xih[0], xih[0]).real   # slices are independent streams
    lq.sample_pseudofermions_(phi, U, fa, xi)
    phh = phi.download()
    assert rel_err(phh, orc.domainwall_sample(Uh, xih, L, M, mass, BC)) < 1e-9
    S = lq.evaluate_FermiAction(fa, U, phi)
    assert abs(S - np.vdot(xih, xih).real) < 1e-9 * S                   # the heat bath identity S(phi) = xi^+ xi
    So, Xo, Yo = orc.domainwall_action(Uh, phh, L, M, mass, BC, eps=1e-24)
    assert abs(S - So) < 1e-10 * So
    G = lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G, fa, U, phi)
    Go = orc.domainwall_force(Uh, phh, L, M, mass, BC, eps=1e-24)
    assert rel_err(G.download(), Go) < 1e-8
    # Nf other than 2 has no Domainwall action
    with pytest.raises(lq.LQCDError):
        lq.FermiAction(D, {"Nf": 1})
    # entry points that do not serve the operator say so
    y = phi.similar()
    with pytest.raises(lq.LQCDError, match="Domainwall"):
        D.method_CG = "bicgstab"
        lq.solve_DinvX_(y, D, phi)


def test_force_is_the_derivative_of_the_device_action(lq, orc):
    import scipy.linalg as sla
    L, L5, M, mass = (4, 4, 4, 4), 3, -1.2, 0.15
    Uh, lat, U, phi, D = _setup(lq, orc, L, L5, M, mass, seed=31, eps_CG=1e-24)
    fa = lq.FermiAction(D, {})
    phi.upload(_rand5(orc, L, L5, 32))
    G = lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G, fa, U, phi)
    Gh = G.download()
    rng = np.random.default_rng(33)
    for _ in range(3):
        mu, t, z, y, x = (int(rng.integers(n)) for n in (4, L[3], L[2], L[1], L[0]))
        T = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
        T = T + T.conj().T
        h, S = 1e-4, []
        for e in (h, -h):
            V = Uh.copy()
            V[mu, t, z, y, x] = V[mu, t, z, y, x] @ sla.expm(1j * e * T).T      # host image [b, a]: the transpose of the matrix
            U2 = lq.Gaugefields(lat).upload(V)
            S.append(lq.evaluate_FermiAction(fa, U2, phi))
        fd = (S[0] - S[1]) / (2 * h)
        an = -2.0 * np.imag(np.trace(T @ Gh[mu, t, z, y, x].T))
        assert abs(fd - an) < 2e-6 * max(1.0, abs(fd)), (fd, an)


def test_reference_test_case_pauli_villars_mass_is_a_spectator(lq, orc):
    """test/test_domainwallhmc.toml: Domainwall_m = 1.0 = the Pauli-Villars mass, M = -1, L5 = 4, beta 5.7, dtau 0.05, 20 MD steps, started from the
    4x4x2x2 configuration of test/confs_HMC_L04040404_beta5.7_Domainwall.  D = D_PV: S_f = phi^+ phi for every gauge field and the fermion force vanishes, so the
    trajectory is the quenched one with a spectator field -- through the reference's unchanged callers, replayed from the trace their run emitted (tests/ref_trace.py)."""
    from ref_trace import Replay, standard_hmc, standard_md
    L = (4, 4, 2, 2)
    Uh = lq.gauge_io.load_BridgeText(os.path.join(GOLDEN, "domainwall_4x4x2x2.ildg.txt"), L)
    plaq_file = orc.plaquette(Uh, L)
    res = {}
    for quench in (False, True):
        lat = lq.Lattice(L)
        U = lq.Gaugefields(lat).upload(Uh)
        assert abs(lq.calculate_Plaquette(U) - plaq_file) < 1e-13
        ga = lq.GaugeAction(U)
        pl = lq.make_loops_fromname("plaquette", Dim=4)
        ga.push_(5.7 / 2, pl + lq.make_loops_fromname("plaquette", Dim=4, adjoint=True))
        fa = None
        if not quench:
            x = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=4, nowing=True)
            D = lq.Dirac_operator(U, x, {"Dirac_operator": "Domainwall", "mass": 1.0, "L5": 4, "M": -1.0, "eps_CG": 1e-19, "verbose_level": 2,
                                         "MaxCGstep": 3000, "boundarycondition": BC})
            fa = lq.FermiAction(D, {})
        md = standard_md(lq, U, ga, 0.05, 20, fermi_action=fa)
        hmc, dHs = standard_hmc(lq, U, md), []
        # (the momenta take the first seed of a trajectory in both runs: the quenched replay skips the two pseudofermion draws, so seeds are set per trajectory)
        rp = Replay(lq, hooks={("after", "update!"): lambda r: dHs.append(r.watch("H_new") - r.watch("H_old"))})
        acc = []
        for traj in range(2):
            rp.seed, rp.rng = 5 + 10 * traj, np.random.default_rng(5 + 10 * traj)
            acc.append(rp.call("update!", hmc, U))
        res[quench] = (U.download(), dHs, acc, lq.calculate_Plaquette(U))
        if not quench:
            S = lq.evaluate_FermiAction(fa, U, md["η"])
            assert abs(S - lq.dot(md["η"], md["η"]).real) < 1e-9 * S      # S_f = phi^+ phi on the evolved links
    assert np.abs(res[False][0] - res[True][0]).max() < 1e-8            # the same links as the quenched trajectory (same momenta seeds) ...
    assert np.abs(np.array(res[False][1]) - np.array(res[True][1])).max() < 1e-7      # ... and the same dH: the spectator's action does not move
    assert all(abs(d) < 0.5 for d in res[False][1])


def test_rccl_self_partition_domainwall(lq, orc):
    """The Domainwall operator on a partitioned lattice: LQCD_FORCE_PARTITION + a world-size-1 RCCL communicator run the halo path of the Wilson slices, the
    rank-summed inner products of the CG and the face exchange of the force sweep exactly as at N > 1; operator, CG, action, heat bath and force equal the oracle."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, L5, M, MASS, BC = (8, 4, 6, 8), 3, -1.3, 0.2, (1, 1, 1, -1)
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        Uh = orc.hot_gauge(L, 111)
        U = lq.Gaugefields(lat).upload(Uh)
        x = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=L5)
        D = lq.Dirac_operator(U, x, {"Dirac_operator": "Domainwall", "mass": MASS, "L5": L5, "M": M, "boundarycondition": BC, "eps_CG": 1e-20})
        rng = np.random.default_rng(5)
        shp = (L5,) + orc.wilson_shape(L)
        ph = rng.standard_normal(shp) + 1j * rng.standard_normal(shp)
        x.upload(ph)
        y = x.similar()
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            assert rel(y.download(), orc.domainwall_D(Uh, ph, L, M, MASS, BC, dagger=dag)) < 1e-13, dag
        lq.clear_fermion_(y)                      # solve_DinvX! starts from what y holds
        it, rr = lq.solve_DinvX_(y, lq.DdagD_operator(D), x, return_info=True)
        xo, ito, _ = orc.domainwall_cg(Uh, ph, L, M, MASS, BC, eps=1e-20)
        assert abs(it - ito) <= 2 and rel(y.download(), xo) < 1e-9
        fa = lq.FermiAction(D, {})
        S = lq.evaluate_FermiAction(fa, U, x)
        So, _, _ = orc.domainwall_action(Uh, ph, L, M, MASS, BC, eps=1e-24)
        assert abs(S - So) < 1e-9 * So
        G = lq.Gaugefields(lat)
        lq.calc_UdSfdU_(G, fa, U, x)
        assert rel(G.download(), orc.domainwall_force(Uh, ph, L, M, MASS, BC, eps=1e-24)) < 1e-8
        print("RCCL_SELF_DW_OK")
    """)
    for mask in ("8", "15"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "RCCL_SELF_DW_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])


def test_slice_views_outlive_their_field_safely(lq, orc):
    """Finalizers of a garbage collector run in any order: a five-dimensional field destroyed while slice views exist keeps its storage until the last view is gone."""
    L, L5 = (8, 8, 8, 8), 6
    lat = lq.Lattice(L)
    x = lq.Fermionfields(lat, lq.DOMAINWALL, L5=L5)
    ph = _rand5(orc, L, L5, 17)
    x.upload(ph)
    import ctypes as C

    def free_bytes():
        f, t = C.c_int64(0), C.c_int64(0)
        lq.lib.check(lq.lib.lib().lqcd_device_mem_info(0, C.byref(f), C.byref(t)))
        return f.value
    free0 = free_bytes()
    views = x.w
    x.close()                                        # the parent goes first
    assert np.array_equal(views[3].download(), ph[3])      # the storage is still there
    for v in views[:-1]:
        v.close()
    assert np.array_equal(views[-1].download(), ph[-1])
    views[-1].close()                                # the last view frees it
    assert free_bytes() >= free0 + ph.nbytes // 2      # the field's storage came back only now
