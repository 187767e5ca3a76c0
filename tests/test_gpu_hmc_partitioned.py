"""A molecular-dynamics trajectory of the 2-flavour Wilson HMC on an in-process PE grid (four domains of one device, every
halo mechanism of the multi-GPU build active: Dslash halos in the CG, the X/Y face exchange of the fermion force, ghost links
and lower-staple faces of the gauge force) against the same trajectory on one domain."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

KAPPA, BETA = 0.141139, 5.7
BC = (1, 1, 1, -1)


def _single(lq, Uh, xi_h, L, dtau, nsteps):
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC, "eps_CG": 1e-22})
    fa = lq.FermiAction(D)
    p, G = lq.Gaugefields(lat), lq.Gaugefields(lat)
    lq.gauss_distribution_(p, 901)
    xi = lq.Fermionfields(lat, lq.WILSON).upload(xi_h)
    eta = xi.similar()
    lq.sample_pseudofermions_(eta, U, fa, xi)
    H0 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.dot(xi, xi).real
    for _ in range(nsteps):
        lq.U_update_(U, p, 0.5 * dtau)
        lq.P_update_(U, p, dtau, BETA)
        lq.calc_UdSfdU_(G, fa, U, eta)
        lq.Traceless_antihermitian_add_(p, dtau, G)
        lq.U_update_(U, p, 0.5 * dtau)
    H1 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA) + lq.evaluate_FermiAction(fa, U, eta)
    return U.download(), p.download(), H1 - H0


def test_md_trajectory_on_four_domains_equals_one_domain(lq, orc):
    assert lq.lib.device_count() > 0
    L, pe, dtau, nsteps = (8, 8, 8, 8), (1, 1, 2, 2), 0.05, 3
    # a mildly disordered start (exp(0.3 P) on a cold field) keeps the solves short
    Uh = orc.unit_gauge(L)
    Uh = orc.link_update(Uh, 0.3 * orc.gaussian_momenta(L, 899), 1.0, L)
    xi_h = orc.gaussian_spinor(orc.wilson_shape(L), 902) * np.sqrt(0.5)
    U1, P1, dH1 = _single(lq, Uh, xi_h, L, dtau, nsteps)

    n = int(np.prod(pe))
    lats = [lq.Lattice(L, pe, r) for r in range(n)]
    lq.link_local(lats)
    view = lambda a, lat, lead: lq.pegrid.local_view(a, lat.local_L, lat.origin, lead=lead)
    Us = [lq.Gaugefields(lat).upload(view(Uh, lat, 1)) for lat in lats]
    Ds = [lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC}) for U in Us]
    Dd = [D.adjoint() for D in Ds]
    ps, Gs = [lq.Gaugefields(lat) for lat in lats], [lq.Gaugefields(lat) for lat in lats]
    for p in ps:
        lq.gauss_distribution_(p, 901)                       # keyed by the global site: identical momenta for every decomposition
    xis = [lq.Fermionfields(lat, lq.WILSON).upload(view(xi_h, lat, 1)) for lat in lats]
    etas, Xs, Ys = [x.similar() for x in xis], [x.similar() for x in xis], [x.similar() for x in xis]
    lq.mdom_mul_(etas, Dd, xis)                              # eta = D' xi
    K = lambda: sum(lq.momentum_action(p) for p in ps)
    H0 = K() + (-BETA * 6 * np.prod(L) * lq.mdom_plaquette(Us)) + lq.mdom_dot(xis, xis).real
    for _ in range(nsteps):
        for U, p in zip(Us, ps):
            lq.U_update_(U, p, 0.5 * dtau)
        lq.mdom_P_update_(Us, ps, dtau, BETA)
        for X in Xs:
            lq.clear_fermion_(X)
        lq.mdom_solve_cg(Ds, Xs, etas, eps=1e-22)
        lq.mdom_mul_(Ys, Ds, Xs)
        lq.mdom_fermion_force_(Gs, Ds, Xs, Ys)
        for p, G in zip(ps, Gs):
            lq.Traceless_antihermitian_add_(p, dtau, G)
        for U, p in zip(Us, ps):
            lq.U_update_(U, p, 0.5 * dtau)
    for X in Xs:
        lq.clear_fermion_(X)
    lq.mdom_solve_cg(Ds, Xs, etas, eps=1e-22)
    H1 = K() + (-BETA * 6 * np.prod(L) * lq.mdom_plaquette(Us)) + lq.mdom_dot(etas, Xs).real
    for lat, U, p in zip(lats, Us, ps):
        assert rel_err(U.download(), view(U1, lat, 1)) < 1e-10
        assert rel_err(p.download(), view(P1, lat, 1)) < 1e-9
    assert abs((H1 - H0) - dH1) < 1e-7          # the same energy change (the start is not thermalised, so it is not small)


def test_wilson_clover_md_trajectory_self_partitioned_equals_one_domain(lq, orc, tmp_path):
    """BASELINE.json configs[3] in small: a 2-flavour Wilson-clover MD trajectory on a partitioned lattice (LQCD_FORCE_PARTITION = y,z,t
    with world-size-1 RCCL communicators: Dslash halos, clover sums by transport, X/Y faces of the hopping force, the halo-extended
    block of the clover force, ghost links and staple faces of the gauge force) against the same trajectory on one domain."""
    import os, subprocess, sys, textwrap
    L, dtau, nsteps, csw = (8, 4, 6, 8), 0.05, 3, 1.0
    Uh = orc.unit_gauge(L)
    Uh = orc.link_update(Uh, 0.3 * orc.gaussian_momenta(L, 911), 1.0, L)
    xi_h = orc.gaussian_spinor(orc.wilson_shape(L), 912) * np.sqrt(0.5)
    np.save(tmp_path / "U.npy", Uh)
    np.save(tmp_path / "xi.npy", xi_h)
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        d, L, dtau, nsteps, csw = sys.argv[1], (8, 4, 6, 8), 0.05, 3, 1.0
        Uh, xi_h = np.load(d + "/U.npy"), np.load(d + "/xi.npy")
        lat = lq.Lattice(L)
        if os.environ.get("LQCD_FORCE_PARTITION"):
            lat.comm_init(lq.comm_unique_id())
        U = lq.Gaugefields(lat).upload(Uh)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": 0.141139, "Clover_coefficient": csw,
                                        "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-22})
        fa = lq.FermiAction(D)
        p, G = lq.Gaugefields(lat), lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 913)
        xi = lq.Fermionfields(lat, lq.WILSON).upload(xi_h)
        eta = xi.similar()
        lq.sample_pseudofermions_(eta, U, fa, xi)
        H0 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, 5.7) + lq.dot(xi, xi).real
        for _ in range(nsteps):
            lq.U_update_(U, p, 0.5 * dtau)
            lq.P_update_(U, p, dtau, 5.7)
            lq.calc_UdSfdU_(G, fa, U, eta)
            lq.Traceless_antihermitian_add_(p, dtau, G)
            lq.U_update_(U, p, 0.5 * dtau)
        H1 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, 5.7) + lq.evaluate_FermiAction(fa, U, eta)
        np.save(d + "/out_" + sys.argv[2] + ".npy", np.concatenate([U.download().ravel(), p.download().ravel(), [H1 - H0]]))
        print("TRAJ_OK")
    """)
    res = {}
    for tag, mask in (("single", None), ("part", "14")):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("LQCD_FORCE_PARTITION", None)
        if mask:
            env["LQCD_FORCE_PARTITION"] = mask
            env["LQCD_HALO_STREAM_MODE"] = "3"      # the folded one-stream schedule (the unpartitioned run it is compared with has no halos)
        r = subprocess.run([sys.executable, "-c", code, str(tmp_path), tag], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "TRAJ_OK" in r.stdout, (tag, r.stdout[-2000:], r.stderr[-3000:])
        res[tag] = np.load(tmp_path / ("out_%s.npy" % tag))
    a, b = res["single"], res["part"]
    assert np.abs(a[:-1] - b[:-1]).max() < 1e-9 * np.abs(a[:-1]).max()
    assert abs(a[-1] - b[-1]) < 1e-7 and np.isfinite(a[-1])


def test_staggered_rhmc_md_trajectory_self_partitioned_equals_one_domain(lq, orc, tmp_path):
    """BASELINE.json configs[4] in small: a staggered rational (Nf = 2 and Nf = 3) MD trajectory on a partitioned lattice (LQCD_FORCE_PARTITION = y,z,t with
    world-size-1 RCCL communicators: Dslash halos inside the multi-shift CG, rank-summed inner products, X/Y faces of every pole's force sweep, ghost links and
    staple faces of the gauge force; the partial fractions are fitted inside the library) against the same trajectory on one domain -- and the mixed-precision
    action solver on the partitioned lattice against the fp64 one."""
    import os, subprocess, sys, textwrap
    L = (8, 4, 6, 8)
    Uh = orc.unit_gauge(L)
    Uh = orc.link_update(Uh, 0.3 * orc.gaussian_momenta(L, 921), 1.0, L)
    xi_h = orc.gaussian_spinor(orc.staggered_shape(L), 922) * np.sqrt(0.5)
    np.save(tmp_path / "U.npy", Uh)
    np.save(tmp_path / "xi.npy", xi_h)
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        d, tag, nf, mixed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        L, dtau, nsteps = (8, 4, 6, 8), 0.04, 3
        Uh, xi_h = np.load(d + "/U.npy"), np.load(d + "/xi.npy")
        lat = lq.Lattice(L)
        if os.environ.get("LQCD_FORCE_PARTITION"):
            lat.comm_init(lq.comm_unique_id())
        lat.set_param("mixed_action_solver", mixed)
        U = lq.Gaugefields(lat).upload(Uh)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": 0.5, "boundarycondition": (1, 1, 1, -1), "eps_CG": 1e-20})
        fa = lq.FermiAction(D, {"Nf": nf})
        assert fa.rational
        p, G = lq.Gaugefields(lat), lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 923)
        xi = lq.Fermionfields(lat, lq.STAGGERED).upload(xi_h)
        eta = xi.similar()
        lq.sample_pseudofermions_(eta, U, fa, xi)
        Sf0 = lq.evaluate_FermiAction(fa, U, eta)
        assert abs(Sf0 - lq.dot(xi, xi).real) < 1e-6 * Sf0          # the heat bath identity, through the fitted partial fractions
        H0 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, 5.7) + Sf0
        for _ in range(nsteps):
            lq.U_update_(U, p, 0.5 * dtau)
            lq.P_update_(U, p, dtau, 5.7)
            lq.calc_UdSfdU_(G, fa, U, eta)
            lq.Traceless_antihermitian_add_(p, dtau, G)
            lq.U_update_(U, p, 0.5 * dtau)
        H1 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, 5.7) + lq.evaluate_FermiAction(fa, U, eta)
        np.save(d + "/out_" + tag + ".npy", np.concatenate([U.download().ravel(), p.download().ravel(), eta.download().ravel(), [H1 - H0]]))
        print("TRAJ_OK")
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for nf in (2, 3):
        res = {}
        for tag, mask, mixed in (("single", None, 0), ("part", "14", 0), ("partmixed", "14", 1)):
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
            env.pop("LQCD_FORCE_PARTITION", None)
            if mask:
                env["LQCD_FORCE_PARTITION"] = mask
                env["LQCD_HALO_STREAM_MODE"] = "3"      # the folded one-stream schedule (the unpartitioned run it is compared with has no halos)
            r = subprocess.run([sys.executable, "-c", code, str(tmp_path), tag, str(nf), str(mixed)], capture_output=True, text=True, env=env, timeout=400, cwd=root)
            assert r.returncode == 0 and "TRAJ_OK" in r.stdout, (nf, tag, r.stdout[-2000:], r.stderr[-3000:])
            res[tag] = np.load(tmp_path / ("out_%s.npy" % tag))
        a, b, m = res["single"], res["part"], res["partmixed"]
        assert np.abs(a[:-1] - b[:-1]).max() < 1e-9 * np.abs(a[:-1]).max(), nf
        assert abs(a[-1] - b[-1]) < 1e-7 and np.isfinite(a[-1]), (nf, a[-1])
        assert np.abs(m[:-1] - b[:-1]).max() < 1e-7 * np.abs(b[:-1]).max() and abs(m[-1] - b[-1]) < 1e-5, nf
