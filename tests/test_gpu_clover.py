"""GPU parity of the Wilson-clover operator against the oracle's textbook definition."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

KAPPA, CSW = 0.141139, 1.5612
BC = (1, 1, 1, -1)


@pytest.mark.parametrize("L", [(4, 4, 4, 4), (8, 4, 6, 2), (6, 6, 4, 2)])       # the last one: a partially filled chunk
def test_wilson_clover_matches_oracle(lq, orc, L):
    assert lq.lib.device_count() > 0
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 501)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW, "boundarycondition": BC,
                                    "eps_CG": 1e-19})
    A = orc.clover_build(Uh, L, KAPPA, CSW)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 502)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    for dag in (False, True):
        lq.mul_(y, D.adjoint() if dag else D, x)
        assert rel_err(y.download(), orc.wilson_clover_D(Uh, A, psi, L, KAPPA, 1.0, BC, dag)) < 1e-13
    # D'D and the CG on it
    lq.mul_(y, lq.DdagD_operator(D), x)
    ref = orc.wilson_clover_D(Uh, A, orc.wilson_clover_D(Uh, A, psi, L, KAPPA, 1.0, BC), L, KAPPA, 1.0, BC, True)
    assert rel_err(y.download(), ref) < 1e-13
    sol = x.similar()
    it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
    xo, ito, rro, st = orc.cg_clover(Uh, A, psi, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st == 0 and abs(it - ito) <= 1 and rel_err(sol.download(), xo) < 1e-9
    # BiCGStab on D_sw
    D.method_CG = "bicgstab"
    lq.clear_fermion_(sol)
    lq.solve_DinvX_(sol, D, x)
    lq.mul_(y, D, sol)
    lq.add_fermion_(y, -1.0, x)
    assert lq.dot(y, y).real < 1e-18
    # the term follows the links: new links in the same handle
    Uh2 = orc.hot_gauge(L, 503)
    U.upload(Uh2)
    lq.mul_(y, D, x)
    assert rel_err(y.download(), orc.wilson_clover_D(Uh2, orc.clover_build(Uh2, L, KAPPA, CSW), psi, L, KAPPA, 1.0, BC)) < 1e-13
    # mixed-precision CG (fp32 links and fp32 clover blocks inside, fp64 defect correction): same stopping rule on the true residual
    A2 = orc.clover_build(Uh2, L, KAPPA, CSW)
    lq.clear_fermion_(sol)
    itm, outer, rrm = lq.solve_mixed_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
    xo2, _, _, st2 = orc.cg_clover(Uh2, A2, psi, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st2 == 0 and rrm < 1e-19 and outer >= 2 and rel_err(sol.download(), xo2) < 1e-8
    lq.mul_(y, lq.DdagD_operator(D), sol)
    lq.add_fermion_(y, -1.0, x)
    assert lq.dot(y, y).real < 1e-19
    # even-odd preconditioned BiCGStab with the inverse clover blocks: same algorithm as the oracle's, D_sw and D_sw^+
    D.method_CG = "bicgstab_evenodd"
    for dag in (False, True):
        Dd = D.adjoint() if dag else D
        lq.clear_fermion_(sol)
        ite, rre = lq.solve_DinvX_(sol, Dd, x, return_info=True)
        xo3, ito3, _, st3 = orc.wilson_clover_bicgstab_eo(Uh2, A2, psi, L, KAPPA, 1.0, BC, dag, eps=1e-19)
        assert st3 == 0 and abs(ite - ito3) <= 2 and rel_err(sol.download(), xo3) < 1e-9
        lq.mul_(y, Dd, sol)
        lq.add_fermion_(y, -1.0, x)
        assert lq.dot(y, y).real < 1e-17
    # fermion force of S_f = phi^+ (D_sw^+ D_sw)^-1 phi: hopping part + derivative of the clover term, against the oracle's scatter
    D.eps_CG = 1e-22
    G = lq.Gaugefields(lat)
    Sf = lq.calc_UdSfdU_(G, lq.FermiAction(D), U, x)
    So, Go, Xo, Yo = orc.clover_fermion_force(Uh2, A2, psi, L, KAPPA, CSW, 1.0, BC, eps=1e-22)
    assert abs(Sf - So) < 1e-10 * So and rel_err(G.download(), Go) < 1e-9
    # the sweep alone from resident X, Y, scaled and accumulated
    X, Y = lq.Fermionfields(lat, lq.WILSON).upload(Xo), lq.Fermionfields(lat, lq.WILSON).upload(Yo)
    lq.fermion_force_(G, D, X, Y)
    assert rel_err(G.download(), Go) < 1e-13
    lq.fermion_force_(G, D, X, Y, scale=0.5, accumulate=True)
    assert rel_err(G.download(), 1.5 * Go) < 1e-13


def test_clover_coefficient_zero_is_wilson(lq, orc):
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 504)
    U = lq.Gaugefields(lat).upload(Uh)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 505)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y, z = x.similar(), x.similar()
    Dw = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": KAPPA})
    Dc = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": 0.0})
    lq.mul_(y, Dw, x)
    lq.mul_(z, Dc, x)
    assert np.array_equal(y.download(), z.download())


@pytest.mark.parametrize("L", [(8, 4, 6, 4), (6, 6, 4, 2)])
def test_clover_sums_by_plaquette_transport_equal_the_direct_leaves(lq, orc, L):
    """The partitioned build (plaquettes + two backward transports) run on an unpartitioned lattice (tunable clover_transport)."""
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 506)
    U = lq.Gaugefields(lat).upload(Uh)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 507)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    lat.set_param("clover_transport", 1)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": CSW, "boundarycondition": BC,
                                    "eps_CG": 1e-22})
    lq.mul_(y, D, x)
    A = orc.clover_build(Uh, L, KAPPA, CSW)
    ref = orc.wilson_clover_D(Uh, A, psi, L, KAPPA, 1.0, BC)
    assert rel_err(y.download(), ref) < 1e-13
    # the same tunable sends the clover force through the halo-extended block (the partitioned path; halos by local wrap here)
    G = lq.Gaugefields(lat)
    lq.calc_UdSfdU_(G, lq.FermiAction(D), U, x)
    _, Go, _, _ = orc.clover_fermion_force(Uh, A, psi, L, KAPPA, CSW, 1.0, BC, eps=1e-22)
    assert rel_err(G.download(), Go) < 1e-9


def test_rccl_self_partition_clover(lq, orc):
    """Wilson-clover on a partitioned lattice (BASELINE.json configs[3]): LQCD_FORCE_PARTITION + world-size-1 RCCL communicators run the
    link-ghost exchange, the two matrix-face exchanges of the clover sums and the halo'ed Dslash with the fused clover epilogue exactly
    as at N > 1; operator, CG, even-odd BiCGStab, the mixed-precision CG and the fermion force must equal the oracle."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, K, CSW, BC = (8, 4, 6, 8), 0.141139, 1.5612, (1, 1, 1, -1)
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        Uh = orc.hot_gauge(L, 111)
        U = lq.Gaugefields(lat).upload(Uh)
        A = orc.clover_build(Uh, L, K, CSW)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": K, "Clover_coefficient": CSW, "boundarycondition": BC, "eps_CG": 1e-19})
        psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 112)
        x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
        y, sol = x.similar(), x.similar()
        rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            assert rel(y.download(), orc.wilson_clover_D(Uh, A, psi, L, K, 1.0, BC, dag)) < 1e-13, dag
        xo, ito, _, st = orc.cg_clover(Uh, A, psi, L, K, 1.0, BC, eps=1e-19)
        it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
        assert st == 0 and abs(it - ito) <= 1 and rel(sol.download(), xo) < 1e-9
        lq.clear_fermion_(sol)
        lq.solve_mixed_DinvX_(sol, lq.DdagD_operator(D), x)
        assert rel(sol.download(), xo) < 1e-8
        D.method_CG = "bicgstab_evenodd"
        lq.clear_fermion_(sol)
        lq.solve_DinvX_(sol, D, x)
        xe, _, _, st = orc.wilson_clover_bicgstab_eo(Uh, A, psi, L, K, 1.0, BC, False, eps=1e-19)
        assert st == 0 and rel(sol.download(), xe) < 1e-9
        D.eps_CG = 1e-22
        G = lq.Gaugefields(lat)
        lq.calc_UdSfdU_(G, lq.FermiAction(D), U, x)                 # hopping force with its X, Y face exchange + clover force through
        _, Go, _, _ = orc.clover_fermion_force(Uh, A, psi, L, K, CSW, 1.0, BC, eps=1e-22)       # the halo-extended links / Lambda block
        assert rel(G.download(), Go) < 1e-9
        print("RCCL_SELF_CLOVER_OK")
    """)
    for mask in ("8", "14", "15"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "RCCL_SELF_CLOVER_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])


def test_changing_csw_invalidates_the_inverse_clover_blocks(lq, orc):
    """lqcd_op_set_clover(csw2) on an operator that already holds A^-1 for csw1 (same links) must rebuild A^-1: the
    even-odd solver would otherwise converge silently to the solution of the wrong Schur system."""
    import ctypes as C
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 801)
    U = lq.Gaugefields(lat).upload(Uh)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 802)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    sol, y = x.similar(), x.similar()
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover", "κ": KAPPA, "Clover_coefficient": 1.0, "boundarycondition": BC,
                                    "eps_CG": 1e-19, "method_CG": "bicgstab_evenodd"})
    lq.solve_DinvX_(sol, D, x)                                   # builds A^-1 for csw = 1
    lq.lib.check(lq.lib.lib().lqcd_op_set_clover(D._h, C.c_double(1.7)))      # same links, new coefficient
    lq.clear_fermion_(sol)
    lq.solve_DinvX_(sol, D, x)
    xo, _, _, st = orc.wilson_clover_bicgstab_eo(Uh, orc.clover_build(Uh, L, KAPPA, 1.7), psi, L, KAPPA, 1.0, BC, False, eps=1e-19)
    assert st == 0 and rel_err(sol.download(), xo) < 1e-9
    lq.mul_(y, D, sol)                                            # true residual with the csw = 1.7 operator
    lq.add_fermion_(y, -1.0, x)
    assert lq.dot(y, y).real < 1e-17
