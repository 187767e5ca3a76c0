"""hipGraph replay of the CG iteration bursts (tunable "graph"): same launches, so bit-identical results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["Wilson", "Staggered"])
def test_cg_with_graph_replay_is_bit_identical(lq, name):
    assert lq.lib.device_count() > 0
    L = (8, 8, 8, 8)
    kind = lq.WILSON if name == "Wilson" else lq.STAGGERED
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=5)
    lat = U.lattice
    lat.set_param("cg_persist", 0)       # graph replay is a form of the launch chain (the one-launch CG: tests/test_gpu_cg_persist.py)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": 0.141139, "mass": 0.5, "eps_CG": 1e-19})
    A = lq.DdagD_operator(D)
    b = lq.Fermionfields(lat, kind)
    lq.gauss_distribution_fermion_(b, 6)
    x = b.similar()
    res = []
    for g in (0, 1):
        lat.set_param("graph", g)
        lq.clear_fermion_(x)
        it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
        res.append((it, rr, x.download()))
    assert res[0][0] == res[1][0] and res[0][1] == res[1][1] and np.array_equal(res[0][2], res[1][2])
    D.MaxCGstep = 11                                   # a burst shorter than the captured one falls back to plain launches
    lq.clear_fermion_(x)
    with pytest.raises(lq.NotConverged):
        lq.solve_DinvX_(x, lq.DdagD_operator(D), b)
    lat.set_param("graph", 0)
