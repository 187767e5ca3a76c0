"""One-launch CG for launch-bound lattices (csrc/cg_persist.hip, tunable cg_persist; BASELINE configs[1]: 8^4 staggered CG to 1e-10): the whole
solve in one kernel with two grid-wide synchronisations per iteration.  Same algorithm as the launch chain (fused CG, reference's absolute
stopping rule), different summation order inside a site and across the lattice -- so: same iteration count (+-1), solutions equal to
rounding, the oracle's solution, true residual recomputed independently."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel_err(a, b):
    return np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)


def _setup(lq, orc, L, seed, mass=0.1, bc=(1, 1, 1, -1), eps=1e-18):
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, seed)
    U = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Staggered", "mass": mass, "boundarycondition": bc, "eps_CG": eps, "MaxCGstep": 3000})
    bh = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), seed + 1)
    b = lq.Fermionfields(lat, lq.STAGGERED).upload(bh)
    return lat, Uh, U, D, bh, b


# 4 ... 256 workgroups; x extents with XH = 2, 3, 4, 6, 8 (rows of a chunk straddle y / z / t), odd chunk counts per parity
@pytest.mark.parametrize("L", [(4, 4, 4, 4), (8, 8, 8, 8), (8, 8, 8, 16), (16, 8, 8, 16), (8, 4, 4, 8), (6, 4, 4, 8), (12, 4, 4, 4), (4, 4, 4, 12)])
@pytest.mark.parametrize("bc", [(1, 1, 1, -1), (1, 1, 1, 1), (-1, 1, -1, 1)])
def test_one_launch_cg_equals_the_launch_chain(lq, orc, L, bc):
    lat, Uh, U, D, bh, b = _setup(lq, orc, L, 2101, bc=bc)
    A = lq.DdagD_operator(D)
    out = {}
    for mode in (1, 0):
        lat.set_param("cg_persist", mode)
        x = b.similar()
        it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
        out[mode] = (it, rr, x.download())
        # true residual, recomputed with the operator kernels
        y = b.similar()
        lq.mul_(y, A, x)
        lq.add_fermion_(y, -1.0, b)
        assert lq.dot(y, y).real < 2e-18, (mode, lq.dot(y, y).real)
    assert abs(out[1][0] - out[0][0]) <= max(1, out[0][0] // 50) and out[1][1] < 1e-18      # hundreds of iterations: rounding moves the count by a few
    assert rel_err(out[1][2], out[0][2]) < 1e-9


def test_one_launch_cg_matches_oracle(lq, orc):
    L, bc, mass = (4, 4, 4, 8), (1, 1, 1, -1), 0.2
    lat, Uh, U, D, bh, b = _setup(lq, orc, L, 2111, mass=mass, bc=bc)
    assert lat.get_param("cg_persist") == 1
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    xo, ito, rro = orc.cg_DdagD(orc.STAGGERED, Uh, bh, L, mass, 1.0, bc, eps=1e-18)[:3]
    assert abs(it - ito) <= 1 and rel_err(x.download(), xo) < 1e-9


def test_one_launch_cg_initial_guess_window_and_exhaustion(lq, orc):
    L = (8, 8, 8, 8)
    lat, Uh, U, D, bh, b = _setup(lq, orc, L, 2121)
    A = lq.DdagD_operator(D)
    x = b.similar()
    it0, _ = lq.solve_DinvX_(x, A, b, return_info=True)
    sol = x.download()
    # a converged start: zero iterations; a nearby start: fewer iterations, same solution
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
    assert it == 0 and rr < 1e-18
    guess = lq.Fermionfields(lat, lq.STAGGERED).upload(sol * (1.0 + 1e-6))
    it, rr = lq.solve_DinvX_(guess, A, b, return_info=True)
    assert 0 < it < it0 and rel_err(guess.download(), sol) < 1e-9
    # fixed-length window = the launch chain's window to rounding
    wins = []
    for mode in (1, 0):
        lat.set_param("cg_persist", mode)
        xw = b.similar()
        lq.lib.check(lq.lib.lib().lqcd_solve_cg_DdagD_fixed(D._h, xw._h, b._h, 7))
        wins.append(xw.download())
    assert rel_err(wins[0], wins[1]) < 1e-12
    # an exhausted solve reports NotConverged like the chain does
    lat.set_param("cg_persist", 1)
    D.MaxCGstep = 3
    with pytest.raises(lq.NotConverged):
        lq.solve_DinvX_(b.similar(), lq.DdagD_operator(D), b)


def test_one_launch_cg_does_not_apply_where_it_must_not(lq, orc):
    """Wilson operators, lattices above 256 chunks and the non-default iteration forms keep the launch chain (same answers either way)."""
    lat, Uh, U, D, bh, b = _setup(lq, orc, (16, 16, 16, 16), 2131)      # 1024 chunks
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, lq.DdagD_operator(D), b, return_info=True)
    assert rr < 1e-18
    lat2, Uh2, U2, D2, bh2, b2 = _setup(lq, orc, (8, 8, 8, 8), 2141)
    ref = None
    for key in (None, "cg_small", "cg_fused"):
        if key:
            lat2.set_param(key, 0)
        x2 = b2.similar()
        lq.solve_DinvX_(x2, lq.DdagD_operator(D2), b2)
        if ref is None:
            ref = x2.download()
        else:
            assert rel_err(x2.download(), ref) < 1e-9


def test_one_launch_cg_that_gives_up_falls_back_to_the_chain(lq, orc):
    """cg_persist = 2 (test hook): the synchronisations wait for one workgroup more than the launch has, every workgroup gives up after the
    50 ms bound, x is left untouched, the solve is repeated by the launch chain and the context stops asking for the one-launch form."""
    lat, Uh, U, D, bh, b = _setup(lq, orc, (8, 8, 8, 8), 2151)
    A = lq.DdagD_operator(D)
    x0 = b.similar()
    lat.set_param("cg_persist", 0)
    it0, rr0 = lq.solve_DinvX_(x0, A, b, return_info=True)
    lat.set_param("cg_persist", 2)
    x = b.similar()
    it, rr = lq.solve_DinvX_(x, A, b, return_info=True)
    assert lat.get_param("cg_persist") == 0
    assert it == it0 and np.array_equal(x.download(), x0.download())
    lat.set_param("cg_persist", 1)                    # and it works again afterwards (counters are re-zeroed)
    x2 = b.similar()
    it2, rr2 = lq.solve_DinvX_(x2, A, b, return_info=True)
    assert abs(it2 - it0) <= 1 and rel_err(x2.download(), x0.download()) < 1e-9
