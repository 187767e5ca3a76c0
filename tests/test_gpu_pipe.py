"""GPU parity of the persistent, software-pipelined Wilson kernel (dslash_pipe = 1, stencil.hip wilson_dirsplit_pipe) against the
oracle and, bit for bit, against the direction-split kernel it is derived from.  Small lattices reach the kernel through the test tunables
pipe_grid (number of persistent workgroups) and pipe_min_chunks (minimum chunks per workgroup); the geometry must have z-planes of whole
64-site chunks and t-slices that split over the 8 XCDs, otherwise the launcher falls back to variant 1 (also covered)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu
KAPPA = 0.141139
BC = (-1, 1, 1, -1)


@pytest.fixture(scope="module")
def gpu(lq):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return lq


def make(lq, orc, L, seed=31):
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, seed)
    Ud = lq.Gaugefields(lat).upload(Uh)
    D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC, "eps_CG": 1e-19, "MaxCGstep": 3000})
    return lat, Uh, Ud, D


def pipe_on(lat, grid, recon=12, mode=1):
    lat.set_param("dslash_variant", 1)
    lat.set_param("dslash_pipe", mode)
    lat.set_param("pipe_grid", grid)
    lat.set_param("pipe_min_chunks", 1)
    lat.set_param("gauge_recon", recon)


# (lattice, persistent grid): XH = 8 / 4 / 16 / 8 (two chunks per plane) / 6 (not a power of two: magic-number division by XH per lane)
CASES = [((16, 8, 8, 4), 8), ((8, 16, 8, 8), 16), ((32, 4, 8, 4), 8), ((16, 16, 16, 4), 24), ((12, 32, 8, 2), 8), ((16, 8, 8, 4), 64)]


@pytest.mark.parametrize("L,grid", CASES)
@pytest.mark.parametrize("recon", [12, 18])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pipe_dslash_matches_oracle_and_variant1_bitwise(gpu, orc, L, grid, recon, mode):
    lq = gpu
    lat, Uh, Ud, D = make(lq, orc, L)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 32)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y9, y1 = x.similar(), x.similar()
    for dagger in (False, True):
        op = D.adjoint() if dagger else D
        pipe_on(lat, grid, recon, mode)
        lq.mul_(y9, op, x)
        assert lat.get_param("recon_active") == (1 if recon == 12 else 0)
        lat.set_param("dslash_pipe", 0)
        lq.mul_(y1, op, x)
        ref = orc.wilson_D(Uh, psi, L, KAPPA, 1.0, BC, dagger)
        assert rel_err(y9.download(), ref) < 1e-13, (L, dagger)
        assert np.array_equal(y9.download(), y1.download()), (L, dagger, "variant 9 differs from variant 1")


@pytest.mark.parametrize("L,grid", [((16, 8, 8, 4), 8), ((16, 16, 16, 4), 16)])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pipe_parity_hops_and_map_settings(gpu, orc, L, grid, mode):
    lq = gpu
    lat, Uh, Ud, D = make(lq, orc, L, seed=33)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 34)
    for nsub, ysplit in ((8, 1), (16, 2), (16, 4), (8, 2)):
        pipe_on(lat, grid, mode=mode)
        lat.set_param("xcd_nsub", nsub)
        lat.set_param("xcd_ysplit", ysplit)
        for dagger in (False, True):
            for out_sub, in_sub, p in ((lq.EVEN, lq.ODD, 0), (lq.ODD, lq.EVEN, 1)):
                xin = lq.Fermionfields(lat, lq.WILSON, in_sub).upload(psi)
                yout = lq.Fermionfields(lat, lq.WILSON, out_sub)
                lq.hop_(yout, D.adjoint() if dagger else D, xin)
                assert rel_err(yout.download(), orc.wilson_hop_parity(Uh, psi, L, 1.0, BC, dagger, p)) < 1e-13, (nsub, ysplit, dagger, p)
        x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
        y = x.similar()
        lq.mul_(y, D, x)
        assert rel_err(y.download(), orc.wilson_D(Uh, psi, L, KAPPA, 1.0, BC, False)) < 1e-13, (nsub, ysplit)


@pytest.mark.parametrize("L,grid", [((16, 8, 8, 4), 8), ((12, 32, 8, 2), 16)])
@pytest.mark.parametrize("mode", [1, 2, 3])
def test_pipe_cg_matches_oracle(gpu, orc, L, grid, mode):
    """fused CG on the persistent kernel: |Dp|^2 partials (one per persistent workgroup), update-mode D^+, deferred x"""
    lq = gpu
    lat, Uh, Ud, D = make(lq, orc, L, seed=35)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 36)
    b = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, psi, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st == 0
    for fused in (2, 1, 0):
        for defer in (1, 0):
            pipe_on(lat, grid, mode=mode)
            lat.set_param("cg_fused", fused)
            lat.set_param("cg_defer_x", defer)
            sol = b.similar()
            it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), b, return_info=True)
            assert abs(it - ito) <= 1 and rr < 1e-19 and rel_err(sol.download(), xo) < 1e-9, (fused, defer, it, ito)
    if mode == 2:       # same |.|^2 partial per workgroup as variant 1: the CG iterates are the same bits
        pipe_on(lat, grid, mode=2)
        lat.set_param("cg_fused", 2); lat.set_param("cg_defer_x", 1); lat.set_param("cg_small", 0)
        s2 = b.similar()
        lq.solve_DinvX_(s2, lq.DdagD_operator(D), b)
        lat.set_param("dslash_pipe", 0)
        s1 = b.similar()
        lq.solve_DinvX_(s1, lq.DdagD_operator(D), b)
        lat.set_param("cg_small", 1)
        assert np.array_equal(s1.download(), s2.download())
    # mixed-precision CG: the fp32 build of the same kernel as the inner operator, true fp64 residual as the stopping rule
    pipe_on(lat, grid, mode=mode)
    lat.set_param("cg_fused", 2)
    lat.set_param("cg_defer_x", 1)
    sol = b.similar()
    info = lq.solve_mixed_DinvX_(sol, lq.DdagD_operator(D), b, return_info=True)
    assert rel_err(sol.download(), xo) < 1e-9, info


def test_pipe_falls_back_where_the_geometry_does_not_fit(gpu, orc):
    """8^4: a z-plane is half a chunk -> variant 1 runs (same answer, same number of |.|^2 partials as variant 1)"""
    lq = gpu
    L = (8, 8, 8, 8)
    lat, Uh, Ud, D = make(lq, orc, L, seed=37)
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 38)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    pipe_on(lat, 8)
    lq.mul_(y, D, x)
    assert rel_err(y.download(), orc.wilson_D(Uh, psi, L, KAPPA, 1.0, BC, False)) < 1e-13
    sol = x.similar()
    it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, psi, L, KAPPA, 1.0, BC, eps=1e-19)
    assert st == 0 and abs(it - ito) <= 1 and rel_err(sol.download(), xo) < 1e-9


def test_pipe_rccl_self_partition(gpu, orc):
    """partitioned directions: the persistent interior kernel multiplies off-rank hops by sign 0, pack / exchange / exterior add them"""
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, K, BC = (16, 8, 8, 4), 0.141139, (1, 1, 1, -1)
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        lat.set_param("dslash_pipe", 1); lat.set_param("pipe_grid", 8); lat.set_param("pipe_min_chunks", 1)
        U = orc.hot_gauge(L, 111)
        Ud = lq.Gaugefields(lat).upload(U)
        D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": K, "boundarycondition": BC, "eps_CG": 1e-19})
        psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 112)
        x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
        y = x.similar()
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            ref = orc.wilson_D(U, psi, L, K, 1.0, BC, dag)
            err = np.abs(y.download() - ref).max() / np.abs(ref).max()
            assert err < 1e-13, (dag, err)
        xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, U, psi, L, K, 1.0, BC, eps=1e-19)
        its = {}
        for pipe in (1, 2, 3, 0):
            lat.set_param("dslash_pipe", pipe)
            sol = x.similar()
            its[pipe], rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
            assert np.abs(sol.download() - xo).max() / np.abs(xo).max() < 1e-9, (pipe, its, ito)
        assert st == 0 and all(abs(v - ito) <= 1 for v in its.values()), (its, ito)
        # round 4: the scalar-addressing kernel on the 18 stored reals and on "12 + delta" links (a field at a text file's precision), partitioned
        lat.set_param("dslash_pipe", 2)
        lat.set_param("gauge_recon", 18)
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            ref = orc.wilson_D(U, psi, L, K, 1.0, BC, dag)
            assert lat.get_param("recon_active") == 0 and np.abs(y.download() - ref).max() / np.abs(ref).max() < 1e-13, ("s18", dag)
        lat.set_param("gauge_recon", 12)
        rng = np.random.default_rng(7)
        Up = U + 1e-10 * (rng.standard_normal(U.shape) + 1j * rng.standard_normal(U.shape)) / 3.0
        Ud.upload(Up)
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            ref = orc.wilson_D(Up, psi, L, K, 1.0, BC, dag)
            # (the folded halo schedule has no "12 + delta" instance: it reads the 18 stored reals of such a field)
            want = 0 if lat.get_param("halo_fold_active") else 2
            assert lat.get_param("recon_active") == want and np.abs(y.download() - ref).max() / np.abs(ref).max() < 1e-13, ("delta", dag, lat.get_param("recon_active"), want)
        print("PIPE_SELF_OK")
    """)
    for mask in ("8", "14", "15"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "PIPE_SELF_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])
