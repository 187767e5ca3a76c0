"""Partitioned fused CG with the folded halo schedules (tunable halo_fold: the boundary hops are taken from the ghost buffers inside the stencil launch, no
exterior kernel -- round 5: schedule 3, one launch behind the exchange; round 6: every schedule, as a bulk launch beside / in front of the exchange and a boundary
launch behind it, incl. the new one-stream schedule 4 and an exchange that completes late, halo_inject_us) and the fused tails of the exterior / update launches (tunable halo_fuse): bit 0 = the
exterior kernel's last block sums the |.|^2 partials (no reduce_final launch), bit 1 = the exterior of D p packs the faces D^+ needs and the
x/p update packs the new search direction (no pack launches).  Run through the real RCCL path on one GPU (self-partition, world-size-1
communicators).  Pre-packed faces carry the bits a pack launch would have produced, so bit 1 alone must not change a single bit of the
solution; bit 0 changes the summation order of the norms (iterates differ in rounding only)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

CODE = textwrap.dedent("""
    import os, sys, numpy as np
    sys.path.insert(0, os.getcwd())
    import latticeqcd_jl_amd as lq
    from oracle import oracle as orc
    L, K, BC = %s, 0.141139, (1, 1, 1, -1)
    FOLDS = %d          # the folded halo schedule applies to this lattice / partition mask
    lat = lq.Lattice(L)
    lat.comm_init(lq.comm_unique_id())
    U = orc.hot_gauge(L, 111)
    Ud = lq.Gaugefields(lat).upload(U)
    D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": K, "boundarycondition": BC, "eps_CG": 1e-19})
    psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 112)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, U, psi, L, K, 1.0, BC, eps=1e-19)
    assert st == 0
    sols = {}
    folded = 0
    y = x.similar()
    for mode in (-1, 0, 1, 2, 3, 4):
      for hfold in (1, 0):
        lat.set_param("halo_fold", hfold)
        lat.set_param("halo_inject_us", 30 if (mode in (0, 4) and hfold) else 0)      # an exchange that completes late (the stand-in for real links) changes nothing but the time
        for fold in (1, 0):
            for fuse in (0, 1, 2, 3):
                lat.set_param("halo_stream_mode", mode); lat.set_param("cg_fold_scalars", fold); lat.set_param("halo_fuse", fuse)
                sol = x.similar()
                it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
                folded += lat.get_param("halo_fold_active")
                s = sol.download()
                err = np.abs(s - xo).max() / np.abs(xo).max()
                assert abs(it - ito) <= 1 and rr < 1e-19 and err < 1e-9, (mode, hfold, fold, fuse, it, ito, rr, err)
                sols[(mode, hfold, fold, fuse)] = s
            if mode == -1:      # the tuner may pick one schedule for one solve and another for the next: equal to rounding only
                continue
            assert np.array_equal(sols[(mode, hfold, fold, 0)], sols[(mode, hfold, fold, 2)]), ("pre-packed faces changed the solution", mode, hfold, fold)
            assert np.array_equal(sols[(mode, hfold, fold, 1)], sols[(mode, hfold, fold, 3)]), ("pre-packed faces changed the solution", mode, hfold, fold)
        # the folded schedule's Dslash (boundary hops from the ghost buffers inside the stencil launch) against the oracle, D and D^+
        lat.set_param("halo_stream_mode", mode)
        for dag in (False, True):
            lq.mul_(y, D.adjoint() if dag else D, x)
            ref = orc.apply_D(lq.WILSON, U, psi, L, K, 1.0, BC, dag)
            e = np.abs(y.download() - ref).max() / np.abs(ref).max()
            assert e < 1e-13, (mode, hfold, dag, e)
        assert lat.get_param("halo_fold_active") == (1 if hfold and FOLDS else 0), ("the folded launches did not run where they apply", mode, hfold)
    assert (folded > 0) == bool(FOLDS), (folded, FOLDS)
    lat.set_param("halo_fold", 1); lat.set_param("halo_inject_us", 0)
    # the staggered operator: the fused reduction only (unfolded schedules), and the folded twin of its direction-split kernel (every schedule)
    Ds = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Staggered", "mass": 0.5, "boundarycondition": BC, "eps_CG": 1e-19})
    ps = orc.gaussian_spinor(lat.fermion_shape(lq.STAGGERED), 113)
    xs = lq.Fermionfields(lat, lq.STAGGERED).upload(ps)
    ys = xs.similar()
    xo, ito, rro, st = orc.cg_DdagD(orc.STAGGERED, U, ps, L, 0.5, 1.0, BC, eps=1e-19)
    for mode, hfold in ((0, 1), (3, 0), (3, 1), (4, 1), (1, 1), (2, 1), (2, 0)):
        lat.set_param("halo_stream_mode", mode); lat.set_param("halo_fold", hfold)
        for dag in (False, True):
            lq.mul_(ys, Ds.adjoint() if dag else Ds, xs)
            ref = orc.apply_D(lq.STAGGERED, U, ps, L, 0.5, 1.0, BC, dag)
            e = np.abs(ys.download() - ref).max() / np.abs(ref).max()
            assert e < 1e-13, ("staggered", mode, hfold, dag, e)
        assert lat.get_param("halo_fold_active") == (1 if hfold else 0)
        for fuse in (0, 3):
            lat.set_param("halo_fuse", fuse)
            sol = xs.similar()
            it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(Ds), xs, return_info=True)
            assert st == 0 and abs(it - ito) <= 1 and np.abs(sol.download() - xo).max() / np.abs(xo).max() < 1e-9, (mode, hfold, fuse, it, ito)
    lat.set_param("halo_stream_mode", -1)
    # general r (two r = 1 passes per application): no pre-packing, same answer
    Dr = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": 0.12, "r": 0.8, "boundarycondition": BC, "eps_CG": 1e-19})
    xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, U, psi, L, 0.12, 0.8, BC, eps=1e-19)
    lat.set_param("halo_fuse", 3)
    sol = x.similar()
    it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(Dr), x, return_info=True)
    assert st == 0 and abs(it - ito) <= 1 and np.abs(sol.download() - xo).max() / np.abs(xo).max() < 1e-9, (it, ito)
    print("HALO_FUSE_OK")
""")


# Every partitioned lattice folds (stencil.hip halo_fold_applies).  Which kernel: the scalar-addressing FOLD instances where z-planes are whole chunks (XH * LY a
# multiple of 64), the chunks of a t-slice split over 8 XCDs and x is unpartitioned -- (16,8,8,4) masks 14 / 8, (16,16,8,4), (32,4,8,6); the folded twins of the
# direction-split kernels everywhere else -- (8,4,6,8) and every mask with the x bit.
@pytest.mark.parametrize("L,mask,folds", [((8, 4, 6, 8), "8", 1), ((8, 4, 6, 8), "14", 1), ((8, 4, 6, 8), "15", 1), ((16, 8, 8, 4), "14", 1), ((16, 8, 8, 4), "8", 1),
                                          ((16, 8, 8, 4), "15", 1), ((16, 16, 8, 4), "12", 1), ((32, 4, 8, 6), "6", 1)])
def test_fused_tails_on_the_rccl_path(lq, L, mask, folds):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CODE % (L, folds)], capture_output=True, text=True, env=env, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "HALO_FUSE_OK" in r.stdout, (mask, r.stdout[-1500:], r.stderr[-3000:])
