"""Parity tests of the opt-in Wilson kernel variants 2-8 (experiments/stencil_alt/stencil_alt.hip): NOT part of the product's test suite.
Build the experiment library first and point the binding at it:

    LQCD_VARIANTS=1 bash latticeqcd.jl_amd/csrc/build.sh          # -> latticeqcd.jl_amd/csrc/liblqcd_hip_variants.so
    LQCD_HIP_LIB=$PWD/latticeqcd.jl_amd/csrc/liblqcd_hip_variants.so python -m pytest experiments/stencil_alt/test_variants.py -q

dslash_variant = 2: eight waves per 64 sites (one per hop), LDS combine; 3: persistent hop split; 4: lane split (four directions in the 16-lane rows of one
wave, v_permlane reduce-scatter, no LDS); 5: direction split with 36 KiB of LDS and the registers of four workgroups per CU; 6: x / y neighbour spinors staged
through LDS behind a mid-kernel barrier; 7: both parities of a chunk in one 512-thread workgroup; 8: both hops of a direction in flight."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_err  # noqa: E402
from test_gpu_parity import DSLASH_TOL, KAPPA, host_spinor, setup  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import latticeqcd_jl_amd as lq
    assert lq.lib.device_count() > 0
    return lq


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle as o
    o.build()
    return o


@pytest.mark.parametrize("L", [(4, 4, 4, 4), (8, 4, 6, 2), (16, 8, 4, 4), (6, 2, 2, 2)])
@pytest.mark.parametrize("dagger", [False, True])
@pytest.mark.parametrize("remap", [0, 1, 2])
@pytest.mark.parametrize("variant", [2, 3, 4, 5, 6, 7, 8])
def test_wilson_variant_matches_oracle(gpu, orc, L, dagger, remap, variant):
    lq = gpu
    lat, Uh, Ud, D = setup(lq, orc, L, lq.WILSON, seed=19, bc=(-1, 1, 1, -1))
    if not lat.get_param("variants_built"):
        pytest.fail("this library was built without the variants: LQCD_VARIANTS=1 bash latticeqcd.jl_amd/csrc/build.sh, then set LQCD_HIP_LIB")
    lat.set_param("dslash_variant", variant)
    lat.set_param("xcd_remap", remap)
    psi = host_spinor(orc, lat, lq.WILSON, 20)
    x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
    y = x.similar()
    lq.mul_(y, D.adjoint() if dagger else D, x)
    assert rel_err(y.download(), orc.wilson_D(Uh, psi, L, KAPPA, 1.0, (-1, 1, 1, -1), dagger)) < DSLASH_TOL
    for out_sub, in_sub, p in ((lq.EVEN, lq.ODD, 0), (lq.ODD, lq.EVEN, 1)):
        xin = lq.Fermionfields(lat, lq.WILSON, in_sub).upload(psi)
        yout = lq.Fermionfields(lat, lq.WILSON, out_sub)
        lq.hop_(yout, D.adjoint() if dagger else D, xin)
        assert rel_err(yout.download(), orc.wilson_hop_parity(Uh, psi, L, 1.0, (-1, 1, 1, -1), dagger, p)) < DSLASH_TOL
    if not dagger:
        sol = x.similar()
        it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
        xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, Uh, psi, L, KAPPA, 1.0, (-1, 1, 1, -1), eps=1e-19)
        assert st == 0 and abs(it - ito) <= 1 and rel_err(sol.download(), xo) < 1e-9
