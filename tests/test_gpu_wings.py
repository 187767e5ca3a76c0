"""Host arrays with the reference's wing (Nwing of Initialize_Gaugefields, universe.jl:41-49; staggered fields of :107 are created
without nowing = true): only the interior travels, bit-exactly, and a download leaves the wings of the caller's array alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w", [1, 2])
def test_winged_gauge_and_spinor_round_trip(lq, orc, w):
    assert lq.lib.device_count() > 0
    L = (4, 6, 2, 8)
    lat = lq.Lattice(L)
    U = orc.hot_gauge(L, 801)
    rng = np.random.default_rng(802)
    big = rng.standard_normal((4, L[3] + 2 * w, L[2] + 2 * w, L[1] + 2 * w, L[0] + 2 * w, 3, 3)) + 0j
    inner = (slice(None), slice(w, -w), slice(w, -w), slice(w, -w), slice(w, -w))
    big[inner] = U
    Ud = lq.Gaugefields(lat).upload(big, nwing=w)
    assert np.array_equal(Ud.download(), U)                                  # interior only, bit-exact
    out = np.full_like(big, 7.0)
    Ud.download(nwing=w, into=out)
    assert np.array_equal(out[inner], U)
    mask = np.ones(big.shape, dtype=bool); mask[inner] = False
    assert (out[mask] == 7.0).all()                                           # wings untouched
    for kind in (lq.WILSON, lq.STAGGERED):
        psi = orc.gaussian_spinor(lat.fermion_shape(kind), 803)
        shp = ((4,) if kind == lq.WILSON else ()) + (L[3] + 2 * w, L[2] + 2 * w, L[1] + 2 * w, L[0] + 2 * w, 3)
        bigp = np.full(shp, 5.0 + 0j)
        sl = ((slice(None),) if kind == lq.WILSON else ()) + (slice(w, -w),) * 4
        bigp[sl] = psi
        f = lq.Fermionfields(lat, kind).upload(bigp, nwing=w)
        assert np.array_equal(f.download(), psi)
        outp = np.full(shp, 9.0 + 0j)
        f.download_wing(outp, w)
        assert np.array_equal(outp[sl], psi)
        m = np.ones(shp, dtype=bool); m[sl] = False
        assert (outp[m] == 9.0).all()
