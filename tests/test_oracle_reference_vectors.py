"""CPU: the oracle against operator-level vectors produced by the reference itself (tests/ref_vectors.py), when they exist.
This is the test that pins oracle/lqcd_oracle.c at the Dslash / CG level ("parity unpinned" until it runs)."""
import os

import numpy as np
import pytest

import ref_vectors as rv
from conftest import rel_err


def test_closed_form_source_is_deterministic_and_matches_its_definition():
    psi = rv.closed_form_source("wilson")
    assert psi.shape == (4, 4, 4, 4, 4, 3)
    s, t, z, y, x, c = 3, 2, 1, 3, 2, 1
    want = np.sin(0.37 + 0.11 * c + 0.23 * x + 0.31 * y + 0.43 * z + 0.59 * t + 0.71 * s) + 1j * np.cos(0.19 + 0.13 * c + 0.29 * x + 0.37 * y + 0.41 * z + 0.53 * t + 0.61 * s)
    assert psi[s, t, z, y, x, c] == want
    st = rv.closed_form_source("staggered")
    assert st.shape == (4, 4, 4, 4, 3) and np.array_equal(st, psi[0])


def test_dump_script_and_consumers_agree_on_names_and_parameters():
    """the Julia script cannot run here; what can be checked is that it writes exactly the files the consumers read, uses the fixtures that
    exist, the closed-form source of ref_vectors.py and the parameters Univ passes"""
    src = open(os.path.join(os.path.dirname(rv.GOLDEN), "..", "scripts", "ref_parity_dump.jl"), encoding="utf-8").read()
    for kind in ("wilson", "staggered"):
        assert rv.FIXTURE[kind] in src and os.path.exists(os.path.join(rv.GOLDEN, rv.FIXTURE[kind]))
    for w in ("D", "Ddag", "cg_x"):
        assert 'ref_$(name)_%s.bin' % w in src
    assert "ref_parity_meta.json" in src
    for token in ("0.37 + 0.11c + 0.23x + 0.31y + 0.43z + 0.59t + 0.71s", "0.19 + 0.13c + 0.29x + 0.37y + 0.41z + 0.53t + 0.61s",
                  "0.141139", '"mass"] = 0.5', "[1, 1, 1, -1]", '"eps_CG" => 1e-19', '"MaxCGstep" => 3000'):
        assert token in src, token


def test_consumers_read_the_layout_the_dump_writes(tmp_path):
    """Julia writes vec(a) of a[c,x,y,z,t,s] (column-major: c fastest, s slowest) as interleaved re/im Float64: that is the C order of
    the C ABI's reference-layout array (s,t,z,y,x,c).  Emulate the writer with numpy and read back through ref_vectors.load."""
    for kind in ("wilson", "staggered"):
        a = rv.closed_form_source(kind)
        flat = np.ascontiguousarray(a if kind == "wilson" else a[None]).reshape(-1)          # element (c,x,y,z,t,s) at c + 3(x + 4(y + 4(z + 4(t + 4 s))))
        ns = 4 if kind == "wilson" else 1
        idx = lambda c, x, y, z, t, s: c + 3 * (x + 4 * (y + 4 * (z + 4 * (t + 4 * s))))
        full = a if kind == "wilson" else a[None]
        assert flat[idx(2, 1, 3, 0, 2, ns - 1)] == full[ns - 1, 2, 0, 3, 1, 2]
        for which in ("D", "Ddag", "cg_x"):
            flat.view(np.float64).astype("<f8").tofile(str(tmp_path / ("ref_%s_%s.bin" % (kind, which))))
        assert np.array_equal(rv.load(kind, "D", str(tmp_path)), a)


@pytest.mark.parametrize("kind", ["wilson", "staggered"])
def test_oracle_matches_reference_vectors(orc, lq, kind):
    if not rv.available():
        pytest.skip(rv.ABSENT)
    U = lq.gauge_io.load_ildg(os.path.join(rv.GOLDEN, rv.FIXTURE[kind]), rv.L)
    psi = rv.closed_form_source(kind)
    k = orc.WILSON if kind == "wilson" else orc.STAGGERED
    km = rv.KAPPA if kind == "wilson" else rv.MASS
    for dag, which in ((False, "D"), (True, "Ddag")):
        assert rel_err(orc.apply_D(k, U, psi, rv.L, km, 1.0, rv.BC, dag), rv.load(kind, which)) < 1e-12, (kind, which)
    xo, it, rr, st = orc.cg_DdagD(k, U, psi, rv.L, km, 1.0, rv.BC, eps=1e-19)
    assert st == 0 and rel_err(xo, rv.load(kind, "cg_x")) < 1e-8
    assert rv.meta()["%s_cg_true_residual" % kind] < 1e-17
