"""Pins the oracle and the product's readers to the reference's own data: gauge-configuration formats, site/link index
order and plaquettes of the fixtures under /root/reference/test/confs_* (copied as data by tests/golden/make_golden.py;
SURVEY.md Appendix B).  This is the only level at which the reference pins anything for this path."""
import json
import os

import numpy as np

from conftest import GOLDEN

# SURVEY.md Appendix B (computed there with numpy on the reference fixtures)
APPENDIX_B = {
    "confs_HMC_L04040404_beta5.7_Wilson_kappa0.141139": 0.5658002268452863,
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5": 0.5755840394747146,
    "confs_HMC_L04040404_beta5.7_Domainwall": 0.5832960791270716,
}
FILES = {
    "confs_HMC_L04040404_beta5.7_Wilson_kappa0.141139": ("wilson_4x4x4x4.ildg", (4, 4, 4, 4)),
    "confs_HMC_L04040404_beta5.7_Staggered_mass0.5": ("staggered_4x4x4x4.ildg", (4, 4, 4, 4)),
    "confs_HMC_L04040404_beta5.7_Domainwall": ("domainwall_4x4x2x2.ildg", (4, 4, 2, 2)),
}


def gold():
    with open(os.path.join(GOLDEN, "golden.json")) as f:
        return json.load(f)


def test_golden_matches_survey_appendix_b():
    g = gold()
    for k, v in APPENDIX_B.items():
        assert abs(g["plaquette"][k] - v) < 1e-13


def test_ildg_fixtures_decode_to_golden_plaquette(orc, lq):
    g = gold()
    for key, (fname, L) in FILES.items():
        U = lq.gauge_io.load_ildg(os.path.join(GOLDEN, fname), L)
        assert U.shape == orc.gauge_shape(L)
        assert abs(orc.plaquette(U, L) - g["plaquette"][key]) < 1e-14
        assert orc.unitarity_dev(U, L) < 1e-9          # text-derived files carry ~11 digits
        # the wrong index order is detectably wrong (transpose of the 3x3 block)
        Ut = np.ascontiguousarray(np.swapaxes(U, -1, -2))
        assert abs(orc.plaquette(Ut, L) - g["plaquette"][key]) > 0.1


def test_bridgetext_equals_ildg(lq):
    L = (4, 4, 2, 2)
    Ub = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "domainwall_4x4x2x2.ildg"), L)
    Ut = lq.gauge_io.load_BridgeText(os.path.join(GOLDEN, "domainwall_4x4x2x2.ildg.txt"), L)
    assert np.abs(Ub - Ut).max() == 0.0


def test_lime_header_bytes():
    with open(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), "rb") as f:
        hdr = f.read(144)
    assert hdr[:4] == bytes.fromhex("456789ab") and hdr[4:6] == b"\x00\x01" and hdr[6:8] == b"\xc0\x00"
    assert int.from_bytes(hdr[8:16], "big") == 256 * 4 * 9 * 16 == 147456
    assert hdr[16:32] == b"ildg-binary-data"


def test_io_round_trip(tmp_path, orc, lq):
    L = (4, 2, 6, 2)
    U = orc.hot_gauge(L, 3)
    p1, p2 = str(tmp_path / "c.ildg"), str(tmp_path / "c.txt")
    lq.gauge_io.save_ildg(p1, U)
    lq.gauge_io.save_BridgeText(p2, U)
    assert np.array_equal(lq.gauge_io.load_ildg(p1, L), U)
    assert np.abs(lq.gauge_io.load_BridgeText(p2, L) - U).max() < 1e-14
    # a written file has exactly the reference's single-record framing
    with open(p1, "rb") as f:
        buf = f.read()
    recs = list(lq.gauge_io.read_lime_records(buf))
    assert [r[0] for r in recs] == ["ildg-binary-data"] and len(buf) == 144 + len(recs[0][1])


def test_bad_sizes_raise(tmp_path, lq):
    import pytest
    with pytest.raises(ValueError):
        lq.gauge_io.load_ildg(os.path.join(GOLDEN, "domainwall_4x4x2x2.ildg"), (4, 4, 4, 4))
    with pytest.raises(ValueError):
        lq.gauge_io.load_BridgeText(os.path.join(GOLDEN, "domainwall_4x4x2x2.ildg.txt"), (4, 4, 4, 4))
    p = tmp_path / "junk.ildg"
    p.write_bytes(b"\0" * 200)
    with pytest.raises(ValueError):
        lq.gauge_io.load_ildg(str(p), (4, 4, 2, 2))
