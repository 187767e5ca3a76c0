"""Unitarity of the links under molecular dynamics.  exp(dt P) U leaves SU(3) only by rounding, but the rounding
accumulates: max |row2 - conj(row0 x row1)| passes the 12-real gate (1e-14) within a few hundred link updates
(profiles/r03_unitarity_drift.log).  lqcd_gauge_exp_update therefore projects the updated link back onto the group in the same pass
(tunable md_reunitarize, default 1; 0 = the reference's literal U_update!, AbstractMD.jl:78-97)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
BETA = 5.7


@pytest.fixture(scope="module")
def gpu(lq):
    assert lq.lib.device_count() > 0, "no HIP device visible: the product has no CPU fallback"
    return lq


def md_leg(lq, U, p, n, dt=0.005):
    for _ in range(n):
        lq.U_update_(U, p, 0.5 * dt)
        lq.P_update_(U, p, dt, BETA)
        lq.U_update_(U, p, 0.5 * dt)


def test_projection_keeps_the_12_real_path_alive_and_the_literal_update_loses_it(gpu, orc):
    lq = gpu
    L = (8, 8, 8, 8)
    devs = {}
    for mode in (1, 0):
        lat = lq.Lattice(L)
        lat.set_param("md_reunitarize", mode)
        U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 5))
        p = lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 6)
        md_leg(lq, U, p, 200)                       # 400 link updates = one trajectory of the reference's Wilson test (20 steps, N_sw = 10)
        devs[mode] = lq.unitarity_deviation(U)
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139})
        x = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(x, 7)
        y = x.similar()
        lq.mul_(y, D, x)
        assert lat.get_param("recon_active") == (1 if devs[mode] <= 1e-14 else 0)
    assert devs[1] < 2e-15, devs          # projected every update: at the rounding of one cross product
    assert devs[0] > 3 * devs[1], devs    # literal update: the deviation random-walks upwards (past 1e-14 within ~300 updates at 16^3 x 32)


def test_one_update_equals_the_oracle_in_both_modes(gpu, orc):
    lq = gpu
    L = (4, 4, 4, 4)
    Uh = orc.hot_gauge(L, 8)
    for mode in (1, 0):
        lat = lq.Lattice(L)
        lat.set_param("md_reunitarize", mode)
        U = lq.Gaugefields(lat).upload(Uh)
        p = lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 9)
        ref = orc.link_update(Uh, p.download(), 0.05, L)
        lq.U_update_(U, p, 0.05)
        assert np.abs(U.download() - ref).max() < 1e-13, mode


def test_trajectory_is_the_same_with_and_without_projection(gpu, orc):
    lq = gpu
    L = (8, 8, 8, 8)
    out = {}
    for mode in (1, 0):
        lat = lq.Lattice(L)
        lat.set_param("md_reunitarize", mode)
        U = lq.Gaugefields(lat).upload(orc.hot_gauge(L, 10))
        p = lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 11)
        h0 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA)
        md_leg(lq, U, p, 40)
        h1 = lq.momentum_action(p) + lq.evaluate_GaugeAction(U, BETA)
        out[mode] = (lq.calculate_Plaquette(U), h1 - h0, U.download())
    assert abs(out[1][0] - out[0][0]) < 1e-13
    assert abs(out[1][1] - out[0][1]) < 1e-8 * max(1.0, abs(out[0][1]))
    assert np.abs(out[1][2] - out[0][2]).max() < 1e-12


def test_fields_that_were_never_on_the_group_are_left_alone(gpu, orc):
    """the reference's fixtures are unitary to 9e-11 (text files): the default update must treat them exactly like the literal one"""
    lq = gpu
    import os
    from conftest import GOLDEN
    L = (4, 4, 4, 4)
    Uh = lq.gauge_io.load_ildg(os.path.join(GOLDEN, "wilson_4x4x4x4.ildg"), L)
    out = {}
    for mode in (1, 0):
        lat = lq.Lattice(L)
        lat.set_param("md_reunitarize", mode)
        U = lq.Gaugefields(lat).upload(Uh)
        assert lq.unitarity_deviation(U) > 1e-12
        p = lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 21)
        md_leg(lq, U, p, 10)
        out[mode] = U.download()
    assert np.array_equal(out[0], out[1])


def test_reunitarize_entry_point(gpu, orc):
    lq = gpu
    L = (4, 4, 4, 4)
    lat = lq.Lattice(L)
    Uh = orc.hot_gauge(L, 12)
    rng = np.random.default_rng(13)
    Up = Uh + 1e-9 * (rng.standard_normal(Uh.shape) + 1j * rng.standard_normal(Uh.shape))
    U = lq.Gaugefields(lat).upload(Up)
    assert lq.unitarity_deviation(U) > 1e-10
    lq.reunitarize_(U)
    assert lq.unitarity_deviation(U) < 1e-15
    V = U.download()
    W = lq.Gaugefields(lat).upload(Uh)
    lq.reunitarize_(W)
    assert np.abs(W.download() - Uh).max() < 2e-15    # an SU(3) field moves by rounding only
    assert np.abs(V - Uh).max() < 1e-8                # and the perturbed one comes back next to where it started
    assert abs(lq.calculate_Plaquette(U) - orc.plaquette(V, L)) < 1e-13


def test_staple_sweep_with_two_row_link_loads(gpu, orc):
    """links known to be on the group (tracked per field version) are read as rows 0, 1 + rebuilt row 2 by the staple sweep (tunable
    staple_recon): same force as the full loads and as the oracle; an uploaded field (unknown provenance) takes the full loads"""
    lq = gpu
    L = (8, 4, 6, 4)
    Uh = orc.hot_gauge(L, 14)
    ref = orc.gauge_force(Uh, L, BETA)
    out = {}
    for recon in (1, 0):
        lat = lq.Lattice(L)
        lat.set_param("staple_recon", recon)
        U = lq.Gaugefields(lat).upload(Uh)
        lq.reunitarize_(U)                    # marks this version as on the group (moves an SU(3) field by rounding only)
        G = lq.Gaugefields(lat)
        lq.gauge_force_(G, U, BETA)
        out[recon] = G.download()
        assert np.abs(out[recon] - ref).max() / np.abs(ref).max() < 1e-13, recon
        p = lq.Gaugefields(lat)
        lq.gauss_distribution_(p, 15)
        p0 = p.download()
        lq.P_update_(U, p, 0.1, BETA)         # fused force + TA projection
        out[("p", recon)] = p.download() - p0
    assert np.abs(out[1] - out[0]).max() / np.abs(out[0]).max() < 1e-14
    assert np.abs(out[("p", 1)] - out[("p", 0)]).max() / np.abs(out[("p", 0)]).max() < 1e-13
