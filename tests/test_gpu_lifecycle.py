"""Object lifecycle on the device: repeated create / use / destroy of contexts, fields, operators and of every lazily
allocated work space (scratch pool, fp32 buffers, 12-real link copy, force halos) returns all HBM."""
import ctypes as C
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu


def _free(lq):
    f, t = C.c_int64(0), C.c_int64(0)
    lq.lib.check(lq.lib.lib().lqcd_device_mem_info(0, C.byref(f), C.byref(t)))
    return f.value


def _cycle(lq, L):
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=1)
    lat = U.lattice
    lat.set_param("gauge_recon", 12)
    for name, kind in (("Wilson", lq.WILSON), ("Staggered", lq.STAGGERED)):
        D = lq.Dirac_operator(U, None, {"Dirac_operator": name, "κ": 0.12, "mass": 0.5, "eps_CG": 1e-12})
        A = lq.DdagD_operator(D)
        b = lq.Fermionfields(lat, kind)
        lq.gauss_distribution_fermion_(b, 2)
        x = b.similar()
        lq.solve_DinvX_(x, A, b)
        lq.clear_fermion_(x)
        lq.solve_mixed_DinvX_(x, A, b)
        xs = [b.similar() for _ in range(3)]
        lq.shiftedcg(xs, [0.1, 0.5, 2.0], x, A, b)
        if kind == lq.WILSON:
            D.method_CG = "bicgstab_evenodd"
            lq.clear_fermion_(x)
            lq.solve_DinvX_(x, D, b)
        G = lq.Gaugefields(lat)
        lq.calc_UdSfdU_(G, lq.FermiAction(D), U, b)
        lq.P_update_(U, G, 0.01, 5.7)
        for o in (G, x, b, D, *xs):
            o.close()
    U.close()
    lat.close()


def test_no_device_memory_leak(lq):
    assert lq.lib.device_count() > 0
    L = (16, 16, 16, 16)
    for _ in range(3):                  # the HIP runtime makes one-time allocations (~80 MB) during the first two cycles
        _cycle(lq, L)
    base = _free(lq)
    for _ in range(6):
        _cycle(lq, L)
    after = _free(lq)
    assert abs(after - base) < 8 << 20, (base, after)      # nothing like the ~150 MB one cycle allocates may remain


def test_destroy_order_and_double_close_are_harmless(lq):
    L = (4, 4, 4, 4)
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="cold")
    lat = U.lattice
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson"})
    x = lq.Fermionfields(lat, lq.WILSON)
    D.close(); D.close()
    x.close(); x.close()
    U.close(); U.close()
    lat.close(); lat.close()
