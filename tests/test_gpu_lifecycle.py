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


def test_destroy_from_another_thread_is_parked_for_the_context_thread(lq):
    """a finalizer thread's lqcd_gauge_destroy must neither run nor read the recorded link operations (the context's thread may be inside a call).
    The field is parked; the context's thread flushes what names it and frees it at its next sync / gauge creation."""
    import threading
    L = (16, 16, 16, 16)
    plaq = {}
    for how in ("same_thread", "other_thread", "adopted"):
        U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=3)
        lat = U.lattice
        assert lat.get_param("lazy_links") == 1 and lat.get_param("parked_fields") == 0      # the bindings switch the recorded link operations on
        p = lq.initialize_TA_Gaugefields(U)
        lq.gauss_distribution_(p, 5)
        lq.U_update_(U, p, 0.01)
        lat.sync()                                           # (work space of the update exists from here on)
        lq.U_update_(U, p, 0.01)
        assert lat.get_param("lazy_deferred") == 4          # the link update waits for one to merge with: it names U and p
        before = _free(lq)
        if how == "same_thread":
            p.close()                                        # runs what names p, then frees it
            assert lat.get_param("parked_fields") == 0 and lat.get_param("lazy_deferred") == 0
        else:
            if how == "adopted":                             # a worker that owns the context from now on destroys at once
                t = threading.Thread(target=lambda: (lat.set_param("adopt_thread", 1), p.close()))
            else:
                t = threading.Thread(target=p.close)
            t.start()
            t.join()
            if how == "adopted":
                assert lat.get_param("parked_fields") == 0 and lat.get_param("lazy_deferred") == 0
                lat.set_param("adopt_thread", 1)             # back to this thread
            else:
                assert lat.get_param("parked_fields") == 1 and lat.get_param("lazy_deferred") == 4      # nothing ran, nothing was read
                assert _free(lq) - before < 8 << 20                                                        # and nothing was freed
                lat.sync()                                                                                 # the context's thread: flush, free
                assert lat.get_param("parked_fields") == 0 and lat.get_param("lazy_deferred") == 0
        assert _free(lq) - before > 30 << 20                 # the 37.7 MB of the momenta are back
        plaq[how] = lq.calculate_Plaquette(U)
        U.close()
        lat.close()
    assert plaq["same_thread"] == plaq["other_thread"] == plaq["adopted"]

    # a context destroyed with fields still parked takes them along
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="cold")
    lat = U.lattice
    G = lq.Gaugefields(lat)
    base = _free(lq)
    t = threading.Thread(target=G.close)
    t.start()
    t.join()
    assert lat.get_param("parked_fields") == 1
    U.close()
    lat.close()
    assert _free(lq) - base > 60 << 20
