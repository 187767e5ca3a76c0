"""Staple force on partitioned lattices: in-process PE grids of every shape of SURVEY.md 8(e) against the oracle on the global
lattice, and the real RCCL path through self-partition."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

BETA = 5.7


@pytest.mark.parametrize("pe", [(1, 1, 1, 2), (1, 1, 2, 2), (1, 2, 2, 2), (2, 2, 2, 2), (1, 1, 1, 4)])
def test_partitioned_gauge_force_equals_single_domain(lq, orc, pe):
    assert lq.lib.device_count() > 0
    gL = (8, 8, 8, 16)
    n = int(np.prod(pe))
    U = orc.hot_gauge(gL, 601)
    P = orc.gaussian_momenta(gL, 602)
    lats = [lq.Lattice(gL, pe, r) for r in range(n)]
    lq.link_local(lats)
    Us = [lq.Gaugefields(lat).upload(lq.pegrid.local_view(U, lat.local_L, lat.origin, lead=1)) for lat in lats]
    Gs = [lq.Gaugefields(lat) for lat in lats]
    Ps = [lq.Gaugefields(lat).upload(lq.pegrid.local_view(P, lat.local_L, lat.origin, lead=1)) for lat in lats]
    lq.mdom_gauge_force_(Gs, Us, BETA)
    Gref = orc.gauge_force(U, gL, BETA)
    for lat, G in zip(lats, Gs):
        assert rel_err(G.download(), lq.pegrid.local_view(Gref, lat.local_L, lat.origin, lead=1)) < 1e-13
    lq.mdom_P_update_(Us, Ps, -0.21, BETA)
    Pref = orc.momentum_add_ta(P.copy(), -0.21, Gref, gL)
    for lat, Pd in zip(lats, Ps):
        assert rel_err(Pd.download(), lq.pegrid.local_view(Pref, lat.local_L, lat.origin, lead=1)) < 1e-13


def test_gauge_force_rccl_self_partition(lq, orc):
    code = textwrap.dedent("""
        import os, sys, numpy as np
        sys.path.insert(0, os.getcwd())
        import latticeqcd_jl_amd as lq
        from oracle import oracle as orc
        L, beta = (8, 4, 6, 8), 5.7
        lat = lq.Lattice(L)
        lat.comm_init(lq.comm_unique_id())
        U = orc.hot_gauge(L, 603)
        P = orc.gaussian_momenta(L, 604)
        Ud, Pd, G = lq.Gaugefields(lat).upload(U), lq.Gaugefields(lat).upload(P), lq.Gaugefields(lat)
        lq.gauge_force_(G, Ud, beta)
        Gref = orc.gauge_force(U, L, beta)
        assert np.abs(G.download() - Gref).max() / np.abs(Gref).max() < 1e-13
        lq.P_update_(Ud, Pd, 0.3, beta)
        Pref = orc.momentum_add_ta(P.copy(), 0.3, Gref, L)
        assert np.abs(Pd.download() - Pref).max() / np.abs(Pref).max() < 1e-13
        # the per-direction interface of the unchanged callers on the partitioned lattice: eager calls, one fused kernel per direction (three
        # directions only: the deferred triples run one by one), and all four (one fused four-direction call) give the same momenta
        ga = lq.GaugeAction(Ud)
        pl = lq.make_loops_fromname("plaquette", Dim=4)
        ga.push_(beta / 2, pl + lq.make_loops_fromname("plaquette", Dim=4, adjoint=True))
        tmp = lq.Gaugefields(lat)
        out = {}
        for mode, dirs in (("eager", (1, 2, 3, 4)), ("lazy4", (1, 2, 3, 4)), ("lazy3", (1, 2, 3))):
            lat.lazy_links = mode != "eager"
            Pm = lq.Gaugefields(lat).upload(P)
            for mu in dirs:
                lq.calc_dSdUmu_(tmp[1], ga, mu, Ud)
                lq.mul_(tmp[2], Ud[mu], tmp[1])
                lq.Traceless_antihermitian_add_(Pm[mu], -0.1, tmp[2])
            if mode == "lazy3":
                lat.lazy_links = False
                lq.calc_dSdUmu_(tmp[1], ga, 4, Ud)
                lq.mul_(tmp[2], Ud[4], tmp[1])
                lq.Traceless_antihermitian_add_(Pm[4], -0.1, tmp[2])
            out[mode] = Pm.download()
        lat.lazy_links = True
        Pref2 = orc.momentum_add_ta(P.copy(), 0.3, Gref, L)
        for mode in ("eager", "lazy4", "lazy3"):
            assert np.abs(out[mode] - Pref2).max() / np.abs(Pref2).max() < 1e-13, mode
        print("GF_SELF_OK")
    """)
    for mask in ("8", "14", "15"):
        env = dict(os.environ, LQCD_FORCE_PARTITION=mask, HSA_ENABLE_IPC_MODE_LEGACY="0", LQCD_HALO_STREAM_MODE=("-1" if mask == "8" else "3"))      # t only: the tuner's choice; the others: the folded one-stream schedule
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and "GF_SELF_OK" in r.stdout, (mask, r.stdout[-2000:], r.stderr[-3000:])
