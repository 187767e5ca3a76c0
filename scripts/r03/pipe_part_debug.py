#!/usr/bin/env python3
"""debug: CG iteration counts of the persistent kernel on a self-partitioned lattice under solver / halo settings"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import latticeqcd_jl_amd as lq
from oracle import oracle as orc
L, K, BC = (16, 8, 8, 4), 0.141139, (1, 1, 1, -1)
lat = lq.Lattice(L)
lat.comm_init(lq.comm_unique_id())
lat.set_param("pipe_grid", 8); lat.set_param("pipe_min_chunks", 1)
U = orc.hot_gauge(L, 111)
Ud = lq.Gaugefields(lat).upload(U)
D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": K, "boundarycondition": BC, "eps_CG": 1e-19})
psi = orc.gaussian_spinor(lat.fermion_shape(lq.WILSON), 112)
x = lq.Fermionfields(lat, lq.WILSON).upload(psi)
xo, ito, rro, st = orc.cg_DdagD(orc.WILSON, U, psi, L, K, 1.0, BC, eps=1e-19)
print("oracle", ito)
for mode in (0, 1, 2):
    for fold in (1, 0):
        for defer in (1, 0):
            for fused in (2, 1):
                for pipe in (0, 1):
                    lat.set_param("halo_stream_mode", mode); lat.set_param("cg_fold_scalars", fold); lat.set_param("cg_defer_x", defer)
                    lat.set_param("cg_fused", fused); lat.set_param("dslash_pipe", pipe)
                    sol = x.similar()
                    try:
                        it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
                        err = np.abs(sol.download() - xo).max() / np.abs(xo).max()
                    except Exception as e:
                        it, rr, err = -1, 0, str(e)[:60]
                    print("mode %d fold %d defer %d fused %d pipe %d: it %d rr %.2e err %s" % (mode, fold, defer, fused, pipe, it, rr, err), flush=True)
