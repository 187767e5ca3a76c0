#!/usr/bin/env python3
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
def free():
    f, t = C.c_int64(0), C.c_int64(0)
    lq.lib.check(lq.lib.lib().lqcd_device_mem_info(0, C.byref(f), C.byref(t)))
    return f.value
L = (16, 16, 16, 16)
def run(label, fn):
    deltas = []
    for _ in range(4):
        U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=1)
        lat = U.lattice
        D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.12, "eps_CG": 1e-12})
        b = lq.Fermionfields(lat, lq.WILSON); lq.gauss_distribution_fermion_(b, 2)
        x = b.similar()
        f0 = free()
        fn(lat, U, D, b, x)
        for o in (x, b, D, U): o.close()
        lat.close()
        deltas.append(free())
    print("%-22s free after each cycle (MB rel. to first): %s" % (label, [round((d - deltas[0]) / 2**20, 1) for d in deltas]))
run("nothing", lambda lat, U, D, b, x: None)
run("cg", lambda lat, U, D, b, x: lq.solve_DinvX_(x, lq.DdagD_operator(D), b))
run("mixed", lambda lat, U, D, b, x: lq.solve_mixed_DinvX_(x, lq.DdagD_operator(D), b))
def ms(lat, U, D, b, x):
    xs = [b.similar() for _ in range(3)]
    lq.shiftedcg(xs, [0.1, 0.5, 2.0], x, lq.DdagD_operator(D), b)
    for o in xs: o.close()
run("multishift", ms)
def eo(lat, U, D, b, x):
    D.method_CG = "bicgstab_evenodd"; lq.solve_DinvX_(x, D, b)
run("bicgstab_eo", eo)
def force(lat, U, D, b, x):
    G = lq.Gaugefields(lat); fa = lq.FermiAction(D)
    lq.calc_UdSfdU_(G, fa, U, b); lq.P_update_(U, G, 0.01, 5.7); G.close(); fa.close()
run("force", force)
def recon(lat, U, D, b, x):
    lat.set_param("gauge_recon", 12); lq.mul_(x, D, b)
run("recon12", recon)
