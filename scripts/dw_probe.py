"""Domainwall application: one five-dimensional launch (dw_batched = 1) vs L5 Wilson launches + a fifth-direction pass -- gpurun helper."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import latticeqcd_jl_amd as lq

sets = [a.split("=") for a in sys.argv[1:] if "=" in a]          # library tunables key=value
cases = (((32, 32, 32, 64), 8),) if sets else (((16, 16, 16, 32), 8), ((32, 32, 32, 64), 8), ((32, 32, 32, 64), 16))
for L, L5 in cases:
    U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
    lat = U.lattice
    for k, v in sets:
        lat.set_param(k, int(v))
    x = lq.Initialize_pseudofermion_fields(U[1], "Domainwall", L5=L5)
    D = lq.Dirac_operator(U, x, {"Dirac_operator": "Domainwall", "mass": 0.05, "L5": L5, "M": -1.8})
    lq.gauss_distribution_fermion_(x, 3)
    y = x.similar()
    V5 = L[0] * L[1] * L[2] * L[3] * L5
    out = {}
    for batched in (1, 0):
        lat.set_param("dw_batched", batched)
        for _ in range(5):
            lq.mul_(y, D, x)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(20):
                lq.mul_(y, D, x)
            best = min(best, (time.perf_counter() - t0) / 20)
        out[batched] = 1e3 * best
    # bytes a five-dimensional site needs at least: psi in + out 384, links 384 / L5 (12-real, shared by the slices)
    print("sets", sets, "L", L, "L5", L5, "D5 ms one launch %.4f / slice by slice %.4f; one launch: %.0f GB/s on (384 + 384/L5) B per 5-d site" %
          (out[1], out[0], V5 * (384 + 384 / L5) / out[1] / 1e6))
    lat.set_param("dw_batched", 1)
    if L[0] == 16:      # the CG on D^+D: fused iteration on the five-dimensional launch vs the generic loop (fixed number of iterations: eps unreachable)
        D2 = lq.Dirac_operator(U, x, {"Dirac_operator": "Domainwall", "mass": 0.05, "L5": L5, "M": -1.8, "eps_CG": 1e-60, "MaxCGstep": 40})
        for fused in (1, 0, 1, 0):
            lat.set_param("dw_fused_cg", fused)
            lq.clear_fermion_(y)
            t0 = time.perf_counter()
            try:
                lq.solve_DinvX_(y, lq.DdagD_operator(D2), x)
            except lq.NotConverged:
                pass
            print("   CG on D^+D, dw_fused_cg %d: %.3f ms per iteration (40 iterations, setup included)" % (fused, 1e3 * (time.perf_counter() - t0) / 40))
        D2.close()
    for o in (y, x, D, U):
        o.close()
