"""Times the momentum update, the link update and the two as one sweep (lazy_merge = 2) at 32^3x64 -- gpurun helper.
usage: pu_probe.py [dt]   (dt: the link step; the series length of exp(dt P) follows its norm)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import latticeqcd_jl_amd as lq

dt = float(sys.argv[1]) if len(sys.argv) > 1 and "=" not in sys.argv[1] else 0.005
sets = [a.split("=") for a in sys.argv[1:] if "=" in a]          # library tunables key=value
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
for k, v in sets:
    lat.set_param(k, int(v))
p = lq.initialize_TA_Gaugefields(U)
lq.gauss_distribution_(p, 7)
beta = 5.7


def tk(fn, reps=20):
    fn(); lq.calculate_Plaquette(U)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        lq.calculate_Plaquette(U)          # runs what waits; itself ~0.3 ms / reps
        best = min(best, (time.perf_counter() - t0) / reps)
    return 1e3 * best


res = {}
lat.set_param("lazy_merge", 0)
res["P_update"] = tk(lambda: lq.P_update_(U, p, 1e-9, beta))
res["U_update"] = tk(lambda: lq.U_update_(U, p, dt))
res["P_then_U_separate"] = tk(lambda: (lq.P_update_(U, p, 1e-9, beta), lq.U_update_(U, p, dt)))
lat.set_param("lazy_merge", 2)
res["P_then_U_one_sweep"] = tk(lambda: (lq.P_update_(U, p, 1e-9, beta), lq.U_update_(U, p, dt)))
print({k: round(v, 4) for k, v in res.items()}, "dt", dt, "sets", sets, "unitarity", lq.unitarity_deviation(U))
