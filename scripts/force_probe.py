"""calc_UdSfdU! at 32^3x64 (Wilson, kappa 0.141139, eps 1e-16) a few times -- for rocprofv3 --kernel-trace --stats (gpurun helper).
usage: force_probe.py [mixed 0|1] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import latticeqcd_jl_amd as lq

mixed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "eps_CG": 1e-16})
fa = lq.FermiAction(D)
eta = lq.Fermionfields(lat, lq.WILSON)
X = eta.similar()
lq.gauss_distribution_fermion_(X, 5)
lq.sample_pseudofermions_(eta, U, fa, X)
G = lq.Gaugefields(lat)
p = lq.initialize_TA_Gaugefields(U)
lq.gauss_distribution_(p, 7)
lat.set_param("mixed_action_solver", mixed)
lq.calc_UdSfdU_(G, fa, U, eta)
t0 = time.perf_counter()
for _ in range(reps):
    lq.U_update_(U, p, 1e-9)          # the links change between force evaluations, as in MD: every cache keyed on them is rebuilt
    lq.calc_UdSfdU_(G, fa, U, eta)
print("mixed", mixed, "ms per (link update + force)", 1e3 * (time.perf_counter() - t0) / reps)
