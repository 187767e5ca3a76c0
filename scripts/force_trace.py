"""calc_UdSfdU! (AbstractMD.jl:129) at 32^3x64, a few calls -- for rocprofv3 --kernel-trace (gpurun helper).  env: MIXED=0|1 (solver), CSW=c (Wilson-clover), LQCD_SET="key=value ..." """
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import latticeqcd_jl_amd as lq
U = lq.Initialize_Gaugefields(3, 0, 32, 32, 32, 64, condition="hot", randomseed=111)
lat = U.lattice
for kv in os.environ.get("LQCD_SET", "").split():
    k, v = kv.split("=")
    lat.set_param(k, int(v))
csw = float(os.environ.get("CSW", "0"))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "WilsonClover" if csw else "Wilson", "Clover_coefficient": csw, "κ": 0.141139, "eps_CG": 1e-16, "MaxCGstep": 3000})
fa = lq.FermiAction(D)
eta = lq.Fermionfields(lat, lq.WILSON); X = eta.similar()
lq.gauss_distribution_fermion_(X, 5)
lq.sample_pseudofermions_(eta, U, fa, X)
G = lq.Gaugefields(lat)
lat.set_param("mixed_action_solver", int(os.environ.get("MIXED", "1")))
lq.calc_UdSfdU_(G, fa, U, eta); lat.sync()
t0 = time.perf_counter()
for _ in range(5):
    lq.calc_UdSfdU_(G, fa, U, eta)
lat.sync()
print("calc_UdSfdU ms", 1e3 * (time.perf_counter() - t0) / 5)
