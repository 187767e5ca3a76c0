import sys, ctypes
sys.path.insert(0, "/root/repo")
import latticeqcd_jl_amd as lq
for L in ((32, 32, 32, 64), (16, 16, 16, 32)):
    lat = lq.Lattice(L)
    U = lq.Gaugefields(lat)
    lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
    D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139, "r": 1.0, "boundarycondition": (1, 1, 1, -1)})
    b = lq.Fermionfields(lat, lq.WILSON)
    lq.gauss_distribution_fermion_(b, 112)
    y = b.similar()
    lq.mul_(y, D, b)
    print(L, "%.17e" % lq.dot(y, y).real, "%.17e" % lq.dot(b, b).real, "%.17e" % lq.calculate_Plaquette(U))
    lat.close()
