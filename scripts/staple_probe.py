"""Time of one Sexton-Weingarten block U_update! P_update! U_update! (standardMD.jl:150-152) at 32^3x64: the one-sweep momentum + link update
(md.hip staple_force_expu) + the merged half steps.  usage: staple_probe.py [blocks] [--set key=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import latticeqcd_jl_amd as lq

args = sys.argv[1:]
sets = []
while "--set" in args:
    i = args.index("--set")
    sets.append(args[i + 1].split("="))
    del args[i:i + 2]
n = int(args[0]) if args else 100
L = (32, 32, 32, 64)
U = lq.Initialize_Gaugefields(3, 0, *L, condition="hot", randomseed=111)
lat = U.lattice
for k, v in sets:
    lat.set_param(k, int(v))
p = lq.initialize_TA_Gaugefields(U)
lq.gauss_distribution_(p, 7)


def block():
    lq.U_update_(U, p, 0.5e-9)
    lq.P_update_(U, p, 1e-9, 5.7)
    lq.U_update_(U, p, 0.5e-9)


for _ in range(10):
    block()
lq.calculate_Plaquette(U)
t0 = time.perf_counter()
for _ in range(n):
    block()
pl = lq.calculate_Plaquette(U)
print("SW block ms %.4f  %s (plaquette %.12f, unitarity %.2e)" % (1e3 * (time.perf_counter() - t0) / n, sets, pl, lq.unitarity_deviation(U)))
