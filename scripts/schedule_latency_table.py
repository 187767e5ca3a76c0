"""The halo schedules against an exchange that completes late (tunable halo_inject_us: the stand-in for the flight time of real links on the one-GPU proxy):
CG iterations/s at the N = 8 local volume for every schedule and for the tuner's pick, at 0 / 20 / 40 us per exchange, on the peer-mapped (selfcomm 2) or the
RCCL (selfcomm 1) backend mapped onto itself.  usage: LQCD_FORCE_PARTITION=14 python scripts/schedule_latency_table.py [selfcomm] [lattice]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes
import numpy as np
import latticeqcd_jl_amd as lq

selfcomm = int(sys.argv[1]) if len(sys.argv) > 1 else 2
L = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "32,16,16,32").split(","))
lat = lq.Lattice(L)
if selfcomm == 2:
    lat.comm_init_peer()
else:
    lat.comm_init(lq.comm_unique_id())
U = lq.Gaugefields(lat)
lq.lib.check(lq.lib.lib().lqcd_gauge_hot_start(U._h, ctypes.c_uint64(111)))
D = lq.Dirac_operator(U, None, {"Dirac_operator": "Wilson", "κ": 0.141139})
b = lq.Fermionfields(lat, lq.WILSON)
lq.gauss_distribution_fermion_(b, 112)
x, y = b.similar(), b.similar()
lat.set_param("halo_stream_mode", 3)
lq.mul_(y, D, b)
ref = y.download().copy()
print("backend %s, local lattice %s, mask %s" % (lat.comm_backend, L, os.environ.get("LQCD_FORCE_PARTITION")))
print("inject_us  " + "  ".join("sched%d" % m for m in range(5)) + "   tuner(pick)     [CG iterations / s; Dslash us in brackets]")
for inject in (0, 20, 40):
    lat.set_param("halo_inject_us", inject)
    row = []
    for mode in (0, 1, 2, 3, 4, -1):
        lat.set_param("halo_stream_mode", mode)
        lq.mul_(y, D, b)
        err = float(np.abs(y.download() - ref).max() / np.abs(ref).max())
        assert err < 1e-13, (mode, inject, err)
        ms = lq.bench_dslash(D, y, b, warm=20, reps=200)
        msi = lq.bench_cg(D, x, b, warm=2, niter=400)
        row.append("%6.0f [%5.1f]" % (1e3 / msi, 1e3 * ms) + (" (%d)" % lat.get_param("halo_stream_mode") if mode < 0 else ""))
    print("%8d   " % inject + "  ".join(row))
