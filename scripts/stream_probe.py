#!/usr/bin/env python3
"""Practical HBM ceiling of this box through the library's own BLAS-1 kernels on a 2 GB field (48^3x96 Wilson spinor):
read-only (norm2), 2 reads (dot), 2 reads + 1 write (axpy)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import latticeqcd_jl_amd as lq
import ctypes as C
L = (48, 48, 48, 96)
lat = lq.Lattice(L)
a = lq.Fermionfields(lat, lq.WILSON); b = a.similar()
lq.gauss_distribution_fermion_(a, 1); lq.gauss_distribution_fermion_(b, 2)
nbytes = 48 ** 3 * 96 * 192
def t(fn, passes, reps=20):
    fn(); lat.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    lat.sync()
    dt = (time.perf_counter() - t0) / reps
    print("%-28s %.3f ms  %.0f GB/s" % (fn.__name__, 1e3 * dt, passes * nbytes / dt / 1e9))
def norm2(): lq.dot(a, a)
def dot2(): lq.dot(a, b)
def axpy(): lq.add_fermion_(b, 1e-9, a)
t(norm2, 1); t(dot2, 2); t(axpy, 3)
