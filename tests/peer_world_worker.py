"""Worker of tests/test_gpu_peer_world2.py: one of N real PROCESSES that share cuda:0 and communicate through the peer-mapped backend
(csrc/comm.hip: hipIpc-mapped windows, flag kernels, slot reductions).  Run under torch.distributed.run; gloo carries only the 256-byte
window descriptions.  Every rank checks its sub-lattice of D, D^+, the CG solution (+ the rarer face exchanges: plaquette, staple force,
fermion force) against the CPU oracle's result on the GLOBAL lattice."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import latticeqcd_jl_amd as lq  # noqa: E402
from oracle import oracle as orc  # noqa: E402

KAPPA, MASS, CSW, BC = 0.125, 0.5, 1.5612, (1, 1, 1, -1)


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def gather_blobs(blob):
    mine = torch.tensor(list(blob), dtype=torch.uint8)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [bytes(t.tolist()) for t in out]


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    orc.set_threads(2)
    gL = tuple(int(v) for v in os.environ["PEER_TEST_LATTICE"].split(","))
    pe = tuple(int(v) for v in os.environ["PEER_TEST_PE"].split(","))
    kinds = os.environ.get("PEER_TEST_KINDS", "Wilson,Staggered,WilsonClover").split(",")
    sched = os.environ.get("PEER_TEST_SCHEDULES", "3,4,0,1,2,-1").split(",")
    assert int(np.prod(pe)) == world
    lat = lq.Lattice(gL, pe, rank, device=0)            # every rank on the ONE device
    lat.set_param("peer_timeout_ms", 20000)
    lat.comm_init_peer(gather_blobs)
    assert lat.comm_backend == "peer"
    U = orc.hot_gauge(gL, 111)
    Ud = lq.Gaugefields(lat).upload(lq.pegrid.local_view(U, lat.local_L, lat.origin, lead=1))
    if os.environ.get("PEER_TEST_DEAD_RANK"):      # a rank that dies: its neighbours' waits time out, the call reports LQCD_ERR_COMM -- and so does every later one, at once
        import time
        lat.set_param("peer_timeout_ms", 1500)
        dist.barrier()
        if rank == int(os.environ["PEER_TEST_DEAD_RANK"]):
            os._exit(0)
        D = lq.Dirac_operator(Ud, None, {"Dirac_operator": "Wilson", "κ": KAPPA, "boundarycondition": BC})
        x = lq.Fermionfields(lat, lq.WILSON)
        lq.gauss_distribution_fermion_(x, 5)
        y = x.similar()
        t0 = time.perf_counter()
        for attempt in range(2):
            try:
                lq.mul_(y, D, x)
                raise AssertionError("an exchange with a dead rank returned a result")
            except lq.LQCDError as e:
                assert e.code == lq.lib.ERR_COMM and "gave up waiting" in str(e), str(e)
        dt = time.perf_counter() - t0
        assert 1.0 < dt < 10.0, dt      # one timeout, then fail-fast
        print(f"PEER_DEAD_OK rank {rank} after {dt:.1f} s", flush=True)
        os._exit(0)
    # the mailbox path: link faces for the plaquette
    assert abs(lq.calculate_Plaquette(Ud) - orc.plaquette(U, gL)) < 1e-13
    for name in kinds:
        kind = lq.STAGGERED if name == "Staggered" else lq.WILSON
        lead = 1 if kind == lq.WILSON else 0
        shape = orc.wilson_shape(gL) if kind == lq.WILSON else orc.staggered_shape(gL)
        psi = orc.gaussian_spinor(shape, 112)
        loc = lambda a: lq.pegrid.local_view(a, lat.local_L, lat.origin, lead=lead)
        D = lq.Dirac_operator(Ud, None, {"Dirac_operator": name, "κ": KAPPA, "mass": MASS, "Clover_coefficient": CSW, "boundarycondition": BC, "eps_CG": 1e-19})
        x = lq.Fermionfields(lat, kind).upload(loc(psi))
        y, sol = x.similar(), x.similar()
        if name == "WilsonClover":
            A = orc.clover_build(U, gL, KAPPA, CSW)
            refD = lambda dag: orc.wilson_clover_D(U, A, psi, gL, KAPPA, 1.0, BC, dag)
            xo, ito, _, st = orc.cg_clover(U, A, psi, gL, KAPPA, 1.0, BC, eps=1e-19)
        else:
            km = KAPPA if kind == lq.WILSON else MASS
            refD = lambda dag: orc.apply_D(kind, U, psi, gL, km, 1.0, BC, dag)
            xo, ito, _, st = orc.cg_DdagD(kind, U, psi, gL, km, 1.0, BC, eps=1e-19)
        assert st == 0
        refs = {dag: loc(refD(dag)) for dag in (False, True)}
        dist.barrier()                                  # the oracle's work is done on every rank: the waits below are short
        for mode in sched:
            lat.set_param("halo_stream_mode", int(mode))
            for dag in (False, True):
                lq.mul_(y, D.adjoint() if dag else D, x)
                e = rel(y.download(), refs[dag])
                assert e < 1e-13, (name, mode, dag, e)
            lq.clear_fermion_(sol)
            it, rr = lq.solve_DinvX_(sol, lq.DdagD_operator(D), x, return_info=True)
            e = rel(sol.download(), loc(xo))
            assert abs(it - ito) <= 1 and rr < 1e-19 and e < 1e-9, (name, mode, it, ito, rr, e)
        # global inner product through the slot reduction
        d = lq.dot(x, x)
        assert abs(d - np.vdot(psi, psi)) < 1e-12 * abs(d)
        if name != "WilsonClover":
            lat.set_param("halo_stream_mode", 3)
            Xg = orc.gaussian_spinor(shape, 113)                  # X, Y faces of the fermion force through the mailboxes
            X = lq.Fermionfields(lat, kind).upload(loc(Xg))
            Y = X.similar()
            lq.mul_(Y, D, X)
            G = lq.Gaugefields(lat)
            lq.fermion_force_(G, D, X, Y)
            km = KAPPA if kind == lq.WILSON else MASS
            Go = orc.fermion_force(kind, U, Xg, orc.apply_D(kind, U, Xg, gL, km, 1.0, BC), gL, km, 1.0, BC)
            dist.barrier()
            e = rel(G.download(), lq.pegrid.local_view(Go, lat.local_L, lat.origin, lead=1))
            assert e < 1e-12, ("force", name, e)
    # staple force: ghost links + lower-staple faces through the mailboxes
    Gs = lq.Gaugefields(lat)
    lq.gauge_force_(Gs, Ud, 5.7)
    Fo = orc.gauge_force(U, gL, 5.7)
    e = rel(Gs.download(), lq.pegrid.local_view(Fo, lat.local_L, lat.origin, lead=1))
    assert e < 1e-12, ("staple force", e)
    lat.sync()
    print(f"PEER_WORLD_OK rank {rank} pe {pe}", flush=True)
    dist.barrier()
    lat.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
