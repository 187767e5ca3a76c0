"""Worker for test_host_logic.test_gloo_world2_domain_decomposed_dslash (run under torch.distributed.run, gloo).

Validates the multi-rank HOST logic that bench.py uses at N > 1 -- PE-grid decomposition, neighbour ranks, which rank
owns the global boundary sign, slicing global fields -- by doing a t-partitioned Wilson and staggered Dslash with the
oracle as the per-rank compute engine and gloo send/recv as the halo transport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import latticeqcd_jl_amd as lq  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def exchange(send_to_fwd, send_to_bwd, fwd, bwd):
    """returns (from_bwd, from_fwd)"""
    a = torch.from_numpy(np.ascontiguousarray(send_to_fwd).view(np.float64).copy())
    b = torch.from_numpy(np.ascontiguousarray(send_to_bwd).view(np.float64).copy())
    ra, rb = torch.empty_like(a), torch.empty_like(b)
    reqs = [dist.isend(a, fwd), dist.isend(b, bwd), dist.irecv(ra, bwd), dist.irecv(rb, fwd)]
    # with 2 ranks fwd == bwd: messages are matched in posting order (first send <-> first recv)
    for r in reqs:
        r.wait()
    return ra.numpy().view(np.complex128).reshape(send_to_fwd.shape), rb.numpy().view(np.complex128).reshape(send_to_bwd.shape)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    gL = (4, 4, 4, 8)
    pe = lq.pegrid.choose_pe_grid(gL, world)
    assert pe == (1, 1, 1, 2)
    local, origin, nf, nb = lq.pegrid.decompose(gL, pe, rank)
    bc = (1, 1, 1, -1)
    U = orc.hot_gauge(gL, 111)
    for kind in (orc.WILSON, orc.STAGGERED):
        shape = orc.wilson_shape(gL) if kind == orc.WILSON else orc.staggered_shape(gL)
        lead = 1 if kind == orc.WILSON else 0
        psi = orc.gaussian_spinor(shape, 112)
        km = 0.141139 if kind == orc.WILSON else 0.5
        ref = orc.apply_D(kind, U, psi, gL, km, 1.0, bc)
        # local pieces
        Ul = lq.pegrid.local_view(U, local, origin, lead=1)
        pl = lq.pegrid.local_view(psi, local, origin, lead=lead)
        tl = local[3]
        tax = lead  # t axis of the spinor array
        first = np.take(pl, [0], axis=tax)
        last = np.take(pl, [tl - 1], axis=tax)
        from_bwd, from_fwd = exchange(last, first, nf[3], nb[3])       # my last slice goes forward, first goes backward
        Ulast = Ul[:, tl - 1:tl]
        Ufrom_bwd, _ = exchange(Ulast, Ul[:, 0:1], nf[3], nb[3])       # backward links of the first local slice
        # global boundary signs are applied by the rank that owns the wrap
        sgn_up = bc[3] if origin[3] + tl == gL[3] else 1
        sgn_dn = bc[3] if origin[3] == 0 else 1
        pad_psi = np.concatenate([sgn_dn * from_bwd, pl, sgn_up * from_fwd], axis=tax)
        pad_U = np.concatenate([Ufrom_bwd, Ul, Ul[:, 0:1]], axis=1)     # top slice links are never used by interior outputs
        padL = (local[0], local[1], local[2], tl + 2)
        out = orc.apply_D(kind, np.ascontiguousarray(pad_U), np.ascontiguousarray(pad_psi), padL, km, 1.0, (1, 1, 1, 1))
        mine = np.take(out, range(1, tl + 1), axis=tax)
        # staggered phases depend only on x,y,z for the t direction and on global parity of lower coords: unaffected by the t pad
        want = lq.pegrid.local_view(ref, local, origin, lead=lead)
        err = np.abs(mine - want).max() / np.abs(want).max()
        assert err < 1e-14, (kind, rank, err)
        # global reductions: dot over ranks == dot of the global field
        loc = np.array([np.vdot(pl, pl).real])
        t = torch.from_numpy(loc)
        dist.all_reduce(t)
        assert abs(t.item() - np.vdot(psi, psi).real) < 1e-9 * t.item()
    print(f"DD_OK rank {rank}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
