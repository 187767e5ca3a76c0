"""PE-grid helpers (host logic, pure Python/numpy).

The reference's only parallelism concept is a 4-D process grid PEs = (px,py,pz,pt) (src/mpirun.jl:17-19,
src/mpi/mpimodule.jl:9-13).  These helpers choose a grid for N GPUs (keep x, the contiguous axis,
unpartitioned; at least three distinct xGMI peers at 8 GPUs; whole-chunk faces first -- see choose_pe_grid) and slice global host arrays.
The same decomposition rule is implemented in C (lqcd_decompose); tests check they agree.
"""
import numpy as np


def rank_coords(pe, rank):
    c = []
    q = rank
    for mu in range(4):
        c.append(q % pe[mu])
        q //= pe[mu]
    return tuple(c)


def coords_rank(pe, c):
    return c[0] + pe[0] * (c[1] + pe[1] * (c[2] + pe[2] * c[3]))


def decompose(global_L, pe, rank):
    """-> (local_L, origin, rank_fwd, rank_bwd); mirrors lqcd_decompose."""
    for mu in range(4):
        if global_L[mu] % pe[mu] or (global_L[mu] // pe[mu]) % 2:
            raise ValueError("global extent must be divisible by the PE grid with even local extents")
    local = tuple(global_L[mu] // pe[mu] for mu in range(4))
    c = rank_coords(pe, rank)
    origin = tuple(c[mu] * local[mu] for mu in range(4))
    fwd, bwd = [], []
    for mu in range(4):
        cf, cb = list(c), list(c)
        cf[mu] = (c[mu] + 1) % pe[mu]
        cb[mu] = (c[mu] - 1) % pe[mu]
        fwd.append(coords_rank(pe, cf))
        bwd.append(coords_rank(pe, cb))
    return local, origin, tuple(fwd), tuple(bwd)


def choose_pe_grid(global_L, ngpu, min_local=8):
    """Partition t and z in turn while their local extents stay >= min_local, then y; never x (the contiguous axis).  8 GPUs on 32^3 x 64: (1,1,2,4) -- three
    distinct xGMI peers (one in z, two in t; at most 3.1 MB per peer and application, as for (1,2,2,2)), 65 536 halo sites instead of 81 920, and -- what decided it,
    round 6 -- z and t faces are whole 64-site chunks of the stencil kernels while a y face cuts through every second chunk: with y unpartitioned 23 % of the chunks
    touch a face instead of 59 %, so the bulk launch of the overlapping schedules has something to hide the exchange behind.  Measured on the one-GPU proxy
    (profiles/r06_schedule_latency_table.log): CG iterations / s at the 8-GPU local volume, (1,1,2,4) vs (1,2,2,2): 6674 vs 6454 with no flight time, 5449 vs 4620
    with 40 us per exchange."""
    pe = [1, 1, 1, 1]
    n = ngpu

    def can_split(mu, floor):
        return global_L[mu] % (pe[mu] * 2) == 0 and (global_L[mu] // (pe[mu] * 2)) % 2 == 0 and global_L[mu] // (pe[mu] * 2) >= floor

    while n > 1:
        if n % 2:
            raise ValueError(f"cannot build a PE grid for {ngpu} GPUs on lattice {global_L}")
        for mu, floor in ((3, min_local), (2, min_local), (1, min_local), (3, 2), (2, 2), (1, 2)):
            # t before z at equal split counts (t is the long axis of the lattices of BASELINE.json); a direction is taken again only after the other had its turn
            if can_split(mu, floor) and not (mu == 3 and pe[3] > pe[2] and can_split(2, floor)):
                pe[mu] *= 2
                n //= 2
                break
        else:
            raise ValueError(f"cannot build a PE grid for {ngpu} GPUs on lattice {global_L}")
    return tuple(pe)


def local_view(arr_tzyx_axes_first, local_L, origin, lead=0):
    """Slice the (t,z,y,x) axes (starting at axis `lead`) of a global array down to one rank's sub-lattice."""
    sl = [slice(None)] * arr_tzyx_axes_first.ndim
    for k, mu in enumerate((3, 2, 1, 0)):
        sl[lead + k] = slice(origin[mu], origin[mu] + local_L[mu])
    return np.ascontiguousarray(arr_tzyx_axes_first[tuple(sl)])
