"""PE-grid helpers (host logic, pure Python/numpy).

The reference's only parallelism concept is a 4-D process grid PEs = (px,py,pz,pt) (src/mpirun.jl:17-19,
src/mpi/mpimodule.jl:9-13).  These helpers choose a grid for N GPUs (keep x, the contiguous axis,
unpartitioned; spread faces over as many distinct xGMI peers as possible) and slice global host arrays.
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


def choose_pe_grid(global_L, ngpu):
    """Partition t first, then z, then y; never x; prefer grids touching >= 3 distinct peers at 8 GPUs: (1,2,2,2)."""
    pe = [1, 1, 1, 1]
    n = ngpu
    order = [3, 2, 1]
    i = 0
    guard = 0
    while n > 1:
        mu = order[i % 3]
        if n % 2 == 0 and global_L[mu] % (pe[mu] * 2) == 0 and (global_L[mu] // (pe[mu] * 2)) % 2 == 0:
            pe[mu] *= 2
            n //= 2
            guard = 0
        else:
            guard += 1
            if guard > 3:
                raise ValueError(f"cannot build a PE grid for {ngpu} GPUs on lattice {global_L}")
        i += 1
    return tuple(pe)


def local_view(arr_tzyx_axes_first, local_L, origin, lead=0):
    """Slice the (t,z,y,x) axes (starting at axis `lead`) of a global array down to one rank's sub-lattice."""
    sl = [slice(None)] * arr_tzyx_axes_first.ndim
    for k, mu in enumerate((3, 2, 1, 0)):
        sl[lead + k] = slice(origin[mu], origin[mu] + local_L[mu])
    return np.ascontiguousarray(arr_tzyx_axes_first[tuple(sl)])
