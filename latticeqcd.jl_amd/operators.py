"""Host-side mirror of the LatticeDiracOperators.jl / Gaugefields.jl interface the reference calls on its hot path.

Julia is not available in the build environment, so this Python layer stands where the thin Julia methods of
INTEGRATION.md stand: same names (a trailing `_` replaces Julia's `!`), same argument order and meaning, same error
behaviour (non-convergence and unsupported operators raise), everything forwarded to the C ABI of liblqcd_hip.so.

Reference call sites mirrored (paths relative to /root/reference):
  Initialize_Gaugefields(NC, Nwing, L...; condition)            src/system/universe.jl:41-49
  Initialize_pseudofermion_fields(U[1], "Wilson"|"staggered"|"Domainwall"; L5)   src/system/universe.jl:107,112,128
  Dirac_operator(U, x, params) / D(U) / D'                      src/system/universe.jl:103-137, unusedfiles/measure_chiral_condensate.jl:173
  DdagD_operator, mul!, solve_DinvX!                            SURVEY.md 8(a) a2-a5 (LatticeDiracOperators.jl)
  dot, clear_fermion!, add_fermion!, substitute_fermion!        src/updates/standardHMC.jl:54, src/md/standardMD.jl:50-51
  gauss_distribution_fermion!, Z4_distribution_fermi!           unusedfiles/measure_chiral_condensate.jl:180
  calculate_Plaquette                                           src/system/lqcd.jl:187-193

Host arrays are numpy complex128, C order, with the memory image of the Julia arrays:
  gauge U[mu,t,z,y,x,b,a], Wilson psi[s,t,z,y,x,c], staggered psi[t,z,y,x,c]   (local sub-lattice of this rank).
"""
import ctypes as C
import os

import numpy as np

from . import lib as _l
from .lib import DOMAINWALL, EVEN, FULL, ODD, STAGGERED, WILSON, LQCDError, NotConverged, check  # noqa: F401

_KIND = {"wilson": WILSON, "staggered": STAGGERED, "wilsonclover": WILSON, "domainwall": DOMAINWALL}


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Lattice:
    """One rank's context: global lattice L=(NX,NY,NZ,NT), PE grid (the reference's PEs), rank, device."""

    def __init__(self, L, pe_grid=(1, 1, 1, 1), rank=0, device=0):
        self.L = tuple(int(v) for v in L)
        self.pe = tuple(int(v) for v in pe_grid)
        self.rank = int(rank)
        self.device = int(device)
        self._h = C.c_void_p()
        check(_l.lib().lqcd_ctx_create(C.byref(self._h), self.device, _l.i4(self.L), _l.i4(self.pe), self.rank))
        lo, org, nf, nb = _l.i4([0] * 4), _l.i4([0] * 4), _l.i4([0] * 4), _l.i4([0] * 4)
        check(_l.lib().lqcd_decompose(_l.i4(self.L), _l.i4(self.pe), self.rank, lo, org, nf, nb))
        self.local_L = tuple(lo)
        self.origin = tuple(org)
        self.nranks = int(np.prod(self.pe))
        # the reference's callers discard the temporaries of their per-direction call triples (unused!, AbstractMD.jl:95-97,113-117): this binding
        # switches the library's fusion of those triples on (the plain C ABI is eager by default)
        self.set_param("lazy_links", 1)

    # The per-direction call triples of the reference's U_update! / P_update! (AbstractMD.jl:91-93, 108-110) are recorded and fused BELOW the C ABI
    # (csrc/md.hip "lazy link triples", tunable lazy_links); this binding makes one stateless call per generic.  The three attributes below only
    # expose the library's state to the tests.
    @property
    def lazy_links(self):
        return bool(self.get_param("lazy_links"))

    @lazy_links.setter
    def lazy_links(self, on):
        self.set_param("lazy_links", 1 if on else 0)

    @property
    def _lazy(self):
        k = self.get_param("lazy_open")
        return None if k == 0 else k

    @property
    def _done(self):
        return [None] * self.get_param("lazy_deferred")

    # -- shapes of the local host arrays
    @property
    def gauge_shape(self):
        l = self.local_L
        return (4, l[3], l[2], l[1], l[0], 3, 3)

    def fermion_shape(self, kind):
        l = self.local_L
        return (4, l[3], l[2], l[1], l[0], 3) if kind in (WILSON, DOMAINWALL) else (l[3], l[2], l[1], l[0], 3)      # Domainwall: per slice

    def local_slices(self):
        """numpy slices (t,z,y,x) selecting this rank's sub-lattice out of a global array."""
        o, l = self.origin, self.local_L
        return (slice(o[3], o[3] + l[3]), slice(o[2], o[2] + l[2]), slice(o[1], o[1] + l[1]), slice(o[0], o[0] + l[0]))

    def set_param(self, key, value):
        check(_l.lib().lqcd_ctx_set_param(self._h, key.encode(), int(value)))

    def get_param(self, key):
        v = C.c_int(0)
        check(_l.lib().lqcd_ctx_get_param(self._h, key.encode(), C.byref(v)))
        return v.value

    def sync(self):
        check(_l.lib().lqcd_ctx_sync(self._h))

    def comm_init(self, unique_id):
        assert len(unique_id) == 256
        # test aid: LQCD_SELFCOMM_BACKEND=peer sends a ONE-rank communicator (the self-partitioned tests: LQCD_FORCE_PARTITION) through the peer-mapped
        # backend instead of RCCL, so that the same test bodies cover both (tests/test_gpu_peer_selfmapped.py)
        if self.nranks == 1 and os.environ.get("LQCD_SELFCOMM_BACKEND", "") == "peer":
            return self.comm_init_peer()
        buf = (C.c_ubyte * 256)(*bytes(unique_id))
        check(_l.lib().lqcd_ctx_comm_init(self._h, buf, self.nranks))

    # the peer-mapped backend (csrc/comm.hip): export -> the host gathers the blobs of all ranks in rank order -> init
    def peer_export(self):
        buf = (C.c_ubyte * 256)()
        check(_l.lib().lqcd_ctx_peer_export(self._h, buf))
        return bytes(buf)

    def peer_init(self, blobs):
        """blobs: the 256-byte descriptions of ranks 0..nranks-1 (a list, or their concatenation)."""
        raw = b"".join(bytes(b) for b in blobs) if not isinstance(blobs, (bytes, bytearray)) else bytes(blobs)
        assert len(raw) == 256 * self.nranks, (len(raw), self.nranks)
        buf = (C.c_ubyte * len(raw))(*raw)
        check(_l.lib().lqcd_ctx_peer_init(self._h, buf, self.nranks))

    def comm_init_peer(self, all_gather=None):
        """One call for both steps; all_gather(blob) -> list of every rank's blob in rank order (default: this rank alone, the self-partitioned proxy)."""
        mine = self.peer_export()
        self.peer_init(all_gather(mine) if all_gather else [mine])

    @property
    def comm_backend(self):
        v = C.c_int(0)
        check(_l.lib().lqcd_ctx_comm_backend(self._h, C.byref(v)))
        return {0: "none", 1: "rccl", 2: "peer"}[v.value]

    def close(self):
        if self._h:
            _l.lib().lqcd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):          # the Julia binding registers finalizers (julia/LatticeQCDHIP.jl); same ownership here
        try:
            self.close()
        except Exception:
            pass


def comm_unique_id():
    buf = (C.c_ubyte * 256)()
    check(_l.lib().lqcd_comm_unique_id(buf))
    return bytes(buf)


def link_local(lattices):
    arr = (C.c_void_p * len(lattices))(*[lat._h for lat in lattices])
    check(_l.lib().lqcd_ctx_link_local(arr, len(lattices)))


# ------------------------------------------------------------------------------------ gauge fields
class Gaugefields:
    """The four link fields U[1:4] of the reference as one device object."""

    def __init__(self, lattice):
        self.lattice = lattice
        self.NC = 3
        self._h = C.c_void_p()
        check(_l.lib().lqcd_gauge_create(lattice._h, C.byref(self._h)))

    def upload(self, U, layout=_l.LAYOUT_REFERENCE, nwing=0):
        """nwing > 0: U has the reference's winged shape (4, NT+2w, NZ+2w, NY+2w, NX+2w, 3, 3) (Nwing of universe.jl:41-49)."""
        U = np.ascontiguousarray(U, dtype=np.complex128)
        if nwing:
            l = self.lattice.local_L
            assert U.shape == (4, l[3] + 2 * nwing, l[2] + 2 * nwing, l[1] + 2 * nwing, l[0] + 2 * nwing, 3, 3), U.shape
            check(_l.lib().lqcd_gauge_upload_wing(self._h, _ptr(U), int(nwing)))
            return self
        assert U.size == int(np.prod(self.lattice.gauge_shape)), (U.shape, self.lattice.gauge_shape)
        check(_l.lib().lqcd_gauge_upload(self._h, _ptr(U), int(layout)))
        return self

    def download(self, layout=_l.LAYOUT_REFERENCE, nwing=0, into=None):
        if nwing:
            l = self.lattice.local_L
            U = into if into is not None else np.zeros((4, l[3] + 2 * nwing, l[2] + 2 * nwing, l[1] + 2 * nwing, l[0] + 2 * nwing, 3, 3), dtype=np.complex128)
            check(_l.lib().lqcd_gauge_download_wing(self._h, _ptr(U), int(nwing)))
            return U
        U = np.empty(self.lattice.gauge_shape, dtype=np.complex128)
        check(_l.lib().lqcd_gauge_download(self._h, _ptr(U), int(layout)))
        return U

    # -- the reference indexes its Vector of link fields one direction at a time: U[mu], p[mu], mu = 1..Dim (AbstractMD.jl:90-93)
    def __getitem__(self, mu):
        if not 1 <= int(mu) <= 4:
            raise IndexError("link fields are indexed mu = 1..4 like the reference's U[mu]")
        return LinkView(self, int(mu) - 1)

    def __len__(self):
        return 4

    def similar(self):
        return Gaugefields(self.lattice)

    def __mul__(self, other):
        """md.p * md.p (standardHMC.jl:49) for a momentum field: sum_a p_a^2 = 2 K."""
        if other is not self:
            raise LQCDError(_l.ERR_UNSUPPORTED, "only p * p (the kinetic term of standardHMC.jl:49) is defined")
        return 2.0 * momentum_action(self)

    def close(self):
        if self._h:
            _l.lib().lqcd_gauge_destroy(self._h)      # recorded link operations that name this field run first (inside the library)
            self._h = C.c_void_p()

    def __del__(self):          # the Julia binding registers finalizers (julia/LatticeQCDHIP.jl); same ownership here
        try:
            self.close()
        except Exception:
            pass


class LinkView:
    """One direction of a gauge-shaped device field: what the reference's callers hold as U[mu], p[mu] or a temporary link field
    (AbstractMD.jl:78-135).  Nothing is copied: the C ABI's single-direction entry points take (field, direction slot)."""

    def __init__(self, field, slot):
        self.field, self.slot = field, slot
        self.NC = 3
        self.lattice = field.lattice

    def download(self):
        return self.field.download()[self.slot]

    def adjoint(self):
        """U[mu]' as the first factor of mul!(C, U[mu]', B) (standardMD.jl:211)."""
        return AdjointLinkView(self.field, self.slot)

    H = property(adjoint)


class AdjointLinkView:
    def __init__(self, field, slot):
        self.field, self.slot = field, slot


def Initialize_Gaugefields(NC, Nwing, NX, NY, NZ, NT, condition="cold", lattice=None, randomseed=111, **kw):
    """Gaugefields.jl Initialize_Gaugefields(NC,Nwing,NX,NY,NZ,NT; condition) (universe.jl:41-49). NC = 3 only;
    Nwing is accepted and ignored (device fields carry no wing)."""
    if NC != 3:
        raise LQCDError(_l.ERR_UNSUPPORTED, "only NC = 3 is supported on the HIP path")
    lat = lattice if lattice is not None else Lattice((NX, NY, NZ, NT), **kw)
    U = Gaugefields(lat)
    if condition == "cold":
        check(_l.lib().lqcd_gauge_unit(U._h))
    elif condition == "hot":
        check(_l.lib().lqcd_gauge_hot_start(U._h, C.c_uint64(int(randomseed))))
    else:
        raise LQCDError(_l.ERR_ARG, f"condition = {condition} is not supported")
    return U


def calculate_Plaquette(U):
    """Plaquette normalised by 1/(6 V NC) (lqcd.jl:187-193 with factor 1/(comb*NV*NC))."""
    p = C.c_double(0)
    check(_l.lib().lqcd_gauge_plaquette(U._h, C.byref(p)))
    return p.value


def calculate_Polyakov_loop(U, temp1=None, temp2=None):
    """calculate_Polyakov_loop(U, temp1, temp2) (the Polyakov_loop measurement of the reference's runs): 1/(NC NX NY NZ) sum_x tr prod_t U_4(x, t)."""
    re, im = C.c_double(0), C.c_double(0)
    check(_l.lib().lqcd_gauge_polyakov(U._h, C.byref(re), C.byref(im)))
    return complex(re.value, im.value)


def reunitarize_(U):
    """Every link back onto SU(3) (lqcd_gauge_reunitarize; no reference counterpart)."""
    check(_l.lib().lqcd_gauge_reunitarize(U._h))


def unitarity_deviation(U):
    """max |row2 - conj(row0 x row1)| over all links (the gate of the 12-real link path); diagnostic, no reference counterpart."""
    d = C.c_double(0)
    check(_l.lib().lqcd_gauge_unitarity_deviation(U._h, C.byref(d)))
    return d.value


# ------------------------------------------------------------------------------------ fermion fields
class Fermionfields:
    """A pseudofermion field.  kind = DOMAINWALL: L5 Wilson fields in one allocation; `x.w[i5]` (the reference's name for a slice) is a Wilson field that
    aliases slice i5, host arrays are [L5][s,t,z,y,x,c]."""

    def __init__(self, lattice, kind, subset=FULL, L5=None, _slice_of=None):
        self.lattice = lattice
        self.kind = kind
        self.subset = subset
        self.L5 = L5
        self._h = C.c_void_p()
        self._parent = None
        if _slice_of is not None:
            parent, i5 = _slice_of
            self._parent = parent           # the view owns nothing: the parent stays alive as long as the view does
            check(_l.lib().lqcd_spinor_slice(parent._h, int(i5), C.byref(self._h)))
        elif kind == DOMAINWALL:
            if not L5:
                raise LQCDError(_l.ERR_ARG, "a Domainwall field needs L5")
            check(_l.lib().lqcd_spinor_create_5d(lattice._h, C.byref(self._h), int(L5)))
        else:
            check(_l.lib().lqcd_spinor_create(lattice._h, C.byref(self._h), int(kind), int(subset)))

    @property
    def w(self):
        if self.kind != DOMAINWALL:
            raise AttributeError("w: only five-dimensional fields have slices")
        return [Fermionfields(self.lattice, WILSON, FULL, _slice_of=(self, i5)) for i5 in range(self.L5)]

    def upload(self, psi, nwing=0):
        """nwing > 0: psi carries the reference's wing (fields created without nowing = true, universe.jl:107)."""
        psi = np.ascontiguousarray(psi, dtype=np.complex128)
        if self.kind == DOMAINWALL:
            assert psi.shape == (self.L5,) + self.lattice.fermion_shape(WILSON), psi.shape
            for i5, v in enumerate(self.w):
                v.upload(psi[i5])
            return self
        if nwing:
            check(_l.lib().lqcd_spinor_upload_wing(self._h, _ptr(psi), int(nwing)))
            return self
        assert psi.shape == self.lattice.fermion_shape(self.kind), (psi.shape, self.lattice.fermion_shape(self.kind))
        check(_l.lib().lqcd_spinor_upload(self._h, _ptr(psi)))
        return self

    def download_wing(self, into, nwing):
        check(_l.lib().lqcd_spinor_download_wing(self._h, _ptr(into), int(nwing)))
        return into

    def download(self, into=None):
        if self.kind == DOMAINWALL:
            out = np.zeros((self.L5,) + self.lattice.fermion_shape(WILSON), dtype=np.complex128) if into is None else into
            for i5, v in enumerate(self.w):
                v.download(out[i5])
            return out
        out = np.zeros(self.lattice.fermion_shape(self.kind), dtype=np.complex128) if into is None else into
        check(_l.lib().lqcd_spinor_download(self._h, _ptr(out)))
        return out

    def similar(self):
        return Fermionfields(self.lattice, self.kind, self.subset, L5=self.L5)

    def close(self):
        if self._h:
            _l.lib().lqcd_spinor_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):          # the Julia binding registers finalizers (julia/LatticeQCDHIP.jl); same ownership here
        try:
            self.close()
        except Exception:
            pass


def Initialize_pseudofermion_fields(U, Dirac_operator, nowing=True, subset=FULL, L5=None):
    """Initialize_pseudofermion_fields(U[1], "Wilson" | "staggered" | "Domainwall"; nowing, L5) (universe.jl:107,112,128)."""
    key = Dirac_operator.lower()
    if key not in _KIND:
        raise LQCDError(_l.ERR_UNSUPPORTED, f"{Dirac_operator} is not supported")
    return Fermionfields(U.lattice, _KIND[key], subset, L5=L5)


def clear_fermion_(x):
    check(_l.lib().lqcd_spinor_zero(x._h))


def substitute_fermion_(dst, src):
    check(_l.lib().lqcd_spinor_copy(dst._h, src._h))


def extract_fermion_(half, full):
    """half (EVEN|ODD subset) <- the sites of that parity of a FULL field."""
    check(_l.lib().lqcd_spinor_extract(half._h, full._h))
    return half


def insert_fermion_(full, half):
    """The sites of half's parity of a FULL field <- half (the other parity is left untouched)."""
    check(_l.lib().lqcd_spinor_insert(full._h, half._h))
    return full


def gauss_distribution_fermion_(x, randomseed=112):
    check(_l.lib().lqcd_spinor_gaussian(x._h, C.c_uint64(int(randomseed))))


def Z4_distribution_fermi_(x, randomseed=113):
    check(_l.lib().lqcd_spinor_z4(x._h, C.c_uint64(int(randomseed))))


def setindex_global_(x, ic, ix, iy, iz, it, ispin):
    """Point source b[ic,ix,iy,iz,it,is] = 1 at a GLOBAL site, all indices 0-based (measure_Pion_correlator.jl:376)."""
    check(_l.lib().lqcd_spinor_point_source(x._h, _l.i4((ix, iy, iz, it)), int(ic), int(ispin)))


def dot(a, b):
    """Hermitian inner product sum conj(a) b (standardHMC.jl:54)."""
    re, im = C.c_double(0), C.c_double(0)
    check(_l.lib().lqcd_dot(a._h, b._h, C.byref(re), C.byref(im)))
    return complex(re.value, im.value)


def add_fermion_(c, alpha, a, beta=None, b=None):
    """add_fermion!(c, alpha, a[, beta, b]):  c += alpha*a (+ beta*b)."""
    al = complex(alpha)
    check(_l.lib().lqcd_axpy(C.c_double(al.real), C.c_double(al.imag), a._h, c._h))
    if b is not None:
        be = complex(beta)
        check(_l.lib().lqcd_axpy(C.c_double(be.real), C.c_double(be.imag), b._h, c._h))


# ------------------------------------------------------------------------------------ Dirac operators
class Dirac_operator:
    """Dirac_operator(U, x, params::Dict) (universe.jl:137).  Keys read: "Dirac_operator", "κ"/"kappa", "r", "mass",
    "eps_CG", "MaxCGstep", "boundarycondition", "method_CG".  `D(U)` rebinds the links, `D.adjoint()` is D'."""

    def __init__(self, U, x, params, _dagger=False, _share=None):
        self.params = dict(params)
        name = self.params.get("Dirac_operator", "Wilson")
        key = name.lower()
        if key not in _KIND:
            raise LQCDError(_l.ERR_UNSUPPORTED, f"{name} is not supported")  # universe.jl:129-131
        self.kind = _KIND[key]
        self.U = U
        self.lattice = U.lattice
        self.dagger = _dagger
        self.eps_CG = float(self.params.get("eps_CG", 1e-19))          # parameter_structs.jl:174
        self.MaxCGstep = int(self.params.get("MaxCGstep", 3000))        # parameter_structs.jl:175
        self.method_CG = self.params.get("method_CG", "bicgstab")
        self.bc = tuple(self.params.get("boundarycondition", (1, 1, 1, -1)))  # parameter_structs.jl:133
        self.L5 = None
        if self.kind == WILSON:
            self.km = float(self.params.get("κ", self.params.get("kappa", 0.141139)))  # parameter_structs.jl:126
            self.r = float(self.params.get("r", 1.0))
        elif self.kind == DOMAINWALL:      # universe.jl:116-128: "mass" = Domainwall_m, "L5", "M" = Domainwall_M
            self.km = float(self.params.get("mass", 0.25))
            self.M = float(self.params.get("M", -1.0))
            self.L5 = int(self.params.get("L5", 4))
            self.r = 1.0
        else:
            self.km = float(self.params.get("mass", 0.5))
            self.r = 1.0
        if _share is not None:
            self._h, self._owner = _share, False
        else:
            self._h, self._owner = C.c_void_p(), True
            if self.kind == DOMAINWALL:
                check(_l.lib().lqcd_op_create_domainwall(self.lattice._h, C.byref(self._h), U._h, C.c_double(self.M), C.c_double(self.km), self.L5,
                                                         _l.i4(self.bc)))
                return
            check(_l.lib().lqcd_op_create(self.lattice._h, C.byref(self._h), self.kind, U._h, C.c_double(self.km),
                                          C.c_double(self.r), _l.i4(self.bc)))
            if key == "wilsonclover":     # Dirac_operator = "WilsonClover", Clover_coefficient (parameter_structs.jl:125)
                self.csw = float(self.params.get("Clover_coefficient", 1.5612))
                check(_l.lib().lqcd_op_set_clover(self._h, C.c_double(self.csw)))

    def __call__(self, U):
        check(_l.lib().lqcd_op_set_gauge(self._h, U._h))
        self.U = U
        return self

    def adjoint(self):
        adj = Dirac_operator(self.U, None, self.params, _dagger=not self.dagger, _share=self._h)
        adj._parent = self        # the handle belongs to the parent: keep it alive as long as the adjoint view is
        adj.eps_CG, adj.MaxCGstep, adj.method_CG = self.eps_CG, self.MaxCGstep, self.method_CG
        return adj

    H = property(adjoint)

    def close(self):
        if self._owner and self._h:
            _l.lib().lqcd_op_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):          # the Julia binding registers finalizers (julia/LatticeQCDHIP.jl); same ownership here
        try:
            self.close()
        except Exception:
            pass


class DdagD_operator:
    """DdagD_operator(D): A = D'D, solved with CG."""

    def __init__(self, D):
        self.D = D
        self.eps_CG, self.MaxCGstep = D.eps_CG, D.MaxCGstep


def _mul_links(C_, A, B):
    """mul!(W, expU, U[mu]) / mul!(temp1, U[mu], dSdUmu) (AbstractMD.jl:92,109): 3x3 products site by site (second call of a lazy triple: the
    library records it, csrc/md.hip)."""
    if isinstance(A, AdjointLinkView):
        check(_l.lib().lqcd_link_mul_adj(C_.field._h, C_.slot, A.field._h, A.slot, B.field._h, B.slot))
        return C_
    check(_l.lib().lqcd_link_mul(C_.field._h, C_.slot, A.field._h, A.slot, B.field._h, B.slot))
    return C_


def mul_(y, A, x):
    """LinearAlgebra.mul!(y, A, x) for A = D, D' or D'D; on link fields mul!(W, A, B) = the site-wise 3x3 product."""
    if isinstance(y, LinkView):
        return _mul_links(y, A, x)
    if isinstance(A, DdagD_operator):
        check(_l.lib().lqcd_op_apply_DdagD(A.D._h, y._h, x._h))
    else:
        check(_l.lib().lqcd_op_apply(A._h, y._h, x._h, int(A.dagger)))
    return y


def hop_(y, D, x):
    """Parity hop: y (EVEN|ODD) = H x (opposite parity)."""
    check(_l.lib().lqcd_op_hop(D._h, y._h, x._h, int(D.dagger)))
    return y


def solve_DinvX_(y, A, x, return_info=False):
    """solve_DinvX!(y, A, x): y = A^{-1} x.  A::DdagD_operator -> CG; A::Dirac_operator -> BiCGStab
    ("bicgstab"), its even-odd preconditioned form ("bicgstab_evenodd") or BiCG ("bicg", the reference's default).  Stopping rule real(r.r) < eps_CG;
    raises NotConverged after MaxCGstep (the reference raises error(...))."""
    it, rr = C.c_int(0), C.c_double(0)
    L = _l.lib()
    if isinstance(A, DdagD_operator):
        st = L.lqcd_solve_cg_DdagD(A.D._h, y._h, x._h, C.c_double(A.eps_CG), A.MaxCGstep, C.byref(it), C.byref(rr))
    elif A.method_CG in ("bicgstab_evenodd", "preconditiond_bicgstab"):
        st = L.lqcd_solve_bicgstab_eo(A._h, y._h, x._h, int(A.dagger), C.c_double(A.eps_CG), A.MaxCGstep, C.byref(it), C.byref(rr))
    elif A.method_CG == "bicg":
        st = L.lqcd_solve_bicg(A._h, y._h, x._h, int(A.dagger), C.c_double(A.eps_CG), A.MaxCGstep, C.byref(it), C.byref(rr))
    elif A.method_CG == "bicgstab":
        st = L.lqcd_solve_bicgstab(A._h, y._h, x._h, int(A.dagger), C.c_double(A.eps_CG), A.MaxCGstep, C.byref(it), C.byref(rr))
    else:
        raise LQCDError(_l.ERR_ARG, f"method_CG = {A.method_CG} is not supported")
    check(st)
    return (it.value, rr.value) if return_info else None


def solve_mixed_DinvX_(y, A, x, inner_tol=0.0, return_info=False):
    """Mixed-precision variant of solve_DinvX!(y, A::DdagD_operator, x): fp32 inner CG, fp64 defect correction; y holds the
    initial guess; the stopping rule real(r.r) < eps_CG is enforced on the true fp64 residual.
    return_info -> (total inner iterations, outer steps, true |r|^2)."""
    if not isinstance(A, DdagD_operator):
        raise LQCDError(_l.ERR_ARG, "solve_mixed_DinvX_ needs a DdagD_operator")
    it, out, rr = C.c_int(0), C.c_int(0), C.c_double(0)
    check(_l.lib().lqcd_solve_mixed_cg_DdagD(A.D._h, y._h, x._h, C.c_double(A.eps_CG), A.MaxCGstep, C.c_double(inner_tol), C.byref(it),
                                             C.byref(out), C.byref(rr)))
    return (it.value, out.value, rr.value) if return_info else None


def mul_f32_(y, D, x, reps=0):
    """y = D x through the fp32 operator of the mixed-precision solvers (lqcd_op_apply_f32; diagnostic, no reference counterpart).
    reps > 0 -> returns the mean time of one fp32 application in ms."""
    ms = C.c_double(0)
    check(_l.lib().lqcd_op_apply_f32(D._h, y._h, x._h, int(D.dagger), int(reps), C.byref(ms)))
    return ms.value if reps else None


def solve_parity_DinvX_(y, A, x, parity=0, return_info=False):
    """Staggered only: y_p = ((D'D)_pp)^-1 x_p on the sites of one parity (0 even, 1 odd) of the FULL fields y (initial guess) and x,
    with half-lattice vectors -- D'D = m^2 - D_hop^2 is block diagonal in parity.  The other half of y is left alone."""
    if not isinstance(A, DdagD_operator):
        raise LQCDError(_l.ERR_ARG, "solve_parity_DinvX_ needs a DdagD_operator")
    it, rr = C.c_int(0), C.c_double(0)
    check(_l.lib().lqcd_solve_cg_DdagD_parity(A.D._h, y._h, x._h, int(parity), C.c_double(A.eps_CG), A.MaxCGstep, C.byref(it), C.byref(rr)))
    return (it.value, rr.value) if return_info else None


def shiftedcg(vec_x, vec_beta, x, A, b, eps=None, maxsteps=None, return_info=False):
    """shiftedcg(vec_x, vec_β, x, A, b): (A + β_j) vec_x[j] = b for every shift and A x = b, A = D'D (RHMC; README.md:132).
    Zero initial guesses; raises NotConverged after maxsteps."""
    if not isinstance(A, DdagD_operator):
        raise LQCDError(_l.ERR_ARG, "shiftedcg needs a DdagD_operator")
    sig = (C.c_double * len(vec_beta))(*[float(v) for v in vec_beta])
    it, rr = C.c_int(0), C.c_double(0)
    check(_l.lib().lqcd_solve_multishift_cg(A.D._h, x._h if x is not None else None, _harr(vec_x), b._h, sig, len(vec_beta),
                                            C.c_double(A.eps_CG if eps is None else eps),
                                            int(A.MaxCGstep if maxsteps is None else maxsteps), C.byref(it), C.byref(rr)))
    return (it.value, rr.value) if return_info else None


def shiftedcg_mixed(vec_x, vec_beta, x, A, b, eps=None, maxsteps=None, inner_tol=0.0, return_info=False):
    """Mixed-precision shiftedcg (BASELINE configs[4]; no counterpart in the reference, whose shiftedcg is fp64): same arguments and
    contract as shiftedcg, the stopping rule real(r.r) < eps holds for the TRUE fp64 residual of every shifted system.  One fp32
    multi-shift CG for all shifts, then fp64 defect correction per shift (lqcd_solve_multishift_mixed_cg).
    return_info: (fp32 iterations, fp32 correction solves, largest true residual)."""
    if not isinstance(A, DdagD_operator):
        raise LQCDError(_l.ERR_ARG, "shiftedcg_mixed needs a DdagD_operator")
    sig = (C.c_double * len(vec_beta))(*[float(v) for v in vec_beta])
    it, outer, rr = C.c_int(0), C.c_int(0), C.c_double(0)
    check(_l.lib().lqcd_solve_multishift_mixed_cg(A.D._h, x._h if x is not None else None, _harr(vec_x), b._h, sig, len(vec_beta),
                                                  C.c_double(A.eps_CG if eps is None else eps),
                                                  int(A.MaxCGstep if maxsteps is None else maxsteps), C.c_double(inner_tol),
                                                  C.byref(it), C.byref(outer), C.byref(rr)))
    return (it.value, outer.value, rr.value) if return_info else None


def apply_inverse_power_(y, A, x, alpha, lam_min, lam_max, tol=1e-10):
    """y = (D'D)^(-alpha) x, 0 < alpha < 1, through ONE multi-shift solve with the partial fractions of rational.py -- the building
    block of the RHMC action and heat bath (README.md:112,132).  [lam_min, lam_max] must enclose the spectrum of D'D
    (staggered: [m^2, m^2 + 16]).  Returns the number of CG iterations."""
    from . import rational
    a0, res, poles, _ = rational.inverse_power_partial_fractions(alpha, lam_min, lam_max, tol)
    xs = [x.similar() for _ in poles]
    it, _ = shiftedcg(xs, list(poles), None, A, x, return_info=True)
    substitute_fermion_(y, x)
    check(_l.lib().lqcd_scale(C.c_double(a0), C.c_double(0.0), y._h))
    for r, xk in zip(res, xs):
        add_fermion_(y, float(r), xk)
        xk.close()
    return it


# ------------------------------------------------------------------------------------ pseudofermion action and force
class FermiAction:
    """FermiAction(D, Dict("Nf"=>...)) (universe.jl:138): the pseudofermion action S_f = eta' (D'D)^-1 eta.
    Wilson: Nf = 2.  Staggered: Nf = 8 (eta on every site) or Nf = 4 (the reference's "4 tastes", test/test_staggered.toml:
    eta lives on the even sites only -- D'D = m^2 - D_hop^2 is block diagonal in parity, so the solve and the force are the same
    kernels with the odd half of eta zero; sample_pseudofermions_ does the masking).
    Staggered with any other 0 < Nf < 8 (test/test_Nf2.toml:8, test/test_Nf3.toml:8) is the rational action
    S_f = eta' (D'D)^(-Nf/8) eta; Wilson / Wilson-clover with 0 < Nf < 2 (Nf = 1: the strange quark of a 2+1 run) is
    S_f = eta' (D'D)^(-Nf/2) eta on an interval from "rhmc_lambda_min" / "rhmc_lambda_max" or from a Lanczos estimate
    (estimate_spectrum) on the links at construction, with margins 0.5 / 1.2 -- the staggered interval is analytic.
    Partial fractions from rational.py on the spectral interval [m^2, m^2 + 16] (|D_hop| <= 4), a
    tighter fit for the action and the heat bath ("rhmc_tol_action", default 1e-12) than for the MD force ("rhmc_tol_MD", 1e-8),
    one multi-shift solve per evaluation (lqcd_rational_apply / lqcd_rational_force).
    Keeps X = (D'D)^-1 eta and Y = D X resident between evaluate_FermiAction and calc_UdSfdU_."""

    _PARAM_KEYS = ("force_rational", "rhmc_lambda_min", "rhmc_lambda_max", "rhmc_tol_action", "rhmc_tol_MD", "rhmc_lanczos_steps")

    def __init__(self, D, params=None):
        params = params or {}
        self.D = D
        keys = [k for k in self._PARAM_KEYS if k in params]
        ckeys = (C.c_char_p * max(len(keys), 1))(*[k.encode() for k in keys])
        cvals = (C.c_double * max(len(keys), 1))(*[float(params[k]) for k in keys])
        h = C.c_void_p()
        # the whole construction -- which Nf is exact and which is rational, the spectral interval, the three fits -- happens below the C ABI
        # (csrc/rational.hip lqcd_action_create); the Julia binding makes the same call
        check(_l.lib().lqcd_action_create(D._h, C.c_double(float(params.get("Nf", 0))), C.c_double(D.eps_CG), int(D.MaxCGstep), len(keys), ckeys, cvals,
                                          C.byref(h)))
        self._fa = h
        self.Nf = self._get("Nf")
        if self.Nf == int(self.Nf):
            self.Nf = int(self.Nf)
        self.rational = bool(self._get("rational"))
        self.evensite = bool(self._get("evensite"))
        self.alpha = self._get("alpha")
        self._temporary_fermionfields = [Fermionfields(D.lattice, D.kind, L5=D.L5) for _ in range(2)]   # standardMD.jl:50-51

    def _get(self, key):
        v = C.c_double(0)
        check(_l.lib().lqcd_action_get(self._fa, key.encode(), C.byref(v)))
        return v.value

    @property
    def _h(self):          # the action handle, with the operator's CURRENT stopping rule (D.eps_CG / D.MaxCGstep are plain attributes)
        check(_l.lib().lqcd_action_set_solver(self._fa, C.c_double(self.D.eps_CG), int(self.D.MaxCGstep)))
        return self._fa

    def _coeffs(self, which):
        n, a0 = C.c_int(0), C.c_double(0)
        check(_l.lib().lqcd_action_coefficients(self._fa, which, None, None, None, 0, C.byref(n), None))
        res, poles = (C.c_double * n.value)(), (C.c_double * n.value)()
        check(_l.lib().lqcd_action_coefficients(self._fa, which, C.byref(a0), res, poles, n.value, C.byref(n), None))
        return a0.value, np.array(res[:]), np.array(poles[:])

    def _set_coeffs(self, which, coeffs):
        a0, res, poles = coeffs
        check(_l.lib().lqcd_action_set_coefficients(self._fa, which, C.c_double(a0), len(res), _darr(res), _darr(poles)))

    # (a0, residues, poles) of x^(-Nf/n0) for the action / for the MD force (looser) and of x^(Nf/2n0 - 1) for the heat bath
    rhmc_action = property(lambda self: self._coeffs(0), lambda self, c: self._set_coeffs(0, c))
    rhmc_MD = property(lambda self: self._coeffs(1), lambda self, c: self._set_coeffs(1, c))
    rhmc_sampling = property(lambda self: self._coeffs(2), lambda self, c: self._set_coeffs(2, c))

    @property
    def spectral_interval(self):
        return self._get("lambda_min"), self._get("lambda_max")

    @property
    def interval_refits(self):
        return int(self._get("interval_refits"))

    def check_interval(self, U):
        """Wilson(-clover) rational action: Lanczos on the CURRENT links; an estimated interval that has become too tight is refitted, a Ritz
        value outside an interval fixed with rhmc_lambda_min / rhmc_lambda_max raises (lqcd_action_check_interval)."""
        self.D(U)
        check(_l.lib().lqcd_action_check_interval(self._h))

    def close(self):
        for f in self._temporary_fermionfields:
            f.close()
        if self._fa is not None:
            _l.lib().lqcd_action_destroy(self._fa)
            self._fa = None


def _darr(v):
    return (C.c_double * len(v))(*[float(t) for t in v])


def estimate_spectrum(A, steps=60, randomseed=4711):
    """Extreme Ritz values (theta_min, theta_max) of the Hermitian positive A = D'D from `steps` Lanczos iterations on the device
    (lqcd_estimate_spectrum): theta_max converges to the largest eigenvalue from below, theta_min to the smallest from above -- use
    them with a margin.  The Wilson rational action takes its fit interval from here when none is given."""
    if not isinstance(A, DdagD_operator):
        raise LQCDError(_l.ERR_ARG, "estimate_spectrum needs a DdagD_operator")
    lo, hi = C.c_double(0), C.c_double(0)
    check(_l.lib().lqcd_estimate_spectrum(A.D._h, int(steps), C.c_uint64(randomseed), C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def _rational_apply(D, y, x, coeffs):
    """y = a0 x + sum_k res_k (D'D + pole_k)^-1 x on the device; returns the multi-shift iteration count."""
    a0, res, poles = coeffs
    it = C.c_int(0)
    check(_l.lib().lqcd_rational_apply(D._h, y._h, x._h, C.c_double(a0), len(res), _darr(res), _darr(poles), C.c_double(D.eps_CG), D.MaxCGstep,
                                       C.byref(it)))
    return it.value


def gauss_sampling_in_action_(xi, U, fa, randomseed=112):
    """gauss_sampling_in_action!(xi, U, fa) (standardMD.jl:95): xi distributed as exp(-xi'xi), i.e. <|xi_i|^2> = 1 (re and im
    of variance 1/2) -- NOT the unit-variance-per-real-part noise of gauss_distribution_fermion_; with the wrong variance the
    pseudofermion weight is exp(-S_f/2) and the HMC equilibrates to the wrong plaquette (tests/test_gpu_md.py)."""
    check(_l.lib().lqcd_action_gauss_sampling(fa._fa, xi._h, C.c_uint64(randomseed)))
    return xi


def sample_pseudofermions_(eta, U, fa, xi):
    """sample_pseudofermions!(eta, U, fa, xi) (standardMD.jl:96): eta = D' xi (restricted to the even sites for 4 staggered tastes,
    in which case the action at the start of a trajectory is evaluate_FermiAction(fa, U, eta), not xi'xi); rational action:
    eta = (D'D)^(Nf/2n0) xi, so that eta' (D'D)^(-Nf/n0) eta = xi' xi."""
    check(_l.lib().lqcd_action_sample_pseudofermions(fa._h, U._h, eta._h, xi._h))
    fa.D.U = U
    return eta


def evaluate_FermiAction(fa, U, eta, return_info=False):
    """evaluate_FermiAction(fa, U, eta) (standardHMC.jl:71): S_f = eta' (D'D)^-1 eta (CG from a zero guess) or its rational form."""
    X, Y = fa._temporary_fermionfields
    S, it = C.c_double(0), C.c_int(0)
    check(_l.lib().lqcd_action_evaluate(fa._h, U._h, eta._h, X._h, Y._h, C.byref(S), C.byref(it)))
    fa.D.U = U
    return (S.value, it.value) if return_info else S.value


def calc_UdSfdU_(UdSfdU, fa, U, eta):
    """calc_UdSfdU!(UdSfdU, fa, U, eta) (AbstractMD.jl:129): UdSfdU[mu](n) = "U dS_f/dU" with
    dS_f/d eps under U_mu(n) -> exp(i eps T) U_mu(n) equal to -2 Im tr(T UdSfdU_mu(n)).  UdSfdU is a Gaugefields-shaped field, or
    the reference's Vector of Dim temporary link fields (get_temp(temps, Dim), AbstractMD.jl:123): the sweep then writes a field
    owned by the action and each direction is handed over with one device copy IN THE CALLER'S SIGN CONVENTION: the reference
    adds factor = -eps dtau times the traceless anti-Hermitian part (AbstractMD.jl:127-132), i.e. its "U dS_f/dU" is -G of the C ABI
    (just as its U dS_g/dU = U[mu] calc_dSdUmu! is -NC times the gauge force field G of lqcd_gauge_force)."""
    if isinstance(UdSfdU, (list, tuple)):
        if getattr(fa, "_force_field", None) is None:
            fa._force_field = Gaugefields(fa.D.lattice)
        r = calc_UdSfdU_(fa._force_field, fa, U, eta)
        for mu, view in enumerate(UdSfdU):
            check(_l.lib().lqcd_link_scaled_copy(view.field._h, view.slot, C.c_double(-1.0), fa._force_field._h, mu))
        return r
    S, it = C.c_double(0), C.c_int(0)
    check(_l.lib().lqcd_action_force(fa._h, U._h, UdSfdU._h, eta._h, C.byref(S), C.byref(it)))
    fa.D.U = U
    return None if fa.rational else S.value


def fermion_force_(UdSfdU, D, X, Y, scale=1.0, accumulate=False):
    """The outer-product sweep alone, from resident X = (D'D)^-1 eta and Y = D X; UdSfdU = (accumulate ? UdSfdU : 0) + scale * G."""
    check(_l.lib().lqcd_fermion_force_acc(D._h, UdSfdU._h, X._h, Y._h, C.c_double(scale), int(bool(accumulate))))
    return UdSfdU


# ------------------------------------------------------------------------------------ gauge side of the MD step
def substitute_U_(dst, src):
    """substitute_U!(Uold, U) (standardHMC.jl:45) on the Vector of link fields, substitute_U!(U[mu], W) (AbstractMD.jl:93) on one (third call of
    the U_update! triple: the library turns exptU! -> mul! -> substitute_U! into one pass, csrc/md.hip)."""
    if isinstance(dst, LinkView):
        check(_l.lib().lqcd_link_copy(dst.field._h, dst.slot, src.field._h, src.slot))
        return dst
    check(_l.lib().lqcd_gauge_copy(dst._h, src._h))
    return dst


# ------------------------------------------------------------------------------------ stout smearing of the fermion action's links
class STOUT_Layer:
    """STOUT_Layer(p.stout_loops, p.stout_ρ, U) (universe.jl:153): one stout layer.  Served: the plaquette loop (one rho)."""

    def __init__(self, loops, rho, U=None):
        loops = [loops] if isinstance(loops, str) else list(loops)
        rho = [rho] if np.isscalar(rho) else list(rho)
        if [l.lower() for l in loops] != ["plaquette"] or len(rho) != 1:
            raise LQCDError(_l.ERR_UNSUPPORTED, f"STOUT_Layer: loops {loops} with rho {rho} are not supported (the plaquette loop with one rho is)")
        self.rho = float(rho[0])


class CovNeuralnet:
    """CovNeuralnet(U) (universe.jl:150): the stack of smearing layers between the links of the MD and the links the fermion action sees."""

    def __init__(self, U):
        self.lattice = U.lattice
        self.layers = []
        self._out = []           # the links after each layer (Uout_multi), owned by the net
        self._force = None       # force field carried back through the layers
        self._bare = None

    def push_(self, layer):
        self.layers.append(layer)
        self._out.append(Gaugefields(self.lattice))
        return self


def calc_smearedU(U, nn):
    """calc_smearedU(U, cov_neural_net) (standardMD.jl:91,207; standardHMC.jl:68) -> (Uout, Uout_multi, nothing); Uout_multi[k] = the links after layer k + 1."""
    if nn is None:      # update! reaches this with cov_neural_net = nothing (standardHMC.jl:67 compares the VALUE nothing with the TYPE Nothing): the links themselves
        return U, None, None
    cur = U
    for layer, out in zip(nn.layers, nn._out):
        check(_l.lib().lqcd_stout_smear(out._h, cur._h, C.c_double(layer.rho)))
        cur = out
    return cur, list(nn._out), None


def back_prop(dSdU, nn, Uout_multi, U):
    """back_prop(md.dSdU, cov_neural_net, Uout_multi, U) (standardMD.jl:216): dSdU[mu] = Uout[mu]' * (Uout dS/dUout)[mu] at the smeared links in, dSdUbare out, with
    U[mu] * dSdUbare[mu] the same derivative with respect to the thin links (the caller multiplies and adds its traceless anti-Hermitian part to p[mu], :220-224)."""
    if not nn.layers:
        return dSdU
    if nn._force is None:
        nn._force, nn._bare = Gaugefields(nn.lattice), Gaugefields(nn.lattice)
    Uout = Uout_multi[-1]
    for mu in range(4):
        check(_l.lib().lqcd_link_mul(nn._force._h, mu, Uout._h, mu, dSdU._h, mu))
    for k in reversed(range(len(nn.layers))):
        thin = U if k == 0 else Uout_multi[k - 1]
        check(_l.lib().lqcd_stout_backprop(nn._force._h, nn._force._h, thin._h, C.c_double(nn.layers[k].rho)))
    for mu in range(4):
        check(_l.lib().lqcd_link_mul_adj(nn._bare._h, mu, U._h, mu, nn._force._h, mu))
    return nn._bare


class GaugeAction:
    """GaugeAction(U) + push!(gauge_action, beta/2, plaqloop) (universe.jl:88-96): the plaquette action of the reference's runs.
    Owns the temporary link fields P_update! / U_update! borrow (get_temporary_gaugefields, AbstractMD.jl:79,101,122)."""

    def __init__(self, U):
        self.lattice = U.lattice
        self.beta_inp = 0.0
        self._temps = Temporalfields(U.lattice)

    def push_(self, beta_inp, loops):
        if loops != make_loops_fromname("plaquette") + make_loops_fromname("plaquette", adjoint=True):
            raise LQCDError(_l.ERR_UNSUPPORTED, "only the plaquette loop and its adjoint (universe.jl:92-95) are supported on the HIP path")
        self.beta_inp += float(beta_inp)
        return self

    @property
    def beta(self):            # beta/2 on the loop and on its adjoint
        return 2.0 * self.beta_inp


def make_loops_fromname(name, Dim=4, adjoint=False):
    """make_loops_fromname("plaquette", Dim=Dim) (universe.jl:92); append!(plaqloop, plaqloop') adds the adjoint loops."""
    if name != "plaquette":
        raise LQCDError(_l.ERR_UNSUPPORTED, f"loop {name} is not supported")
    return [("plaquette-adjoint" if adjoint else "plaquette", mu, nu) for mu in range(1, Dim + 1) for nu in range(mu + 1, Dim + 1)]


class Temporalfields:
    """Gaugefields.Temporalfields_module: a pool of single-direction link fields, get_temp / unused! (AbstractMD.jl:80-97)."""

    def __init__(self, lattice):
        self.lattice = lattice
        self._fields, self._free, self._used = [], [], set()

    def _grow(self):
        f = Gaugefields(self.lattice)
        self._fields.append(f)
        base = 4 * (len(self._fields) - 1)
        self._free.extend(range(base, base + 4))

    def _view(self, it):
        return LinkView(self._fields[it // 4], it % 4)


def get_temporary_gaugefields(gauge_action):
    return gauge_action._temps


def get_temp(temps, n=None):
    """get_temp(temps) -> (field, index); get_temp(temps, Dim) -> (fields, indices) (AbstractMD.jl:80-83,123)."""
    def one():
        if not temps._free:
            temps._grow()
        it = temps._free.pop(0)
        temps._used.add(it)
        return temps._view(it), it
    if n is None:
        return one()
    pairs = [one() for _ in range(n)]
    return [v for v, _ in pairs], [i for _, i in pairs]


def unused_(temps, it):
    for i in (it if isinstance(it, (list, tuple)) else [it]):
        temps._used.discard(i)
        temps._free.append(i)
    temps._free.sort()


def initialize_TA_Gaugefields(U):
    """initialize_TA_Gaugefields(U) (standardMD.jl:34): the traceless anti-Hermitian momenta p[mu], one device object."""
    return Gaugefields(U.lattice)


def exptU_(expU, t, p_mu, temps=None):
    """exptU!(expU, t, p[mu], temps) (AbstractMD.jl:91): expU = exp(t p[mu]) site by site (first call of the U_update! triple: recorded by the library)."""
    check(_l.lib().lqcd_link_exp(expU.field._h, expU.slot, C.c_double(t), p_mu.field._h, p_mu.slot))
    return expU


def calc_dSdUmu_(dSdUmu, gauge_action, mu, U):
    """calc_dSdUμ!(dSdUμ, gauge_action, μ, U) (AbstractMD.jl:108): beta_inp * (sum of the staples of U[μ]), μ = 1..4 (first call of the P_update!
    triple: recorded by the library)."""
    check(_l.lib().lqcd_link_staple(dSdUmu.field._h, dSdUmu.slot, U._h, int(mu) - 1, C.c_double(gauge_action.beta)))
    return dSdUmu


def evaluate_GaugeAction(a, b):
    """Reference form evaluate_GaugeAction(gauge_action, U) (standardHMC.jl:50; S_g = -that/NC), and the direct
    evaluate_GaugeAction(U, beta) = S_g = -(beta/3) sum_plaq Re tr U_p."""
    s = C.c_double(0)
    if isinstance(a, GaugeAction):
        check(_l.lib().lqcd_gauge_action(b._h, C.c_double(a.beta), C.byref(s)))
        return -3.0 * s.value
    check(_l.lib().lqcd_gauge_action(a._h, C.c_double(b), C.byref(s)))
    return s.value


def gauge_force_(G, U, beta):
    """calc_dSdUmu! followed by mul!(temp, U, dSdUmu) (AbstractMD.jl:108-109): G = -(beta/6) U * staples."""
    check(_l.lib().lqcd_gauge_force(G._h, U._h, C.c_double(beta)))
    return G


def Traceless_antihermitian_add_(p, factor, G):
    """Traceless_antihermitian_add!(p[mu], factor, temp) (AbstractMD.jl:110,131) on one direction (third call of the P_update! triple: the library
    runs calc_dSdUmu! -> mul! -> Traceless_antihermitian_add! as one pass); on whole fields all four at once."""
    if isinstance(p, LinkView):
        check(_l.lib().lqcd_link_add_ta(p.field._h, p.slot, C.c_double(factor), G.field._h, G.slot))
        return p
    check(_l.lib().lqcd_momentum_add_ta(p._h, C.c_double(factor), G._h))
    return p


def P_update_(U, p, factor, beta):
    """P_update!(U, p, eps, md) (AbstractMD.jl:99-118) in one pass: p += factor * TA(-(beta/6) U staples)."""
    check(_l.lib().lqcd_momentum_add_gauge_force(p._h, C.c_double(factor), U._h, C.c_double(beta)))
    return p


def U_update_(U, p, dt):
    """U_update!(U, p, eps, md) (AbstractMD.jl:78-97): U <- exp(dt p) U."""
    check(_l.lib().lqcd_gauge_exp_update(U._h, C.c_double(dt), p._h))
    return U


def gauss_distribution_(p, randomseed=114):
    """gauss_distribution!(md.p) (standardMD.jl:86)."""
    check(_l.lib().lqcd_momentum_gaussian(p._h, C.c_uint64(randomseed)))
    return p


def momentum_action(p):
    """md.p * md.p / 2 (standardHMC.jl:49)."""
    k = C.c_double(0)
    check(_l.lib().lqcd_momentum_action(p._h, C.byref(k)))
    return k.value


# ------------------------------------------------------------------------------------ timing helpers (bench.py)
def bench_dslash(D, out, inp, warm=20, reps=200):
    ms = C.c_double(0)
    check(_l.lib().lqcd_bench_dslash(D._h, out._h, inp._h, int(D.dagger), int(warm), int(reps), C.byref(ms)))
    return ms.value


def bench_dslash_median(D, out, inp, warm=20, reps=200):
    """every application between its own HIP events -> (median ms, mean ms)."""
    med, mean = C.c_double(0), C.c_double(0)
    check(_l.lib().lqcd_bench_dslash_median(D._h, out._h, inp._h, int(D.dagger), int(warm), int(reps), C.byref(med), C.byref(mean)))
    return med.value, mean.value


def bench_cg(D, x, b, warm=5, niter=50):
    ms = C.c_double(0)
    check(_l.lib().lqcd_bench_cg(D._h, x._h, b._h, int(warm), int(niter), C.byref(ms)))
    return ms.value


class CGSession:
    """Externally timed CG window: begin (r = b - D'D x, p = r), then iterate(n) enqueues n iterations with the exit test
    disabled and waits for them."""

    def __init__(self, D, x, b):
        self.D = D
        check(_l.lib().lqcd_cg_session_begin(D._h, x._h, b._h))

    def iterate(self, n):
        check(_l.lib().lqcd_cg_session_iterate(self.D._h, int(n)))

    def close(self):
        check(_l.lib().lqcd_cg_session_end(self.D._h))


# ------------------------------------------------------------------------------------ in-process PE-grid emulation (tests)
def _harr(objs):
    return (C.c_void_p * len(objs))(*[o._h for o in objs])


def mdom_mul_(ys, Ds, xs):
    check(_l.lib().lqcd_mdom_op_apply(len(Ds), _harr(Ds), _harr(ys), _harr(xs), int(Ds[0].dagger)))


def mdom_fermion_force_(Gs, Ds, Xs, Ys):
    check(_l.lib().lqcd_mdom_fermion_force(len(Ds), _harr(Ds), _harr(Gs), _harr(Xs), _harr(Ys)))


def mdom_gauge_force_(Gs, Us, beta):
    check(_l.lib().lqcd_mdom_gauge_force(len(Us), _harr(Gs), _harr(Us), C.c_double(beta), C.c_double(0.0), 0))


def mdom_P_update_(Us, Ps, factor, beta):
    check(_l.lib().lqcd_mdom_gauge_force(len(Us), _harr(Ps), _harr(Us), C.c_double(beta), C.c_double(factor), 1))


def mdom_dot(As, Bs):
    re, im = C.c_double(0), C.c_double(0)
    check(_l.lib().lqcd_mdom_dot(len(As), _harr(As), _harr(Bs), C.byref(re), C.byref(im)))
    return complex(re.value, im.value)


def mdom_plaquette(Us):
    p = C.c_double(0)
    check(_l.lib().lqcd_mdom_plaquette(len(Us), _harr(Us), C.byref(p)))
    return p.value


def mdom_solve_cg(Ds, xs, bs, eps=1e-19, maxiter=3000):
    it, rr = C.c_int(0), C.c_double(0)
    check(_l.lib().lqcd_mdom_solve_cg_DdagD(len(Ds), _harr(Ds), _harr(xs), _harr(bs), C.c_double(eps), int(maxiter),
                                            C.byref(it), C.byref(rr)))
    return it.value, rr.value
