// capi.hip -- context management, PE-grid decomposition, RCCL bootstrap, error reporting, host-only index helpers.
// The PE grid is the reference's PEs concept (/root/reference/src/mpirun.jl:17-19, src/mpi/mpimodule.jl:9-13):
// rank = px + PX*(py + PY*(pz + PZ*pt)); one process (one context) per GPU.
#include "lqcd_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace lqcd {

static thread_local std::string g_err;
// contexts that exist: a gauge-shaped field checks here before it looks at its context's recorded link operations (finalizers run in any order)
static std::mutex g_live_mu;
static std::vector<const lqcd_ctx_s*> g_live;
bool ctx_is_live(const lqcd_ctx_s* c) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (const lqcd_ctx_s* p : g_live)
        if (p == c) return true;
    return false;
}
static void ctx_set_live(const lqcd_ctx_s* c, bool live) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    for (size_t i = 0; i < g_live.size(); i++)
        if (g_live[i] == c) { g_live.erase(g_live.begin() + i); break; }
    if (live) g_live.push_back(c);
}
bool ctx_park_gauge(lqcd_gauge_s* g) {
    std::lock_guard<std::mutex> lk(g_live_mu);      // lqcd_ctx_destroy leaves the live set under the same lock before it drains: nothing is parked on a dying context
    lqcd_ctx_s* c = g->ctx;
    bool live = false;
    for (const lqcd_ctx_s* p : g_live) live = live || p == c;
    if (!live || !c->tun.lazy_links || std::this_thread::get_id() == c->home_thread) return false;
    c->parked_gauges.push_back(g);
    return true;
}
int ctx_drain_parked(lqcd_ctx_s* c) {
    std::vector<lqcd_gauge_s*> mine;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        mine.swap(c->parked_gauges);
    }
    for (lqcd_gauge_s* g : mine) (void)gauge_destroy_now(g);
    return LQCD_OK;
}
void set_error(const std::string& msg) { g_err = msg; }
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    g_err = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
    (void)hipGetLastError();
    return LQCD_ERR_HIP;
}
int nccl_fail(ncclResult_t e, const char* what, const char* file, int line) {
    g_err = std::string("RCCL error: ") + ncclGetErrorString(e) + " in " + what + " (" + file + ":" + std::to_string(line) + ")";
    return LQCD_ERR_COMM;
}

lqcd_spinor_s* scratch_get(lqcd_ctx_s* c, int kind, int subset) {
    if (std::this_thread::get_id() == c->home_thread && !c->parked_gauges.empty()) (void)ctx_drain_parked(c);      // what finalizer threads handed over (a long
                                                                                                                    // run may never create a gauge field or sync again)
    const int want_sub = subset == LQCD_FULL ? LQCD_FULL : LQCD_EVEN;  // half fields are interchangeable
    for (lqcd_spinor_s* s : c->scratch)
        if (!s->in_use && s->kind == kind && (s->subset == LQCD_FULL) == (want_sub == LQCD_FULL)) {
            s->in_use = true;
            s->subset = subset;
            return s;
        }
    lqcd_spinor_s* s = nullptr;
    if (lqcd_spinor_create(c, &s, kind, subset) != LQCD_OK) return nullptr;
    s->in_use = true;
    c->scratch.push_back(s);
    return s;
}
void scratch_put(lqcd_spinor_s* s) {
    if (s) s->in_use = false;
}

int plaquette_local_sum(lqcd_gauge_s* g, const double2* const ghost[4], double* sum);
int gauge_pack_face(lqcd_gauge_s* g, int mu, double2* dst);

static int decompose(const int gL[4], const int pe[4], int rank, int L[4], int origin[4], int coord[4], int nf[4], int nb[4]) {
    int nranks = 1;
    for (int mu = 0; mu < 4; mu++) {
        ARGCHK(pe[mu] >= 1 && gL[mu] >= 2, "decompose: bad PE grid or lattice extent");
        ARGCHK(gL[mu] % pe[mu] == 0, "decompose: global extent not divisible by the PE grid");
        L[mu] = gL[mu] / pe[mu];
        ARGCHK(L[mu] % 2 == 0, "decompose: local extents must be even (checkerboard layout)");
        nranks *= pe[mu];
    }
    ARGCHK(rank >= 0 && rank < nranks, "decompose: rank outside the PE grid");
    int q = rank;
    for (int mu = 0; mu < 4; mu++) { coord[mu] = q % pe[mu]; q /= pe[mu]; origin[mu] = coord[mu] * L[mu]; }
    for (int mu = 0; mu < 4; mu++) {
        int cf[4] = {coord[0], coord[1], coord[2], coord[3]}, cb[4] = {coord[0], coord[1], coord[2], coord[3]};
        cf[mu] = (coord[mu] + 1) % pe[mu];
        cb[mu] = (coord[mu] + pe[mu] - 1) % pe[mu];
        nf[mu] = cf[0] + pe[0] * (cf[1] + pe[1] * (cf[2] + pe[2] * cf[3]));
        nb[mu] = cb[0] + pe[0] * (cb[1] + pe[1] * (cb[2] + pe[2] * cb[3]));
    }
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

extern "C" int lqcd_version(void) { return 100; }
extern "C" const char* lqcd_last_error(void) { return g_err.c_str(); }

extern "C" int lqcd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

extern "C" int lqcd_device_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes) {
    ARGCHK(free_bytes && total_bytes, "lqcd_device_mem_info: null argument");
    HIPCHK(hipSetDevice(device));
    size_t f = 0, t = 0;
    HIPCHK(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return LQCD_OK;
}

extern "C" int64_t lqcd_index_lex(const int L[4], int x, int y, int z, int t) {
    return x + (int64_t)L[0] * (y + (int64_t)L[1] * (z + (int64_t)L[2] * t));
}

static void geom_from_L(Geom& g, const int L[4]) {
    memset(&g, 0, sizeof(g));
    for (int mu = 0; mu < 4; mu++) { g.L[mu] = L[mu]; g.gL[mu] = L[mu]; g.bc_fwd[mu] = g.bc_bwd[mu] = 1.0; }
    g.XH = L[0] / 2;
    g.Vh = (L[0] / 2) * L[1] * L[2] * L[3];
    g.nch = (g.Vh + 63) / 64;
    g.dXH = make_fastdiv(g.XH); g.dL1 = make_fastdiv(L[1]); g.dL2 = make_fastdiv(L[2]);
    // component stride: Vh rounded up to 64 sites (1 KiB, keeps every wave load on 8 whole cache lines) plus
    // LQCD_PAD_CHUNKS * 64 sites (default 1) so that consecutive component arrays are NOT a power-of-two apart
    int pad_chunks = 1;
    if (const char* e = getenv("LQCD_PAD_CHUNKS")) pad_chunks = atoi(e);
    if (pad_chunks < 0) pad_chunks = 0;
    g.Vs = ((g.Vh + 63) / 64) * 64 + 64 * pad_chunks;
}

extern "C" int lqcd_index_cb(const int L[4], int x, int y, int z, int t, int* parity, int64_t* cb) {
    ARGCHK(L && parity && cb, "lqcd_index_cb: null");
    ARGCHK(L[0] % 2 == 0, "lqcd_index_cb: NX must be even");
    ARGCHK(x >= 0 && x < L[0] && y >= 0 && y < L[1] && z >= 0 && z < L[2] && t >= 0 && t < L[3], "lqcd_index_cb: site out of range");
    Geom g;
    geom_from_L(g, L);
    const int c[4] = {x, y, z, t};
    *parity = (x + y + z + t) & 1;
    *cb = coords_to_cb(g, c);
    return LQCD_OK;
}

extern "C" int lqcd_coords_cb(const int L[4], int parity, int64_t cb, int xyzt[4]) {
    ARGCHK(L && xyzt, "lqcd_coords_cb: null");
    ARGCHK(L[0] % 2 == 0, "lqcd_coords_cb: NX must be even");
    Geom g;
    geom_from_L(g, L);
    ARGCHK((parity == 0 || parity == 1) && cb >= 0 && cb < g.Vh, "lqcd_coords_cb: index out of range");
    cb_to_coords(g, parity, (int)cb, xyzt);
    return LQCD_OK;
}

extern "C" int lqcd_decompose(const int gL[4], const int pe[4], int rank, int L[4], int origin[4], int rank_fwd[4], int rank_bwd[4]) {
    ARGCHK(gL && pe && L && origin && rank_fwd && rank_bwd, "lqcd_decompose: null");
    int coord[4];
    return decompose(gL, pe, rank, L, origin, coord, rank_fwd, rank_bwd);
}

// ---------------------------------------------------------------------------------- context
extern "C" int lqcd_ctx_create(lqcd_ctx_t* out, int device, const int gL[4], const int pe[4], int rank) {
    ARGCHK(out && gL && pe, "lqcd_ctx_create: null argument");
    int L[4], origin[4], coord[4], nf[4], nb[4];
    LQCHK(decompose(gL, pe, rank, L, origin, coord, nf, nb));
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0 || device < 0 || device >= ndev) {
        set_error("lqcd_ctx_create: no such HIP device " + std::to_string(device) + " (" + std::to_string(ndev) + " visible) -- "
                  "this library has no CPU fallback");
        return LQCD_ERR_HIP;
    }
    HIPCHK(hipSetDevice(device));
    const long long V = (long long)L[0] * L[1] * L[2] * L[3];
    ARGCHK(V / 2 < (1ll << 31) / 16, "lqcd_ctx_create: local volume too large for 32-bit site indices");
    lqcd_ctx_s* c = new lqcd_ctx_s;
    c->home_thread = std::this_thread::get_id();
    c->device = device;
    c->rank = rank;
    c->nranks = pe[0] * pe[1] * pe[2] * pe[3];
    geom_from_L(c->geom, L);
    for (int mu = 0; mu < 4; mu++) {
        c->gL[mu] = gL[mu]; c->pe[mu] = pe[mu]; c->coord[mu] = coord[mu];
        c->nbr_fwd[mu] = nf[mu]; c->nbr_bwd[mu] = nb[mu];
        c->geom.gL[mu] = gL[mu]; c->geom.origin[mu] = origin[mu];
        c->geom.part[mu] = pe[mu] > 1 ? 1 : 0;
    }
    // testing aid: LQCD_FORCE_PARTITION=<bitmask> treats direction mu as partitioned even when pe[mu] == 1 (the neighbour
    // rank is then this rank itself), which drives the whole pack / RCCL send-recv / interior-exterior machinery on ONE GPU
    if (const char* e = getenv("LQCD_FORCE_PARTITION")) {
        const int mask = atoi(e);
        for (int mu = 0; mu < 4; mu++)
            if ((mask >> mu) & 1) c->geom.part[mu] = 1;
    }
    // Volume-adaptive defaults (round 5, profiles/r05_small_volume_hints.log; every one of them can be set afterwards).  The map and the cache hints were tuned at
    // 32^3 x 64, where links and spinors stream from HBM.  At the local volumes of a partitioned run the picture changes: (i) a t-slice of at most 128 chunks per
    // parity splits into 8 sub-domains (one per XCD, tiled 2 x in y) instead of 16 -- with 16 every XCD sweeps t twice over sub-domains of 4-8 chunks and loses the
    // t-neighbour locality; (ii) 12-real links of at most 256 MB stay in the Infinity Cache between applications unless their last use streams them through (nt_gauge
    // bit 0); (iii) an output spinor of at most 64 MB is read again (by D^+, by the update) before it would leave the cache: plain stores.  N = 8 local volume of
    // 32^3 x 64: 187.7 -> 181.9 us per CG iteration; N = 4: 325.1 -> 319.5; N = 2 and N = 1: unchanged (the defaults stay).
    {
        const Geom& g = c->geom;
        const long slice = (long)g.XH * g.L[1] * g.L[2];
        if (slice % 64 == 0 && slice / 64 <= 128) { c->tun.xcd_nsub = 8; c->tun.xcd_ysplit = 2; }
        if (gauge12_elems(g) * sizeof(double2) <= ((size_t)256 << 20)) c->tun.nt_gauge = 0;
        if ((size_t)24 * g.Vh * sizeof(double2) <= ((size_t)64 << 20)) c->tun.nt_store = 0;
    }
    // testing aids of the same kind: the halo schedule (0..3, -1 = the collective one-off timing) and the folded form of schedule 3, so that the self-partition tests
    // can pin the schedule they mean to cover without touching their drivers
    if (const char* e = getenv("LQCD_HALO_STREAM_MODE")) {
        const int v = atoi(e);
        if (v < -1 || v > 4) { set_error(std::string("LQCD_HALO_STREAM_MODE=") + e + ": the halo schedule is -1 (timed once) or 0..4"); delete c; return LQCD_ERR_ARG; }
        c->tun.halo_stream_mode = v;
    }
    if (const char* e = getenv("LQCD_HALO_FOLD")) {
        const int v = atoi(e);
        if (v != 0 && v != 1) { set_error(std::string("LQCD_HALO_FOLD=") + e + ": 0 or 1"); delete c; return LQCD_ERR_ARG; }
        c->tun.halo_fold = v;
    }
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
    HIPCHK(hipEventCreate(&c->ev_t0));
    HIPCHK(hipEventCreate(&c->ev_t1));
    HIPCHK(hipEventCreate(&c->ev_tune0));
    HIPCHK(hipEventCreate(&c->ev_tune1));
    const size_t npart = (size_t)2 * c->geom.Vh / 64 + 4096 + 8 * ((size_t)2 * c->geom.Vh / 128 + 8);  // interior + exterior partials
    HIPCHK(hipMalloc((void**)&c->d_partial, npart * 2 * sizeof(double)));
    HIPCHK(hipMalloc((void**)&c->d_scal, SCAL_DOUBLES * sizeof(double)));
    HIPCHK(hipMemset(c->d_scal, 0, SCAL_DOUBLES * sizeof(double)));
    HIPCHK(hipMalloc((void**)&c->pipe_ctr, PIPE_CTR_WORDS * sizeof(unsigned)));      // work-queue heads of the persistent stencil kernel: zero between launches
    HIPCHK(hipMemset(c->pipe_ctr, 0, PIPE_CTR_WORDS * sizeof(unsigned)));
    HIPCHK(hipMalloc((void**)&c->cgp_ctr, 9 * 32 * sizeof(unsigned)));
    HIPCHK(hipMemset(c->cgp_ctr, 0, 9 * 32 * sizeof(unsigned)));
    HIPCHK(hipHostMalloc((void**)&c->h_scal, SCAL_DOUBLES * sizeof(double), hipHostMallocDefault));
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        c->halo_elems[mu] = (size_t)2 * 6 * face_half_sites(c->geom, mu);
        const size_t bytes = c->halo_elems[mu] * sizeof(double2);
        // [send_fwd | send_bwd] and [recv_bwd | recv_fwd] are adjacent: when both neighbours of a direction are the same rank
        // (PE extent 2) a full-size exchange is ONE message each way instead of two (ops.hip halo_exchange_rccl)
        HIPCHK(hipMalloc((void**)&c->send_fwd[mu], 2 * bytes));
        c->send_bwd[mu] = c->send_fwd[mu] + c->halo_elems[mu];
        HIPCHK(hipMalloc((void**)&c->recv_bwd[mu], 2 * bytes));
        c->recv_fwd[mu] = c->recv_bwd[mu] + c->halo_elems[mu];
    }
    ctx_set_live(c, true);
    *out = c;
    return LQCD_OK;
}

extern "C" int lqcd_ctx_destroy(lqcd_ctx_t c) {
    if (!c) return LQCD_OK;
    ctx_set_live(c, false);      // recorded link operations are dropped with the context: their fields cannot be used without it
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)ctx_drain_parked(c);      // fields that finalizer threads handed over (the context is no longer live: freed without a flush)
    for (lqcd_spinor_s* s : c->scratch) { (void)hipFree(s->data); delete s; }
    for (int mu = 0; mu < 4; mu++) {
        (void)hipFree(c->send_fwd[mu]); (void)hipFree(c->recv_bwd[mu]);   // send_bwd / recv_fwd are the second halves of these
        (void)hipFree(c->force_send[mu]); (void)hipFree(c->force_recv[mu]);
        (void)hipFree(c->gf_ghost[mu]); (void)hipFree(c->gf_gsend[mu]); (void)hipFree(c->gf_wsend[mu]); (void)hipFree(c->gf_wrecv[mu]);
    }
    for (void* b : c->mix_buf) (void)hipFree(b);
    (void)hipFree(c->clover_q[0]); (void)hipFree(c->clover_q[1]);
    for (lqcd_gauge_s*& t : c->stout_tmp) if (t) { (void)hipFree(t->data); (void)hipFree(t->data12); (void)hipFree(t->data12d); delete t; t = nullptr; }
    (void)hipFree(c->gauge_spare);
    (void)hipFree(c->clover_ext); (void)hipFree(c->clover_ext_buf[0]); (void)hipFree(c->clover_ext_buf[1]);
    if (c->has_comm && !c->peer.on) { ncclCommDestroy(c->comm); ncclCommDestroy(c->comm_red); }
    comm_teardown(c);      // the peer-mapped backend's windows (comm.hip)
    for (lqcd::FoldLists& f : c->fold_lists) { (void)hipFree(f.d_list[0]); (void)hipFree(f.d_list[1]); }
    delete static_cast<lqcd::StencilCall*>(c->waiting_pack);
    (void)hipFree(c->d_partial); (void)hipFree(c->d_scal); (void)hipFree(c->pipe_ctr); (void)hipFree(c->cgp_ctr); (void)hipHostFree(c->h_scal);
    (void)hipEventDestroy(c->ev_pack); (void)hipEventDestroy(c->ev_comm); (void)hipEventDestroy(c->ev_t0); (void)hipEventDestroy(c->ev_t1);
    (void)hipEventDestroy(c->ev_tune0); (void)hipEventDestroy(c->ev_tune1);
    (void)hipStreamDestroy(c->stream); (void)hipStreamDestroy(c->comm_stream);
    delete c;
    return LQCD_OK;
}

extern "C" int lqcd_ctx_sync(lqcd_ctx_t c) {
    ARGCHK(c, "lqcd_ctx_sync: null");
    LQCHK(ctx_drain_parked(c));
    LQCHK(links_flush_of(c));
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->comm_stream));
    return comm_check(c);      // peer-mapped backend: a wait that gave up (dead rank) surfaces here
}

static int* param_ptr(lqcd_ctx_s* c, const char* key) {
    if (!strcmp(key, "dslash_block")) return &c->tun.dslash_block;
    if (!strcmp(key, "xcd_remap")) return &c->tun.xcd_remap;
    if (!strcmp(key, "dslash_variant")) return &c->tun.dslash_variant;
    if (!strcmp(key, "nt_gauge")) return &c->tun.nt_gauge;
    if (!strcmp(key, "nt_store")) return &c->tun.nt_store;
    if (!strcmp(key, "cg_fused")) return &c->tun.cg_fused;
    if (!strcmp(key, "graph")) return &c->tun.graph;
    if (!strcmp(key, "gauge_recon")) return &c->tun.gauge_recon;
    if (!strcmp(key, "mixed_action_solver")) return &c->tun.mixed_action_solver;
    if (!strcmp(key, "mixed_pair32")) return &c->tun.mixed_pair32;
    if (!strcmp(key, "mixed_links16")) return &c->tun.mixed_links16;
    if (!strcmp(key, "bicg_reliable")) return &c->tun.bicg_reliable;
    if (!strcmp(key, "bicg_rec_guard")) return &c->tun.bicg_rec_guard;
    if (!strcmp(key, "bicg_dot_soa")) return &c->tun.bicg_dot_soa;
    if (!strcmp(key, "clover_hop_s")) return &c->tun.clover_hop_s;
    if (!strcmp(key, "mixed_lean_residual")) return &c->tun.mixed_lean_residual;
    if (!strcmp(key, "mixed_xfuse")) return &c->tun.mixed_xfuse;
    if (!strcmp(key, "mixed_defer_x")) return &c->tun.mixed_defer_x;
    if (!strcmp(key, "pair32_active")) return &c->tun.pair32_active;
    if (!strcmp(key, "stag_both")) return &c->tun.stag_both;
    if (!strcmp(key, "clover_fused")) return &c->tun.clover_fused;
    if (!strcmp(key, "clover_transport")) return &c->tun.clover_transport;
    if (!strcmp(key, "halo_stream_mode")) return &c->tun.halo_stream_mode;
    if (!strcmp(key, "cg_skip_done")) return &c->tun.cg_skip_done;
    if (!strcmp(key, "cg_small")) return &c->tun.cg_small;
    if (!strcmp(key, "cg_persist")) return &c->tun.cg_persist;
    if (!strcmp(key, "md_remap")) return &c->tun.md_remap;
    if (!strcmp(key, "staple_recon")) return &c->tun.staple_recon;
    if (!strcmp(key, "staple_tile")) return &c->tun.staple_tile;
    if (!strcmp(key, "md_reunitarize")) return &c->tun.md_reunitarize;
    if (!strcmp(key, "nt_blas")) return &c->tun.nt_blas;
    if (!strcmp(key, "cg_fold_scalars")) return &c->tun.cg_fold_scalars;
    if (!strcmp(key, "halo_fuse")) return &c->tun.halo_fuse;
    if (!strcmp(key, "halo_fold")) return &c->tun.halo_fold;
    if (!strcmp(key, "halo_fold_active")) return &c->tun.halo_fold_active;
    if (!strcmp(key, "cg_defer_x")) return &c->tun.cg_defer_x;
    if (!strcmp(key, "staggered_parity_solve")) return &c->tun.staggered_parity_solve;
    if (!strcmp(key, "halo_tuned_us0")) return &c->tun.halo_tuned_us[0];
    if (!strcmp(key, "halo_tuned_us1")) return &c->tun.halo_tuned_us[1];
    if (!strcmp(key, "halo_tuned_us2")) return &c->tun.halo_tuned_us[2];
    if (!strcmp(key, "halo_tuned_us3")) return &c->tun.halo_tuned_us[3];
    if (!strcmp(key, "halo_tuned_us4")) return &c->tun.halo_tuned_us[4];
    if (!strcmp(key, "halo_inject_us")) return &c->tun.halo_inject_us;
    if (!strcmp(key, "halo_merge")) return &c->tun.halo_merge;
    if (!strcmp(key, "recon_active")) return &c->tun.recon_active;
    if (!strcmp(key, "variants_built")) return &c->tun.variants_built;
    if (!strcmp(key, "lds_pad_kb")) return &c->tun.lds_pad_kb;
    if (!strcmp(key, "xcd_nsub")) return &c->tun.xcd_nsub;
    if (!strcmp(key, "xcd_ysplit")) return &c->tun.xcd_ysplit;
#ifdef LQCD_ABLATE     /* timing ablations (wrong results): only in -DLQCD_ABLATE builds of the library, never in the shipped one */
    if (!strcmp(key, "dbg")) return &c->tun.dbg;
#endif
    if (!strcmp(key, "persist_per_cu")) return &c->tun.persist_per_cu;
    if (!strcmp(key, "dslash_pipe")) return &c->tun.dslash_pipe;
    if (!strcmp(key, "pipe_per_cu")) return &c->tun.pipe_per_cu;
    if (!strcmp(key, "pipe_grid")) return &c->tun.pipe_grid;
    if (!strcmp(key, "pipe_chunks_per_wg")) return &c->tun.pipe_chunks_per_wg;
    if (!strcmp(key, "pipe_min_chunks")) return &c->tun.pipe_min_chunks;
    if (!strcmp(key, "lazy_links")) return &c->tun.lazy_links;
    if (!strcmp(key, "lazy_merge")) return &c->tun.lazy_merge;
    if (!strcmp(key, "dw_batched")) return &c->tun.dw_batched;
    if (!strcmp(key, "dw_fused_cg")) return &c->tun.dw_fused_cg;
    if (!strcmp(key, "bicg_fused")) return &c->tun.bicg_fused;
    if (!strcmp(key, "gauge_delta")) return &c->tun.gauge_delta;
    if (!strcmp(key, "dslash_s18")) return &c->tun.dslash_s18;
    if (!strcmp(key, "bicg_mixed")) return &c->tun.bicg_mixed;
    if (!strcmp(key, "action_eo_solver")) return &c->tun.action_eo_solver;
    if (!strcmp(key, "bicg_xrp_active")) return &c->tun.bicg_xrp_active;
    if (!strcmp(key, "peer_timeout_ms")) return &c->peer.timeout_ms;
    return nullptr;
}
extern "C" int lqcd_ctx_set_param(lqcd_ctx_t c, const char* key, int value) {
    ARGCHK(c && key, "lqcd_ctx_set_param: null");
    if (!strcmp(key, "adopt_thread")) {      // the calling thread is the context's own from here on (a host that hands a context to a worker thread for good)
        {
            std::lock_guard<std::mutex> lk(g_live_mu);
            c->home_thread = std::this_thread::get_id();
        }
        return ctx_drain_parked(c);
    }
    int* p = param_ptr(c, key);
    ARGCHK(p, std::string("lqcd_ctx_set_param: unknown key ") + key);
    if (!strcmp(key, "dslash_block")) ARGCHK(value == 64 || value == 128 || value == 256, "dslash_block must be 64, 128 or 256");
    if (!strcmp(key, "halo_stream_mode")) ARGCHK(value >= -1 && value <= 4, "halo_stream_mode must be -1 (timed once) or 0..4");
    if (!strcmp(key, "halo_inject_us")) ARGCHK(value >= 0 && value <= 100000, "halo_inject_us: 0..100000");
    if ((!strcmp(key, "lazy_links") || !strcmp(key, "lazy_merge")) && !value) LQCHK(links_flush_of(c));      // switching to eager calls: what is recorded runs now
    *p = value;
    return LQCD_OK;
}
extern "C" int lqcd_ctx_get_param(lqcd_ctx_t c, const char* key, int* value) {
    ARGCHK(c && key && value, "lqcd_ctx_get_param: null");
    // read-only views of the recorded link operations (md.hip): the open triple (0 none, 1 exp, 2 exp + mul, 3 staple, 4 staple + mul), deferred triples
    if (!strcmp(key, "dw_active")) { *value = c->tun.dw_active; return LQCD_OK; }
    if (!strcmp(key, "adopt_thread")) {      // 1: the calling thread is the context's own (the creating one, or the last to set adopt_thread)
        std::lock_guard<std::mutex> lk(g_live_mu);
        *value = std::this_thread::get_id() == c->home_thread ? 1 : 0;
        return LQCD_OK;
    }
    if (!strcmp(key, "parked_fields")) {      // gauge-shaped fields that another thread's destroy call left for this context's thread to free
        std::lock_guard<std::mutex> lk(g_live_mu);
        *value = (int)c->parked_gauges.size();
        return LQCD_OK;
    }
    if (!strcmp(key, "lazy_open")) { *value = c->lazy.kind; return LQCD_OK; }
    if (!strcmp(key, "lazy_deferred")) { *value = (int)c->lazy.done.size() + (c->lazy.has_pend ? 4 : 0) + (c->lazy.has_pp ? 4 : 0); return LQCD_OK; }
    int* p = param_ptr(c, key);
    ARGCHK(p, std::string("lqcd_ctx_get_param: unknown key ") + key);
    *value = *p;
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- RCCL bootstrap
// Two communicators per context: halo send/recv run on the communication stream, scalar all-reduces on the compute
// stream; giving each stream its own communicator keeps RCCL's one-communicator-one-stream-at-a-time rule trivially true.
extern "C" int lqcd_comm_unique_id(unsigned char id[256]) {
    ARGCHK(id, "lqcd_comm_unique_id: null");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");
    for (int k = 0; k < 2; k++) {
        ncclUniqueId u;
        NCCLCHK(ncclGetUniqueId(&u));
        memcpy(id + 128 * k, &u, 128);
    }
    return LQCD_OK;
}

extern "C" int lqcd_ctx_comm_init(lqcd_ctx_t c, const unsigned char id[256], int nranks) {
    ARGCHK(c && id, "lqcd_ctx_comm_init: null");
    ARGCHK(nranks == c->nranks, "lqcd_ctx_comm_init: nranks does not match the PE grid");
    ARGCHK(!c->has_comm, "lqcd_ctx_comm_init: communicator already initialised");
    ARGCHK(!c->peer.exported, "lqcd_ctx_comm_init: this context has exported a peer-mapped window (lqcd_ctx_peer_export): one backend per context");
    HIPCHK(hipSetDevice(c->device));
    // The first thing a multi-GPU run does with the fabric.  A failure here must say WHO failed and WHAT RCCL said: the message goes to
    // lqcd_last_error() and, because a job with a dead rank usually never gets to print it, to stderr as well.
    auto fail = [&](const char* which, ncclResult_t e) {
        char pci[32] = "?";
        (void)hipDeviceGetPCIBusId(pci, sizeof pci, c->device);
        const char* detail = ncclGetLastError(nullptr);
        std::string msg = std::string("lqcd_ctx_comm_init: ncclCommInitRank (") + which + " communicator) failed on rank " + std::to_string(c->rank) + " of " +
                          std::to_string(nranks) + ", PE grid " + std::to_string(c->pe[0]) + "x" + std::to_string(c->pe[1]) + "x" + std::to_string(c->pe[2]) + "x" +
                          std::to_string(c->pe[3]) + ", HIP device " + std::to_string(c->device) + " (" + pci + "): " + ncclGetErrorString(e) +
                          (detail && *detail ? std::string(" -- ") + detail : std::string("")) +
                          "; check that every rank got the SAME 256-byte id from rank 0, one rank per GPU, and HSA_ENABLE_IPC_MODE_LEGACY=0";
        fprintf(stderr, "%s\n", msg.c_str());
        fflush(stderr);
        set_error(msg);
        return LQCD_ERR_COMM;
    };
    ncclUniqueId u;
    memcpy(&u, id, 128);
    ncclResult_t e = ncclCommInitRank(&c->comm, nranks, u, c->rank);
    if (e != ncclSuccess) return fail("halo", e);
    memcpy(&u, id + 128, 128);
    e = ncclCommInitRank(&c->comm_red, nranks, u, c->rank);
    if (e != ncclSuccess) { ncclCommDestroy(c->comm); c->comm = nullptr; return fail("reduction", e); }
    c->has_comm = true;
    return LQCD_OK;
}

extern "C" int lqcd_ctx_link_local(lqcd_ctx_t* ctxs, int n) {
    ARGCHK(ctxs && n >= 1, "lqcd_ctx_link_local: null");
    for (int r = 0; r < n; r++) {
        ARGCHK(ctxs[r] && ctxs[r]->rank == r && ctxs[r]->nranks == n, "lqcd_ctx_link_local: contexts must be ranks 0..n-1 of one PE grid");
        ARGCHK(ctxs[r]->device == ctxs[0]->device, "lqcd_ctx_link_local: all contexts must live on one device");
        for (int mu = 0; mu < 4; mu++) ARGCHK(ctxs[r]->pe[mu] == ctxs[0]->pe[mu] && ctxs[r]->gL[mu] == ctxs[0]->gL[mu], "lqcd_ctx_link_local: PE grid mismatch");
    }
    for (int r = 0; r < n; r++) ctxs[r]->local_peers.assign(ctxs, ctxs + n);
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- plaquette (single rank or RCCL ranks)
extern "C" int lqcd_gauge_plaquette(lqcd_gauge_t g, double* plaq) {
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(g && plaq, "lqcd_gauge_plaquette: null");
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    ARGCHK(c->local_peers.empty(), "lqcd_gauge_plaquette: context is part of an in-process PE grid, use lqcd_mdom_plaquette");
    double2* ghost[4] = {nullptr, nullptr, nullptr, nullptr};
    double2* sendb[4] = {nullptr, nullptr, nullptr, nullptr};
    int st = LQCD_OK;
    bool part = false;
    for (int mu = 0; mu < 4; mu++) part = part || c->geom.part[mu];
    if (part) {
        ARGCHK(c->has_comm, "lqcd_gauge_plaquette: communicator not initialised");
        for (int mu = 0; mu < 4 && st == LQCD_OK; mu++) {
            if (!c->geom.part[mu]) continue;
            const size_t bytes = (size_t)2 * 4 * 9 * face_half_sites(c->geom, mu) * sizeof(double2);
            if (hipMalloc((void**)&ghost[mu], bytes) != hipSuccess || hipMalloc((void**)&sendb[mu], bytes) != hipSuccess) { st = LQCD_ERR_HIP; break; }
            st = gauge_pack_face(g, mu, sendb[mu]);
        }
        if (st == LQCD_OK) {
            CommXfer x[4];
            int n = 0;
            for (int mu = 0; mu < 4; mu++) {
                if (!c->geom.part[mu]) continue;
                x[n++] = CommXfer{sendb[mu], ghost[mu], (size_t)2 * 4 * 9 * face_half_sites(c->geom, mu) * sizeof(double2), mu, 1};      // lower faces travel backward
            }
            st = comm_sendrecv(c, x, n, c->stream, false);
        }
    }
    double sum = 0;
    if (st == LQCD_OK) st = plaquette_local_sum(g, ghost, &sum);
    for (int mu = 0; mu < 4; mu++) { if (ghost[mu]) (void)hipFree(ghost[mu]); if (sendb[mu]) (void)hipFree(sendb[mu]); }
    if (st != LQCD_OK) return st;
    LQCHK(allreduce_host(c, &sum, 1));
    const double V = (double)c->gL[0] * c->gL[1] * c->gL[2] * c->gL[3];
    *plaq = sum / (6.0 * V * 3.0);
    return LQCD_OK;
}
