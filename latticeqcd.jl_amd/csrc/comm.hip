// comm.hip -- the communication layer of a partitioned lattice: ONE set of calls (face exchange of the stencil, generic face send/receive, sum of device
// scalars over the ranks) over TWO backends.
//
//  * RCCL over xGMI (lqcd_comm_unique_id / lqcd_ctx_comm_init): grouped ncclSend / ncclRecv on the communication stream, ncclAllReduce on the compute stream.
//  * Peer-mapped windows (lqcd_ctx_peer_export / lqcd_ctx_peer_init, round 6; SURVEY.md 8(e) "or peer-mapped writes"): every rank exports one device allocation
//    (hipIpcGetMemHandle) that holds its ghost buffers, mailboxes, flag words and reduction slots, and maps the others'.  The pack kernels (and every producer
//    that packs faces in its epilogue) store straight into the NEIGHBOUR'S ghost buffer; the exchange step that remains is a one-wave kernel per application
//    that raises the neighbours' flag words to the exchange number and waits for its own (peer_halo_sync).  Scalar reductions: each rank stores its partial
//    into its slot in every rank's window, the consumer adds the slots in rank order (peer_allreduce_wave, inside the reduction launches themselves).
//    No RCCL kernel, no dispatch gap behind it, and -- unlike RCCL, which refuses two ranks on one device -- testable with two processes on ONE GPU.
//
// The reference's counterpart is the PE grid of its MPI build: /root/reference/src/mpirun.jl:17-19 (PEs), src/mpi/mpimodule.jl:4-13 (rank -> grid coordinates);
// the wing exchange and MPI.Allreduce live in the un-vendored Gaugefields.jl / LatticeDiracOperators.jl MPI field types.
#include "lqcd_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

namespace lqcd {

// ---------------------------------------------------------------------------------- window layout
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Mailbox capacity per side of direction mu: the largest face message any entry point sends there -- link faces of the staple force and the plaquette
// (72 fh elements), X and Y of the fermion force (48 fh), one layer of the halo-extended block of the clover / stout force (90 x the extended face).
static size_t peer_aux_cap(const Geom& g, int mu) {
    const size_t fh = (size_t)face_half_sites(g, mu);
    size_t ext = 1;
    for (int d = 0; d < 4; d++) if (d != mu) ext *= (size_t)(g.L[d] + 2);
    const size_t elems = std::max<size_t>(std::max<size_t>(72 * fh, 48 * fh), 90 * ext);
    return align_up(elems * sizeof(double2), 256);
}
static PeerLayout peer_layout(const Geom& g) {
    PeerLayout L;
    size_t off = 0;
    L.halo_flag = off; off += 8 * PEER_FLAG_STRIDE;
    L.aux_flag = off; off += 8 * PEER_FLAG_STRIDE;
    L.aux_ack = off; off += 8 * PEER_FLAG_STRIDE;
    L.red_flag = off; off += align_up((size_t)PEER_RED_RING * PEER_MAX_RANKS * sizeof(unsigned long long), 256);
    L.red_val = off; off += align_up((size_t)PEER_RED_RING * PEER_MAX_RANKS * PEER_RED_VALS * sizeof(double), 256);
    off = align_up(off, 4096);
    for (int mu = 0; mu < 4; mu++) {
        if (!g.part[mu]) continue;
        L.ghost_bytes[mu] = align_up((size_t)2 * 2 * 6 * face_half_sites(g, mu) * sizeof(double2), 4096);      // [from -mu | from +mu], Wilson full-lattice message
        L.ghost[mu] = off; off += 2 * L.ghost_bytes[mu];
    }
    for (int mu = 0; mu < 4; mu++) {
        if (!g.part[mu]) continue;
        L.aux_cap[mu] = peer_aux_cap(g, mu);
        L.aux_box[mu] = off; off += 2 * L.aux_cap[mu];
    }
    L.total = align_up(off, 4096);
    return L;
}
static uint64_t peer_layout_hash(const lqcd_ctx_s* c) {
    uint64_t h = 0x6c716364u;
    for (int mu = 0; mu < 4; mu++) {
        h = splitmix64(h ^ (uint64_t)c->geom.L[mu]); h = splitmix64(h ^ (uint64_t)c->pe[mu]); h = splitmix64(h ^ (uint64_t)c->geom.part[mu]);
    }
    return splitmix64(h ^ (uint64_t)c->peer.lay.total);
}

static unsigned long long* flag_at(char* win, size_t base, int idx) { return (unsigned long long*)(win + base + (size_t)idx * PEER_FLAG_STRIDE); }
static unsigned long long peer_limit(const lqcd_ctx_s* c) { return (unsigned long long)std::max(1, c->peer.timeout_ms) * 100000ull; }      // 100 MHz ticks

// what a rank publishes about its window (lqcd_ctx_peer_export): LQCD_PEER_BLOB_BYTES = 256
struct PeerBlob {
    uint32_t magic, version;
    int32_t rank, nranks, device, pid;
    uint64_t window_ptr, window_bytes, layout_hash;
    int32_t finegrained, pad;
    char host[64];
    hipIpcMemHandle_t handle;
};
static_assert(sizeof(PeerBlob) <= 256, "PeerBlob must fit LQCD_PEER_BLOB_BYTES");
constexpr uint32_t PEER_MAGIC = 0x5051434cu;   // "LCQP"

// ---------------------------------------------------------------------------------- kernels
// One wave: the exchange step as a launch of its own (lqcd_internal.h peer_signal_wait_wave)
__global__ __launch_bounds__(64) void peer_sync_kernel(PeerSyncArgs a) { peer_signal_wait_wave(a); }

// sum over the ranks of n device doubles in place + the CG scalar step behind it: the stand-alone form of peer_allreduce_wave (the reduction launches of
// blas.hip / stencil.hip carry it in their own tails)
__global__ __launch_bounds__(64) void peer_allreduce_kernel(PeerRedArgs a, double* d, int n, double* scal, int cg_op) {
    const int k = (int)threadIdx.x >> 3;
    const double mine = k < n ? d[k] : 0.0;
    const double s = peer_allreduce_wave(a, mine, n);
    for (int q = 0; q < n; q++) {
        const double v = __shfl(s, 8 * q, 64);
        if (threadIdx.x == 0) d[q] = v;
    }
    if (cg_op && threadIdx.x == 0) cg_scalar_step(scal, cg_op);      // the thread that stored the sums reads them back
}
__global__ void peer_scalar_step_kernel(double* s, int op) { cg_scalar_step(s, op); }

// TEST AID (tunable halo_inject_us): the faces of an exchange "arrive" that many microseconds after they were sent -- the way a transfer over a real link would,
// on the one-GPU proxy.  comm_inject_stamp (behind the producer of the faces, in stream order) records the time of the send; comm_inject_delay (behind the exchange
// step, on its stream) is one wave that spins until stamp + delay.  A schedule that puts work between the two hides the flight time; one that does not, pays it.
__global__ void comm_stamp_kernel(unsigned long long* stamp) { *stamp = wall_clock64(); }
__global__ __launch_bounds__(64) void comm_delay_kernel(const unsigned long long* stamp, unsigned long long ticks) {
    const unsigned long long t0 = *stamp;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}
static unsigned long long* inject_word(lqcd_ctx_s* c) { return (unsigned long long*)(c->pipe_ctr + 9 * 32); }      // a line of its own in the context's counter block
int comm_inject_stamp(lqcd_ctx_s* c, hipStream_t s) {
    if (c->tun.halo_inject_us <= 0) return LQCD_OK;
    hipLaunchKernelGGL(comm_stamp_kernel, dim3(1), dim3(1), 0, s, inject_word(c));
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
int comm_inject_delay(lqcd_ctx_s* c, hipStream_t s) {
    if (c->tun.halo_inject_us <= 0) return LQCD_OK;
    hipLaunchKernelGGL(comm_delay_kernel, dim3(1), dim3(64), 0, s, inject_word(c), (unsigned long long)c->tun.halo_inject_us * 100ull);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

__global__ __launch_bounds__(256) void peer_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// ---------------------------------------------------------------------------------- peer backend: host side
PeerRedArgs comm_red_args(lqcd_ctx_s* c) {
    PeerRedArgs a;
    memset(&a, 0, sizeof a);
    if (!c->peer.on) return a;
    const uint64_t seq = ++c->peer.red_seq;
    const int slot = (int)(seq % PEER_RED_RING);
    for (int r = 0; r < c->nranks; r++) {
        a.val[r] = (double*)(c->peer.win[r] + c->peer.lay.red_val) + (size_t)slot * PEER_MAX_RANKS * PEER_RED_VALS;
        a.flag[r] = (unsigned long long*)(c->peer.win[r] + c->peer.lay.red_flag) + (size_t)slot * PEER_MAX_RANKS;
    }
    a.nranks = c->nranks; a.rank = c->rank;
    a.seq = seq; a.limit = peer_limit(c);
    a.status = c->peer.status;
    a.fence = 0;      // values and flags are system-scope atomics (they bypass the caches); completion of the value stores orders them before the flags
    return a;
}

int comm_check(lqcd_ctx_s* c) {
    if (!c->peer.on || !c->peer.status || c->peer.status[0] == 0u) return LQCD_OK;
    static const char* what[] = {"?", "a stencil face", "a scalar reduction", "a mailbox message", "a mailbox acknowledgement"};
    const unsigned w = c->peer.status[0], idx = c->peer.status[1];
    const unsigned long long seq = (unsigned long long)c->peer.status[2] | ((unsigned long long)c->peer.status[3] << 32);
    set_error(std::string("peer-mapped communication: rank ") + std::to_string(c->rank) + " gave up waiting for " + what[w < 5 ? w : 0] + " (index " + std::to_string(idx) +
              ", number " + std::to_string(seq) + ") after " + std::to_string(c->peer.timeout_ms) + " ms -- a peer rank has died, or the ranks do not issue the same sequence of exchanges");
    return LQCD_ERR_COMM;      // (sticky: the window protocol has lost a sequence number; the context cannot communicate any more)
}

double2* halo_send_base(lqcd_ctx_s* c, int mu, int toward_bwd) {
    if (!c->peer.on) return c->send_fwd[mu];
    const int nbr = toward_bwd ? c->nbr_bwd[mu] : c->nbr_fwd[mu];
    return (double2*)(c->peer.win[nbr] + c->peer.lay.ghost[mu] + (size_t)(c->peer.xchg_seq & 1) * c->peer.lay.ghost_bytes[mu]);
}
const double2* halo_recv_base(lqcd_ctx_s* c, int mu) {
    if (!c->peer.on) return c->recv_bwd[mu];
    return (const double2*)(c->peer.win[c->rank] + c->peer.lay.ghost[mu] + (size_t)((c->peer.xchg_seq + 1) & 1) * c->peer.lay.ghost_bytes[mu]);
}

// flag words of the next stencil exchange (counts it)
static PeerSyncArgs halo_sync_args(lqcd_ctx_s* c) {
    PeerSyncArgs a;
    memset(&a, 0, sizeof a);
    const uint64_t seq = ++c->peer.xchg_seq;
    int n = 0;
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        // my forward face is the +mu neighbour's "from -mu" ghost (its flag (mu, 0)); my backward face the -mu neighbour's "from +mu" ghost (its flag (mu, 1))
        a.sig[n] = flag_at(c->peer.win[c->nbr_fwd[mu]], c->peer.lay.halo_flag, 2 * mu);
        a.wt[n] = flag_at(c->peer.win[c->rank], c->peer.lay.halo_flag, 2 * mu);
        a.sig_seq[n] = a.wt_seq[n] = seq; n++;
        a.sig[n] = flag_at(c->peer.win[c->nbr_bwd[mu]], c->peer.lay.halo_flag, 2 * mu + 1);
        a.wt[n] = flag_at(c->peer.win[c->rank], c->peer.lay.halo_flag, 2 * mu + 1);
        a.sig_seq[n] = a.wt_seq[n] = seq; n++;
    }
    a.nsig = a.nwt = n;
    a.limit = peer_limit(c); a.status = c->peer.status; a.what = 1u;
    a.fence = 1;
    return a;
}
// the exchange step of a stencil application: the faces are in the neighbours' ghost buffers already (halo_send_base), what is left is the ordering
static int halo_exchange_peer(lqcd_ctx_s* c, int where) {
    const bool in_order = where == 1;
    hipStream_t xs = in_order ? c->stream : c->comm_stream;
    if (where == 0) {
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    }
    hipLaunchKernelGGL(peer_sync_kernel, dim3(1), dim3(64), 0, xs, halo_sync_args(c));
    HIPCHK(hipGetLastError());
    LQCHK(comm_inject_delay(c, xs));
    if (!in_order) HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
    return LQCD_OK;
}

static int copy_launch(lqcd_ctx_s* c, void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes % 16 != 0 || ((uintptr_t)dst | (uintptr_t)src) % 16 != 0) { set_error("peer mailbox: message not a multiple of 16 bytes"); return LQCD_ERR_ARG; }
    const size_t n16 = bytes / 16;
    const int nb = (int)std::min<size_t>((n16 + 255) / 256, (size_t)c->num_cu * 4);
    if (nb < 1) return LQCD_OK;
    hipLaunchKernelGGL(peer_copy_kernel, dim3(nb), dim3(256), 0, s, (uint4*)dst, (const uint4*)src, n16);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// Mailbox exchange (the rarer face messages: link faces, force spinors, extended-block layers).  Message m in direction (mu, dirn): wait until the receiver has
// emptied box (mu, side = dirn) of message m - 1 (its acknowledgement, in MY window) -> copy into its box -> raise its flag to m | wait for my own boxes ->
// copy out -> acknowledge to the senders.  All on `stream`, in order; every wait is a one-wave kernel.
static int sendrecv_peer(lqcd_ctx_s* c, const CommXfer* x, int n, hipStream_t s) {
    ARGCHK(n >= 0 && n <= PEER_MAX_RANKS, "peer mailbox: more than 8 messages in one group");
    if (n == 0) return LQCD_OK;
    PeerComm& P = c->peer;
    PeerSyncArgs w1, w2, w3;
    memset(&w1, 0, sizeof w1); memset(&w2, 0, sizeof w2); memset(&w3, 0, sizeof w3);
    for (int i = 0; i < n; i++) {
        const int mu = x[i].mu, d = x[i].dirn ? 1 : 0;
        ARGCHK(mu >= 0 && mu < 4 && c->geom.part[mu], "peer mailbox: direction not partitioned");
        if (x[i].bytes > P.lay.aux_cap[mu]) { set_error("peer mailbox: message of " + std::to_string(x[i].bytes) + " bytes exceeds the box of direction " + std::to_string(mu) + " (" + std::to_string(P.lay.aux_cap[mu]) + ")"); return LQCD_ERR_COMM; }
        const int box = 2 * mu + d;                          // the message arrives from -mu (dirn 0: side 0) or from +mu (dirn 1: side 1)
        // 1: my previous message in this direction has been taken out of the receiver's box
        w1.wt[i] = flag_at(P.win[c->rank], P.lay.aux_ack, box); w1.wt_seq[i] = P.aux_sent[mu][d];
        // 2: raise the receiver's flag to this message's number, wait for the message that comes to me in the same direction
        const int dst = d ? c->nbr_bwd[mu] : c->nbr_fwd[mu];
        w2.sig[i] = flag_at(P.win[dst], P.lay.aux_flag, box); w2.sig_seq[i] = P.aux_sent[mu][d] + 1;
        w2.wt[i] = flag_at(P.win[c->rank], P.lay.aux_flag, box); w2.wt_seq[i] = P.aux_rcvd[mu][d] + 1;
        // 3: acknowledge to the sender of what I received (it waits for this before its next message in this direction)
        const int src = d ? c->nbr_fwd[mu] : c->nbr_bwd[mu];
        w3.sig[i] = flag_at(P.win[src], P.lay.aux_ack, box); w3.sig_seq[i] = P.aux_rcvd[mu][d] + 1;
    }
    w1.nwt = n; w1.limit = peer_limit(c); w1.status = P.status; w1.what = 4u;
    w1.fence = 0; w2.fence = 1; w3.fence = 0;      // the data ride on step 2; steps 1 and 3 order no data (kernel boundaries do: the copies are launches of their own)
    w2.nsig = w2.nwt = n; w2.limit = w1.limit; w2.status = P.status; w2.what = 3u;
    w3.nsig = n; w3.limit = w1.limit; w3.status = P.status; w3.what = 4u;
    hipLaunchKernelGGL(peer_sync_kernel, dim3(1), dim3(64), 0, s, w1);
    for (int i = 0; i < n; i++) {
        const int mu = x[i].mu, d = x[i].dirn ? 1 : 0, dst = d ? c->nbr_bwd[mu] : c->nbr_fwd[mu];
        LQCHK(copy_launch(c, P.win[dst] + P.lay.aux_box[mu] + (size_t)d * P.lay.aux_cap[mu], x[i].send, x[i].bytes, s));
    }
    hipLaunchKernelGGL(peer_sync_kernel, dim3(1), dim3(64), 0, s, w2);
    for (int i = 0; i < n; i++) {
        const int mu = x[i].mu, d = x[i].dirn ? 1 : 0;
        LQCHK(copy_launch(c, x[i].recv, P.win[c->rank] + P.lay.aux_box[mu] + (size_t)d * P.lay.aux_cap[mu], x[i].bytes, s));
    }
    hipLaunchKernelGGL(peer_sync_kernel, dim3(1), dim3(64), 0, s, w3);
    HIPCHK(hipGetLastError());
    for (int i = 0; i < n; i++) { P.aux_sent[x[i].mu][x[i].dirn ? 1 : 0]++; P.aux_rcvd[x[i].mu][x[i].dirn ? 1 : 0]++; }
    return LQCD_OK;
}

void comm_teardown(lqcd_ctx_s* c) {
    PeerComm& P = c->peer;
    for (int r = 0; r < PEER_MAX_RANKS; r++) {
        if (P.opened[r] && P.win[r]) (void)hipIpcCloseMemHandle(P.win[r]);
        P.opened[r] = false;
        if (r != c->rank) P.win[r] = nullptr;
    }
    if (P.exported && c->rank >= 0 && c->rank < PEER_MAX_RANKS && P.win[c->rank]) { (void)hipFree(P.win[c->rank]); P.win[c->rank] = nullptr; }
    if (P.status) { (void)hipHostFree(P.status); P.status = nullptr; }
    P.on = false; P.exported = false;
}

// ---------------------------------------------------------------------------------- the backend-neutral calls
// a wait that gave up (a rank died) fails every later exchange at once instead of letting each of them time out in turn
#define PEER_FAIL_FAST(c) do { if ((c)->peer.on && (c)->peer.status && (c)->peer.status[0] != 0u) return comm_check(c); } while (0)

int comm_halo_exchange(lqcd_ctx_s* c, int kind, int parity_mode, int prec, int where) {
    ARGCHK(c->has_comm, "halo exchange: communicator not initialised (call lqcd_ctx_comm_init or lqcd_ctx_peer_init)");
    PEER_FAIL_FAST(c);
    if (c->peer.on) return halo_exchange_peer(c, where);
    return halo_exchange_rccl(c, kind, parity_mode, prec, where);
}

int comm_sendrecv(lqcd_ctx_s* c, const CommXfer* x, int n, hipStream_t stream, bool halo_comm) {
    ARGCHK(c->has_comm, "face exchange: communicator not initialised (call lqcd_ctx_comm_init or lqcd_ctx_peer_init)");
    PEER_FAIL_FAST(c);
    if (c->peer.on) return sendrecv_peer(c, x, n, stream);
    ncclComm_t comm = halo_comm ? c->comm : c->comm_red;
    NCCLCHK(ncclGroupStart());
    for (int i = 0; i < n; i++) {
        const int dst = x[i].dirn ? c->nbr_bwd[x[i].mu] : c->nbr_fwd[x[i].mu], src = x[i].dirn ? c->nbr_fwd[x[i].mu] : c->nbr_bwd[x[i].mu];
        NCCLCHK(ncclSend(x[i].send, x[i].bytes / sizeof(double), ncclDouble, dst, comm, stream));
        NCCLCHK(ncclRecv(x[i].recv, x[i].bytes / sizeof(double), ncclDouble, src, comm, stream));
    }
    NCCLCHK(ncclGroupEnd());
    return LQCD_OK;
}

int comm_allreduce(lqcd_ctx_s* c, double* d, int n, int cg_op) {
    PEER_FAIL_FAST(c);
    if (c->has_comm && c->peer.on) {
        ARGCHK(n >= 1 && n <= PEER_RED_VALS, "comm_allreduce: 1..8 values");
        hipLaunchKernelGGL(peer_allreduce_kernel, dim3(1), dim3(64), 0, c->stream, comm_red_args(c), d, n, c->d_scal, cg_op);
        HIPCHK(hipGetLastError());
        return LQCD_OK;
    }
    if (c->has_comm) NCCLCHK(ncclAllReduce(d, d, n, ncclDouble, ncclSum, c->comm_red, c->stream));
    if (cg_op) {
        hipLaunchKernelGGL(peer_scalar_step_kernel, dim3(1), dim3(1), 0, c->stream, c->d_scal, cg_op);
        HIPCHK(hipGetLastError());
    }
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

// ---------------------------------------------------------------------------------- C ABI: peer-mapped backend bootstrap
// lqcd_ctx_peer_export: allocate this rank's window, write the 256-byte description the other ranks need (IPC handle, process, device, layout check).
// The binding gathers the blobs of all ranks in rank order (MPI.Allgather in the Julia host, torch.distributed.all_gather in the Python one) ...
extern "C" int lqcd_ctx_peer_export(lqcd_ctx_t c, unsigned char blob[256]) {
    ARGCHK(c && blob, "lqcd_ctx_peer_export: null");
    ARGCHK(!c->has_comm, "lqcd_ctx_peer_export: this context already has a communicator");
    ARGCHK(c->nranks <= PEER_MAX_RANKS, "lqcd_ctx_peer_export: the peer-mapped backend serves one node (at most 8 ranks); use lqcd_ctx_comm_init (RCCL)");
    ARGCHK(c->local_peers.empty(), "lqcd_ctx_peer_export: this context belongs to an in-process PE grid");
    HIPCHK(hipSetDevice(c->device));
    PeerComm& P = c->peer;
    if (!P.exported) {
        if (const char* e = getenv("LQCD_PEER_FINEGRAINED")) P.finegrained = atoi(e) ? 1 : 0;
        P.lay = peer_layout(c->geom);
        char* w = nullptr;
        if (P.finegrained) HIPCHK(hipExtMallocWithFlags((void**)&w, P.lay.total, hipDeviceMallocFinegrained));
        else HIPCHK(hipMalloc((void**)&w, P.lay.total));
        HIPCHK(hipMemset(w, 0, P.lay.total));
        HIPCHK(hipDeviceSynchronize());
        P.win[c->rank] = w;
        HIPCHK(hipHostMalloc((void**)&P.status, 64, hipHostMallocDefault));
        memset(P.status, 0, 64);
        P.exported = true;
    }
    PeerBlob b;
    memset(&b, 0, sizeof b);
    b.magic = PEER_MAGIC; b.version = 1;
    b.rank = c->rank; b.nranks = c->nranks; b.device = c->device; b.pid = (int32_t)getpid();
    b.window_ptr = (uint64_t)(uintptr_t)P.win[c->rank]; b.window_bytes = P.lay.total; b.layout_hash = peer_layout_hash(c);
    b.finegrained = P.finegrained;
    (void)gethostname(b.host, sizeof b.host - 1);
    HIPCHK(hipIpcGetMemHandle(&b.handle, P.win[c->rank]));
    memset(blob, 0, 256);
    memcpy(blob, &b, sizeof b);
    return LQCD_OK;
}

// ... and hands them to every rank: lqcd_ctx_peer_init maps the other ranks' windows (a window of this very process -- the self-partitioned one-rank proxy -- is used
// through its own address) and switches the context's exchanges and reductions to them.
extern "C" int lqcd_ctx_peer_init(lqcd_ctx_t c, const unsigned char* blobs, int nranks) {
    ARGCHK(c && blobs, "lqcd_ctx_peer_init: null");
    ARGCHK(nranks == c->nranks, "lqcd_ctx_peer_init: nranks does not match the PE grid");
    ARGCHK(!c->has_comm, "lqcd_ctx_peer_init: communicator already initialised");
    PeerComm& P = c->peer;
    ARGCHK(P.exported, "lqcd_ctx_peer_init: call lqcd_ctx_peer_export first (every rank, then gather the blobs in rank order)");
    HIPCHK(hipSetDevice(c->device));
    char host[64] = {0};
    (void)gethostname(host, sizeof host - 1);
    for (int r = 0; r < nranks; r++) {
        PeerBlob b;
        memcpy(&b, blobs + (size_t)256 * r, sizeof b);
        auto bad = [&](const std::string& why) {
            set_error("lqcd_ctx_peer_init (rank " + std::to_string(c->rank) + "): blob " + std::to_string(r) + " " + why);
            for (int q = 0; q < r; q++) if (P.opened[q]) { (void)hipIpcCloseMemHandle(P.win[q]); P.opened[q] = false; P.win[q] = nullptr; }
            return LQCD_ERR_COMM;
        };
        if (b.magic != PEER_MAGIC || b.version != 1) return bad("is not a window description (gather the 256-byte blobs of lqcd_ctx_peer_export in rank order)");
        if (b.rank != r || b.nranks != nranks) return bad("belongs to rank " + std::to_string(b.rank) + " of " + std::to_string(b.nranks));
        if (b.layout_hash != peer_layout_hash(c) || b.window_bytes != P.lay.total) return bad("was made for another lattice / PE grid");
        if (strncmp(b.host, host, sizeof host) != 0) return bad(std::string("lives on host ") + b.host + ": the peer-mapped backend serves one node, use lqcd_ctx_comm_init (RCCL)");
        if (r == c->rank) {
            if ((uint64_t)(uintptr_t)P.win[r] != b.window_ptr || b.pid != (int32_t)getpid()) return bad("is not the one this context exported");
            continue;
        }
        if (b.pid == (int32_t)getpid()) { P.win[r] = (char*)(uintptr_t)b.window_ptr; continue; }      // a window of this process: no mapping needed
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, b.handle, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return bad(std::string("could not be mapped: ") + hipGetErrorString(e) + " (device " + std::to_string(b.device) + ", process " + std::to_string(b.pid) +
                       "; HSA_ENABLE_IPC_MODE_LEGACY=0 must be set in every rank)");
        }
        P.win[r] = (char*)p;
        P.opened[r] = true;
    }
    // the stencil's own receive buffers are not used any more (the ghosts live in the window); the send buffers neither (producers store into the neighbours' windows)
    P.on = true;
    c->has_comm = true;
    return LQCD_OK;
}

extern "C" int lqcd_ctx_comm_backend(lqcd_ctx_t c, int* backend) {
    ARGCHK(c && backend, "lqcd_ctx_comm_backend: null");
    *backend = !c->has_comm ? LQCD_COMM_NONE : (c->peer.on ? LQCD_COMM_PEER : LQCD_COMM_RCCL);
    return LQCD_OK;
}
