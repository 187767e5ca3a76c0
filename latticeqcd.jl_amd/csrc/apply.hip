// apply.hip -- operator application on one rank: halo exchange (RCCL), the pack / interior / exterior schedule, the mapping from an
// operator object to stencil calls, and the operator part of the C ABI.
//
// Replaces, behind the C ABI, LatticeDiracOperators.jl's mul!(y,D,x) / mul!(y,D',x) / DdagD_operator -- SURVEY.md 8(a) a1-a3, a7;
// reference call sites /root/reference/src/system/universe.jl:103-137, src/md/standardMD.jl:95-96.
#include "ops_internal.h"
#include <cstring>

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>

namespace lqcd {

// ---------------------------------------------------------------------------------- halo exchange
static size_t halo_count(lqcd_ctx_s* c, int mu, int kind, int parity_mode) {
    const int nh = kind == LQCD_WILSON ? 6 : 3;
    return (size_t)(parity_mode == 2 ? 2 : 1) * nh * face_half_sites(c->geom, mu);
}

// Buffers: send = [fwd face | bwd face], recv = [from bwd | from fwd], packed back to back for the message size of the call
// (stencil.hip make_hargs uses the same rule), elements of 16 bytes (fp64) or 8 bytes (fp32).
// RCCL path: grouped send/recv on the communication stream, so the transfer over xGMI overlaps the interior stencil running
// on the compute stream.  When both neighbours of a direction are the same rank (PE extent 2) the two faces are ONE message
// each way: my [fwd | bwd] lands in its [from bwd | from fwd].
// where = 0: the exchange runs on the communication stream behind an event of the pack kernel (compute stream) and signals ev_comm;
// where = 1: it is enqueued on the compute stream itself, right behind the pack kernel (halo_stream_mode = 1);
// where = 2: on the communication stream, which already holds the pack kernel (halo_stream_mode = 2); signals ev_comm
int halo_exchange_rccl(lqcd_ctx_s* c, int kind, int parity_mode, int prec, int where) {
    const bool in_order = where == 1;
    const ncclDataType_t dt = prec ? ncclFloat : ncclDouble;   // same element counts, float2 instead of double2 elements
    const size_t esize = prec ? sizeof(float2) : sizeof(double2);
    ARGCHK(c->has_comm, "halo exchange: communicator not initialised (call lqcd_ctx_comm_init)");
    hipStream_t xs = in_order ? c->stream : c->comm_stream;
    if (where == 0) {
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    }
    NCCLCHK(ncclGroupStart());
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        const size_t cnt = halo_count(c, mu, kind, parity_mode), n = cnt * 2;  // n: scalars per face
        char* sf = (char*)c->send_fwd[mu];
        char* rb = (char*)c->recv_bwd[mu];
        if (c->tun.halo_merge && c->nbr_fwd[mu] == c->nbr_bwd[mu]) {
            NCCLCHK(ncclSend(sf, 2 * n, dt, c->nbr_fwd[mu], c->comm, xs));
            NCCLCHK(ncclRecv(rb, 2 * n, dt, c->nbr_bwd[mu], c->comm, xs));
            continue;
        }
        NCCLCHK(ncclSend(sf, n, dt, c->nbr_fwd[mu], c->comm, xs));
        NCCLCHK(ncclSend(sf + cnt * esize, n, dt, c->nbr_bwd[mu], c->comm, xs));
        NCCLCHK(ncclRecv(rb, n, dt, c->nbr_bwd[mu], c->comm, xs));
        NCCLCHK(ncclRecv(rb + cnt * esize, n, dt, c->nbr_fwd[mu], c->comm, xs));
    }
    NCCLCHK(ncclGroupEnd());
    LQCHK(comm_inject_delay(c, xs));
    if (!in_order) HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
    return LQCD_OK;
}

// in-process emulation (fp64): every rank has packed; copy sender buffers into the peers' receive buffers
int halo_exchange_local_all(lqcd_ctx_s** ctxs, int n, int kind, int parity_mode) {
    for (int r = 0; r < n; r++) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    for (int r = 0; r < n; r++) {
        lqcd_ctx_s* c = ctxs[r];
        for (int mu = 0; mu < 4; mu++) {
            if (!c->geom.part[mu]) continue;
            const size_t cnt = halo_count(c, mu, kind, parity_mode), bytes = cnt * sizeof(double2);
            // my fwd face lands in the +mu neighbour's "from bwd" half; my bwd face in the -mu neighbour's "from fwd" half
            HIPCHK(hipMemcpy(ctxs[c->nbr_fwd[mu]]->recv_bwd[mu], c->send_fwd[mu], bytes, hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(ctxs[c->nbr_bwd[mu]]->recv_bwd[mu] + cnt, c->send_fwd[mu] + cnt, bytes, hipMemcpyDeviceToDevice));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    return LQCD_OK;
}

// Wilson, r != 1, as two r = 1 calls (see stencil_apply): the first forms a xin + b (1+r)/2 H in, the second adds b (r-1)/2 H' in, H' being
// the hop sum with the opposite projector sign (the dagger form).  Plain mode: the second call reads its diagonal term from `out`
// (each thread reads its own site before it writes it; the exterior kernel read-modify-writes).  CG update mode (r -= alpha v is
// linear in v): the second call runs with a = 0.  The |.|^2 partials of the second call are those of the complete result.
void split_general_r(const StencilCall& s, StencilCall& s1, StencilCall& s2) {
    s1 = s; s2 = s;
    s1.r = 1.0; s2.r = 1.0;
    s1.b = s.b * 0.5 * (1.0 + s.r);
    s2.b = s.b * 0.5 * (s.r - 1.0);
    s2.dagger = s.dagger ? 0 : 1;
    s1.gauge12 = nullptr; s2.gauge12 = nullptr;
    if (s.upd_scal) {
        s2.a = 0.0;
    } else {
        s2.a = 1.0;
        s2.xin[0] = s.out[0]; s2.xin[1] = s.out[1];
        s2.clover = nullptr;           // a fused clover term belongs to the first call's diagonal term only
    }
}

bool any_partitioned(lqcd_ctx_s* c) {
    return c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3];
}

// full stencil on one rank: pack -> (exchange || interior) -> exterior
int flush_waiting_pack(lqcd_ctx_s* c) {      // a pack launch left waiting by the folded schedule whose reduction never came: run it on its own
    if (!c->has_waiting_pack) return LQCD_OK;
    c->has_waiting_pack = false;
    return launch_stencil_pack(c, *static_cast<StencilCall*>(c->waiting_pack));
}

int stencil_apply(lqcd_ctx_s* c, const StencilCall& s) {
    HIPCHK(hipSetDevice(c->device));
    LQCHK(flush_waiting_pack(c));
    c->tun.halo_fold_active = 0;      // (before any early return: the key describes THIS application)
    if (s.prec == 2) return launch_pair32_interior(c, s);      // fp32 site-pair fields (unpartitioned lattices only: checked by the launcher)
    if (!any_partitioned(c)) return s.prec ? p32::launch_stencil_interior(c, s) : launch_stencil_interior(c, s);
    if (s.kind == LQCD_WILSON && s.r != 1.0) {
        // general r on a partitioned lattice: the halos carry spin-projected half spinors, i.e. r = 1 hops.  r -+ gamma is a combination
        // of the two projectors, (r - gamma) = (1+r)/2 (1 - gamma) + (r-1)/2 (1 + gamma), so the general-r hop sum is
        // (1+r)/2 H + (r-1)/2 H^dagger-form: two r = 1 applications through the same pack / exchange / exterior sequence.
        StencilCall s1, s2;
        split_general_r(s, s1, s2);
        LQCHK(stencil_apply(c, s1));
        return stencil_apply(c, s2);
    }
    ARGCHK(c->local_peers.empty(), "this context belongs to an in-process PE grid: use the lqcd_mdom_* collectives");
    // norm partials: the interior writes |.|^2 of what it produced, the exterior appends the corrections of the sites it updates
    if (c->tun.halo_stream_mode < 0) {
        // auto: time the five schedules once on the first plain full-lattice application (idempotent: it only rewrites `out`).  Every rank
        // issues the same exchanges whichever schedule it ends up with, so the choice is local.
        if (s.parity_mode != 2 || s.upd_scal || s.upd[0] || s.upd[1]) {
            c->tun.halo_stream_mode = 0;
            const int st = stencil_apply(c, s);
            c->tun.halo_stream_mode = -1;
            return st;
        }
        constexpr int NM = 5;
        float ms[NM] = {0.f, 0.f, 0.f, 0.f, 0.f};
        StencilCall t = s;          // the timed applications pack for themselves and leave the fused tails alone
        t.prepacked = 0; t.pack_next = -1; t.red_slot = -1;
        const bool tails = s.pack_next >= 0 || s.red_slot >= 0;
        // two rounds over the schedules, eight applications each, the smaller time counts: four applications once (round 5) picked a schedule that was 5 % slower than
        // schedule 3 at the N = 2 local volume (profiles/r06_proxy_scaling_peer.log: 1674 vs 1760 CG iterations / s)
        for (int round = 0; round < 2; round++)
            for (int mode = 0; mode < NM; mode++) {
                c->tun.halo_stream_mode = mode;
                LQCHK(stencil_apply(c, t));
                HIPCHK(hipEventRecord(c->ev_tune0, c->stream));
                for (int k = 0; k < 8; k++) LQCHK(stencil_apply(c, t));
                HIPCHK(hipEventRecord(c->ev_tune1, c->stream));
                HIPCHK(hipEventSynchronize(c->ev_tune1));
                float tm = 0.f;
                HIPCHK(hipEventElapsedTime(&tm, c->ev_tune0, c->ev_tune1));
                tm *= 0.5f;      // (the figures below are per four applications, as before)
                if (round == 0 || tm < ms[mode]) ms[mode] = tm;
            }
        // the choice is collective: every rank adopts the schedule with the smallest time summed over the ranks (one 40-byte all-reduce), so that
        // no two ranks interleave their sends and receives differently and a slow rank's view counts
        if (c->has_comm) {      // (a one-rank communicator -- self-partition tests -- takes the same path: the all-reduce is then the identity)
            double* d_ms = c->d_scal + SCAL_DOUBLES - 8;      // (the host-value all-reduce's staging doubles)
            double dms[NM];
            for (int mode = 0; mode < NM; mode++) dms[mode] = ms[mode];
            HIPCHK(hipMemcpyAsync(d_ms, dms, sizeof(dms), hipMemcpyHostToDevice, c->stream));
            LQCHK(comm_allreduce(c, d_ms, NM));      // on the compute stream
            HIPCHK(hipMemcpyAsync(dms, d_ms, sizeof(dms), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            for (int mode = 0; mode < NM; mode++) ms[mode] = (float)dms[mode] / (float)c->nranks;
        }
        // the one-stream folded schedule is what the solvers' fused tails are built around (pack in the reduction launch, no pack launch at all): another schedule has to
        // beat it by more than 3 % to be taken
        int best = 3;
        for (int mode = 0; mode < NM; mode++)
            if (mode != 3 && ms[mode] < 0.97f * ms[3] && (best == 3 || ms[mode] < ms[best])) best = mode;
        c->tun.halo_stream_mode = best;
        for (int mode = 0; mode < NM; mode++) c->tun.halo_tuned_us[mode] = (int)(250.f * ms[mode]);
        if (tails) { t = s; t.prepacked = 0; return stencil_apply(c, t); }      // once more with the caller's tails (the send buffers hold this input's faces)
        return LQCD_OK;      // `out` holds the result of the last tuning application
    }
    const int mode = c->tun.halo_stream_mode;
    ARGCHK(mode >= 0 && mode <= 4, "halo_stream_mode must be -1 (timed once) or 0..4");
    auto launch_interior = [&](const StencilCall& q) { return q.prec ? p32::launch_stencil_interior(c, q) : launch_stencil_interior(c, q); };
    auto launch_pack = [&](const StencilCall& q) { return q.prec ? p32::launch_stencil_pack(c, q) : launch_stencil_pack(c, q); };
    auto launch_exterior = [&](const StencilCall& q) { return q.prec ? p32::launch_stencil_exterior(c, q) : launch_stencil_exterior(c, q); };
    auto on_comm_stream = [&](const std::function<int()>& f) {      // the launchers enqueue on c->stream
        hipStream_t main_stream = c->stream;
        c->stream = c->comm_stream;
        const int st = f();
        c->stream = main_stream;
        return st;
    };
    // ---- folded schedules (halo_fold): the stencil kernel takes the boundary hops from the ghost buffers itself -- no exterior kernel, no norm corrections, complete
    // |.|^2 partials (stencil_num_partials follows halo_fold_applies).  Schedule 3 (round 5): everything in order on one stream, ONE launch behind the exchange.
    // Round 6: the same kernel runs on two disjoint sets of chunks -- "bulk" (no site on a partitioned face; needs no ghost) beside the exchange, "boundary" after
    // arrival -- which gives the overlapping schedules 0-2 the fold (they used to fall back to interior + exterior) and a new schedule 4: pack -> bulk -> exchange step ->
    // boundary in order on ONE stream.  With the peer-mapped backend the faces travel while the bulk runs (the pack stored them into the neighbour's window; the
    // exchange step is only the wait), and there is no cross-queue join (~13 us) to pay for the overlap.
    if (halo_fold_applies(c, s.kind, s.r, s.parity_mode, s.prec, s.clover != nullptr) && !s.dot_partial && !s.alpha_partials && s.dw_ls <= 1) {
        StencilCall all = s, bulk = s, bnd = s;
        all.fold = 1; bulk.fold = 2; bnd.fold = 3;
        c->tun.halo_fold_active = 1;
        if (mode == 3) {
            if (!s.prepacked) LQCHK(launch_pack(s));
            LQCHK(comm_inject_stamp(c, c->stream));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 1));
            LQCHK(launch_interior(all));
        } else if (mode == 4) {
            if (!s.prepacked) LQCHK(launch_pack(s));
            LQCHK(comm_inject_stamp(c, c->stream));
            LQCHK(launch_interior(bulk));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 1));
            LQCHK(launch_interior(bnd));
        } else if (mode == 0) {          // exchange on the communication stream behind the pack, bulk on the compute stream
            if (!s.prepacked) LQCHK(launch_pack(s));
            LQCHK(comm_inject_stamp(c, c->stream));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 0));
            LQCHK(launch_interior(bulk));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
            LQCHK(launch_interior(bnd));
        } else if (mode == 2) {          // bulk enqueued first; pack -> exchange on the communication stream behind an event
            HIPCHK(hipEventRecord(c->ev_pack, c->stream));
            HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
            LQCHK(launch_interior(bulk));
            if (!s.prepacked) LQCHK(on_comm_stream([&] { return launch_pack(s); }));
            LQCHK(comm_inject_stamp(c, c->comm_stream));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 2));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
            LQCHK(launch_interior(bnd));
        } else {                         // 1: pack -> exchange -> boundary in order on the compute stream, the bulk beside them on the second stream
            HIPCHK(hipEventRecord(c->ev_pack, c->stream));
            HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
            LQCHK(on_comm_stream([&] { return launch_interior(bulk); }));
            HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
            if (!s.prepacked) LQCHK(launch_pack(s));
            LQCHK(comm_inject_stamp(c, c->stream));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 1));
            HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
            LQCHK(launch_interior(bnd));
        }
        // a following application's faces (pack_next: the D p -> D^+ pair of the fused CG) are packed from the finished output by a pack launch
        if (s.pack_next >= 0) {
            StencilCall pk = s;
            pk.in[0] = s.out[0]; pk.in[1] = s.out[1];
            pk.dagger = s.pack_next;
            if (s.defer_pack && !s.prec && s.kind == LQCD_WILSON) {      // the caller's reduction of this application's partials is the next launch: the pack rides in it (reduce_pack_to_slot)
                if (!c->waiting_pack) c->waiting_pack = new StencilCall;
                *static_cast<StencilCall*>(c->waiting_pack) = pk;
                c->has_waiting_pack = true;
            } else LQCHK(launch_pack(pk));
        }
        return LQCD_OK;
    }
    // ---- interior + exterior schedules (halo_fold = 0, or a call / kernel form without a folded instance)
    if (mode == 1) {
        // pack -> exchange -> exterior stay in order on the compute stream (no queue hop on the path that carries the messages);
        // the interior runs beside them on the second stream, forked and joined by events
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));                 // the inputs of this call are complete
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
        LQCHK(on_comm_stream([&] { return launch_interior(s); }));
        HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
        if (!s.prepacked) LQCHK(launch_pack(s));
        LQCHK(comm_inject_stamp(c, c->stream));
        LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 1));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        return launch_exterior(s);
    }
    if (mode == 3 || mode == 4) {
        // everything in order on the compute stream, no overlap and no cross-queue join: pack -> exchange -> interior -> exterior.  A join costs
        // ~13 us (barrier packets) and the exchange kernel slows the interior it runs beside; at small local volumes with a short exchange that
        // is more than the overlap hides.  (4 without a folded instance: the interior in front of the exchange step.)
        if (!s.prepacked) LQCHK(launch_pack(s));
        LQCHK(comm_inject_stamp(c, c->stream));
        if (mode == 4) LQCHK(launch_interior(s));
        LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 1));
        if (mode == 3) LQCHK(launch_interior(s));
        return launch_exterior(s);
    }
    if (mode == 2) {
        // the interior is enqueued FIRST on the compute stream (the GPU starts it while the host is still busy issuing the RCCL group)
        // and stays in order with the exterior; pack -> exchange run on the second stream behind an event -- the schedule for an
        // exchange that is shorter than the interior: the fork / join latencies hide behind the interior kernel
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));                 // the inputs of this call are complete
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
        LQCHK(launch_interior(s));
        if (!s.prepacked) LQCHK(on_comm_stream([&] { return launch_pack(s); }));
        LQCHK(comm_inject_stamp(c, c->comm_stream));
        LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 2));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        return launch_exterior(s);
    }
    if (!s.prepacked) LQCHK(launch_pack(s));
    LQCHK(comm_inject_stamp(c, c->stream));
    LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, s.prec, 0));
    LQCHK(launch_interior(s));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    return launch_exterior(s);
}

// The halo schedule of a partitioned context is chosen by timing, at the first plain full-lattice application (halo_stream_mode = -1 above), and the choice decides
// how many |.|^2 partials an application writes (the folded schedule has no exterior corrections).  A solver that counts partials settles the choice first.
int halo_schedule_settle(lqcd_op_s* op) {
    lqcd_ctx_s* c = op->ctx;
    if (c->tun.halo_stream_mode >= 0 || !any_partitioned(c) || !c->has_comm || !c->local_peers.empty() || op->kind == LQCD_DOMAINWALL) return LQCD_OK;
    lqcd_spinor_s* a = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* b = scratch_get(c, op->kind, LQCD_FULL);
    int st = (a && b) ? LQCD_OK : LQCD_ERR_HIP;
    if (st == LQCD_OK && hipMemsetAsync(a->data, 0, a->elems * sizeof(double2), c->stream) != hipSuccess) st = LQCD_ERR_HIP;
    if (st == LQCD_OK) st = op_apply_async(op, b, a, 0, nullptr);
    if (st == LQCD_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = LQCD_ERR_HIP;
    if (a) scratch_put(a);
    if (b) scratch_put(b);
    return st;
}

// ---------------------------------------------------------------------------------- operator -> stencil calls
static void fill_blocks(const double2* dst[2], lqcd_spinor_s* s) {
    dst[0] = s ? spinor_block(s, 0) : nullptr;
    dst[1] = s ? spinor_block(s, 1) : nullptr;
}

// opt-in 12-real links for the Wilson r = 1 split kernel (tunable gauge_recon = 12): lazily (re)built, used only when every
// link of the current field is unitary to 1e-14, otherwise the 18-real field is read as usual
static const double2* recon12_links(lqcd_op_s* op, int* delta = nullptr) {
    lqcd_ctx_s* c = op->ctx;
    c->tun.recon_active = 0;
    if (delta) *delta = 0;
    if (c->tun.gauge_recon != 12) return nullptr;
    if (op->kind == LQCD_WILSON && (op->r != 1.0 || (c->tun.dslash_variant != 1 && c->tun.dslash_variant < 4))) return nullptr;   // only the direction-split kernels
    if (op->kind == LQCD_STAGGERED && !(c->tun.dslash_variant >= 1 && c->tun.dslash_variant <= 8)) return nullptr;
    if (gauge_ensure_recon12(op->gauge) != LQCD_OK) return nullptr;
    if (op->gauge->recon_ok) {
        c->tun.recon_active = 1;
        return op->gauge->data12;
    }
    // not on the group to 1e-14 (a reference-format configuration): rows 0, 1 + the fp32 deviation of row 2, for the launches that can read it
    // (the scalar-addressing Wilson kernel; stencil.hip declines for the others and reads the 18 stored reals)
    if (delta && c->tun.gauge_delta && op->kind == LQCD_WILSON && c->tun.dslash_variant == 1 && c->tun.dslash_pipe == 2 && op->gauge->recon_dev <= 1e-9 &&
        gauge_ensure_recon12d(op->gauge) == LQCD_OK && op->gauge->delta_ok) {
        *delta = 1;
        c->tun.recon_active = 2;
        return op->gauge->data12d;
    }
    return nullptr;
}

// out = D in  /  D^+ in on FULL spinors.  Wilson-clover: A follows the links lazily (rebuilt here when the field's version moved;
// clover_version changes only if the build succeeded) and, unless the split kernel applies it in its epilogue, A in is formed by
// a separate pass enqueued here -- both return their status to the caller.
int make_full_call(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, StencilCall& s) {
    s = StencilCall();
    s.kind = op->kind;
    s.gauge = op->gauge->data;
    s.out[0] = spinor_block(out, 0);
    s.out[1] = spinor_block(out, 1);
    fill_blocks(s.in, in);
    fill_blocks(s.xin, in);
    if (op->kind == LQCD_WILSON) { s.a = 1.0; s.b = -op->km; }
    else { s.a = op->km; s.b = dagger ? -0.5 : 0.5; }
    s.r = op->r;
    s.dagger = dagger;
    s.parity_mode = 2;
    s.norm_partial = nullptr;
    s.gauge12 = recon12_links(op, &s.gauge12_delta);
    if (op->csw != 0.0 && op->clover && op->clover_tmp) {
        LQCHK(op_refresh_clover(op));
        if (op->r == 1.0 && op->ctx->tun.dslash_variant == 1 && op->ctx->tun.clover_fused) {
            s.clover = op->clover;            // fused: the direction-split kernel forms A in in its epilogue (one pass, 1536 B/site)
        } else {
            LQCHK(clover_apply(op->ctx, op->clover, op->clover_tmp, in));      // separate streaming pass, then xin = A in
            fill_blocks(s.xin, op->clover_tmp);
        }
    }
    return LQCD_OK;
}

int op_refresh_clover(lqcd_op_s* op) {
    if (op->csw != 0.0 && op->clover && op->clover_version != op->gauge->version) {
        LQCHK(clover_build(op->ctx, op->gauge, op->clover, op->km, op->csw));
        op->clover_version = op->gauge->version;       // only a successful build marks A as current
    }
    return LQCD_OK;
}

// out(parity subset) = a*xin + b*H in, in of the opposite subset
StencilCall make_hop_call(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, lqcd_spinor_s* xin, double a, double b, int dagger) {
    StencilCall s;
    s.kind = op->kind;
    s.gauge = op->gauge->data;
    s.out[0] = spinor_block(out, 0);
    s.out[1] = spinor_block(out, 1);
    fill_blocks(s.in, in);
    fill_blocks(s.xin, xin);
    s.a = a;
    s.b = (op->kind == LQCD_STAGGERED) ? b * (dagger ? -0.5 : 0.5) : b;
    s.r = op->r;
    s.dagger = dagger;
    s.parity_mode = out->subset == LQCD_EVEN ? 0 : 1;
    s.norm_partial = nullptr;
    s.gauge12 = recon12_links(op, &s.gauge12_delta);
    return s;
}

void apply_bc(lqcd_ctx_s* c, const int bc[4]) {
    for (int mu = 0; mu < 4; mu++) {
        // unpartitioned direction: this rank owns both ends, a local wrap is a global wrap
        c->geom.bc_fwd[mu] = (double)bc[mu];
        c->geom.bc_bwd[mu] = (double)bc[mu];
    }
}

int check_full(lqcd_op_s* op, lqcd_spinor_s* a, lqcd_spinor_s* b, const char* who) {
    if (!(op && a && b && a->ctx == op->ctx && b->ctx == op->ctx && a->kind == op->kind && b->kind == op->kind &&
          a->subset == LQCD_FULL && b->subset == LQCD_FULL && a != b && a->data != b->data)) {
        set_error(std::string(who) + ": need two distinct FULL spinors of the operator's kind on the operator's context");
        return LQCD_ERR_ARG;
    }
    if (op->kind == LQCD_DOMAINWALL && strncmp(who, "dw:", 3) != 0) {      // (domainwall.hip passes its own names)
        set_error(std::string(who) + ": not available for the Domainwall operator (mul!, DdagD, the CG and the action entry points are)");
        return LQCD_ERR_UNSUPPORTED;
    }
    return links_flush_of(op);      // the operator reads its links: recorded single-direction link operations run first (md.hip)
}

int op_apply_async(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, double* norm_partial, const double* skip_flag) {
    apply_bc(op->ctx, op->bc);
    StencilCall s;
    LQCHK(make_full_call(op, out, in, dagger, s));
    s.norm_partial = norm_partial;
    s.skip_flag = skip_flag;
    return stencil_apply(op->ctx, s);
}

}  // namespace lqcd

using namespace lqcd;

// ---------------------------------------------------------------------------------- C API: operator
extern "C" int lqcd_op_create(lqcd_ctx_t ctx, lqcd_op_t* op, int kind, lqcd_gauge_t g, double km, double r, const int bc[4]) {
    ARGCHK(kind != LQCD_DOMAINWALL, "lqcd_op_create: the Domainwall operator is made by lqcd_op_create_domainwall (it needs M and L5)");
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(ctx && op && g && bc, "lqcd_op_create: null argument");
    ARGCHK(kind == LQCD_WILSON || kind == LQCD_STAGGERED, "lqcd_op_create: Dirac_operator not supported");
    ARGCHK(g->ctx == ctx, "lqcd_op_create: gauge field belongs to another context");
    for (int mu = 0; mu < 4; mu++) ARGCHK(bc[mu] == 1 || bc[mu] == -1, "lqcd_op_create: boundarycondition entries must be +1 or -1");
    lqcd_op_s* o = new lqcd_op_s;
    o->ctx = ctx; o->kind = kind; o->gauge = g; o->km = km; o->r = r;
    for (int mu = 0; mu < 4; mu++) o->bc[mu] = bc[mu];
    *op = o;
    return LQCD_OK;
}
extern "C" int lqcd_op_destroy(lqcd_op_t op) {
    if (!op) return LQCD_OK;
    (void)hipFree(op->clover);
    (void)hipFree(op->clover_inv);
    (void)hipFree(op->clover_lambda);
    if (op->clover_tmp) lqcd_spinor_destroy(op->clover_tmp);
    if (op->dw_wilson) lqcd_op_destroy(op->dw_wilson);
    for (lqcd_spinor_s* w : op->dw_work) if (w) lqcd_spinor_destroy(w);
    (void)hipFree(op->dw_partial);
    delete op;
    return LQCD_OK;
}

// Dirac_operator = "WilsonClover", Clover_coefficient (parameter_structs.jl:125; test/test_wilsonclover.toml:9): D_sw = D + (A - 1),
// A = 1 + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu} (clover.hip).  csw = 0 switches the term off again.
extern "C" int lqcd_op_set_clover(lqcd_op_t op, double csw) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op, "lqcd_op_set_clover: null argument");
    ARGCHK(op->kind == LQCD_WILSON, "lqcd_op_set_clover: the clover term belongs to the Wilson operator");
    lqcd_ctx_s* c = op->ctx;
    if (csw != 0.0 && any_partitioned(c) && !c->local_peers.empty()) {
        set_error("lqcd_op_set_clover: not available on an in-process PE grid (RCCL ranks only)");
        return LQCD_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    op->csw = csw;
    if (csw == 0.0) return LQCD_OK;
    if (!op->clover) HIPCHK(hipMalloc((void**)&op->clover, clover_elems(c->geom) * sizeof(double2)));
    if (!op->clover_tmp) LQCHK(lqcd_spinor_create(c, &op->clover_tmp, LQCD_WILSON, LQCD_FULL));
    op->clover_version = 0;
    op->clover_inv_version = 0;      // A^-1 of the even-odd solver belongs to the previous (links, csw): rebuilt at its next use
    LQCHK(clover_build(c, op->gauge, op->clover, op->km, csw));
    op->clover_version = op->gauge->version;
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}
extern "C" int lqcd_op_set_gauge(lqcd_op_t op, lqcd_gauge_t g) {
    ARGCHK(op && g && g->ctx == op->ctx, "lqcd_op_set_gauge: bad gauge field");
    op->gauge = g;
    if (op->dw_wilson) op->dw_wilson->gauge = g;
    op->clover_version = 0;   // another field: the clover term is rebuilt at the next application
    op->clover_inv_version = 0;
    return LQCD_OK;
}

extern "C" int lqcd_op_apply(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger) {
    if (op && op->kind == LQCD_DOMAINWALL) return dw_op_apply(op, out, in, dagger);
    LQCHK(check_full(op, out, in, "lqcd_op_apply"));
    LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return comm_check(op->ctx);      // peer-mapped backend: a face that never came (dead rank) is an error, not a result
}

extern "C" int lqcd_op_apply_DdagD(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in) {
    if (op && op->kind == LQCD_DOMAINWALL) return dw_op_apply_DdagD(op, out, in);
    LQCHK(check_full(op, out, in, "lqcd_op_apply_DdagD"));
    lqcd_spinor_s* tmp = scratch_get(op->ctx, op->kind, LQCD_FULL);
    if (!tmp) return LQCD_ERR_HIP;
    int st = op_apply_async(op, tmp, in, 0, nullptr);
    if (st == LQCD_OK) st = op_apply_async(op, out, tmp, 1, nullptr);
    hipError_t e = hipStreamSynchronize(op->ctx->stream);
    scratch_put(tmp);
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync DdagD", __FILE__, __LINE__);
    if (st == LQCD_OK) st = comm_check(op->ctx);
    return st;
}

extern "C" int lqcd_op_hop(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op && out && in && out->ctx == op->ctx && in->ctx == op->ctx && out->kind == op->kind && in->kind == op->kind,
           "lqcd_op_hop: bad arguments");
    ARGCHK((out->subset == LQCD_EVEN && in->subset == LQCD_ODD) || (out->subset == LQCD_ODD && in->subset == LQCD_EVEN),
           "lqcd_op_hop: out and in must be opposite parity subsets");
    apply_bc(op->ctx, op->bc);
    StencilCall s = make_hop_call(op, out, in, nullptr, 0.0, 1.0, dagger ? 1 : 0);
    LQCHK(stencil_apply(op->ctx, s));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return LQCD_OK;
}
