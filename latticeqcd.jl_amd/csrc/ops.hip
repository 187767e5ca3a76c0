// ops.hip -- operator application (with halo exchange), Krylov solvers, timing entry points.
//
// Replaces, behind the C ABI, LatticeDiracOperators.jl's mul!(y,D,x), DdagD_operator, and solve_DinvX!
// (cg / bicgstab / even-odd bicgstab) -- SURVEY.md 8(a) a2-a5; reference call sites
// /root/reference/src/md/AbstractMD.jl:129, src/updates/standardHMC.jl:71, src/md/standardMD.jl:95-96.
#include "lqcd_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>

namespace lqcd {

double2* spinor_block(lqcd_spinor_s* s, int p);
int stream_grid(lqcd_ctx_s* c, size_t n);

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

bool any_partitioned(lqcd_ctx_s* c) {
    return c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3];
}

// full stencil on one rank: pack -> (exchange || interior) -> exterior
int stencil_apply(lqcd_ctx_s* c, const StencilCall& s) {
    HIPCHK(hipSetDevice(c->device));
    if (!any_partitioned(c)) return s.prec ? p32::launch_stencil_interior(c, s) : launch_stencil_interior(c, s);
    if (s.kind == LQCD_WILSON && s.r != 1.0) {
        set_error("Wilson r != 1 is not supported on a partitioned lattice (halos carry spin-projected half spinors)");
        return LQCD_ERR_UNSUPPORTED;
    }
    ARGCHK(c->local_peers.empty(), "this context belongs to an in-process PE grid: use the lqcd_mdom_* collectives");
    // norm partials: the interior writes |.|^2 of what it produced, the exterior appends the corrections of the sites it updates
    if (c->tun.halo_stream_mode < 0) {
        // auto: time both schedules once on the first plain full-lattice application (idempotent: it only rewrites `out`).  Every rank
        // issues the same exchanges whichever schedule it ends up with, so the choice is local.
        if (s.parity_mode != 2 || s.upd_scal || s.upd[0] || s.upd[1]) {
            c->tun.halo_stream_mode = 0;
            const int st = stencil_apply(c, s);
            c->tun.halo_stream_mode = -1;
            return st;
        }
        float ms[3] = {0.f, 0.f, 0.f};
        for (int mode = 0; mode < 3; mode++) {
            c->tun.halo_stream_mode = mode;
            LQCHK(stencil_apply(c, s));
            HIPCHK(hipEventRecord(c->ev_tune0, c->stream));
            for (int k = 0; k < 4; k++) LQCHK(stencil_apply(c, s));
            HIPCHK(hipEventRecord(c->ev_tune1, c->stream));
            HIPCHK(hipEventSynchronize(c->ev_tune1));
            HIPCHK(hipEventElapsedTime(&ms[mode], c->ev_tune0, c->ev_tune1));
        }
        int best = 0;
        for (int mode = 1; mode < 3; mode++)
            if (ms[mode] < ms[best]) best = mode;
        c->tun.halo_stream_mode = best;
        for (int mode = 0; mode < 3; mode++) c->tun.halo_tuned_us[mode] = (int)(250.f * ms[mode]);
        return LQCD_OK;      // `out` holds the result of the last tuning application
    }
    if (c->tun.halo_stream_mode == 1) {
        // pack -> exchange -> exterior stay in order on the compute stream (no queue hop on the path that carries the messages);
        // the interior runs beside them on the second stream, forked and joined by events
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));                 // the inputs of this call are complete
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
        {
            hipStream_t main_stream = c->stream;
            c->stream = c->comm_stream;                                // the launchers enqueue on c->stream
            const int st = s.prec ? p32::launch_stencil_interior(c, s) : launch_stencil_interior(c, s);
            c->stream = main_stream;
            LQCHK(st);
        }
        HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
        LQCHK(s.prec ? p32::launch_stencil_pack(c, s) : launch_stencil_pack(c, s));
        LQCHK(halo_exchange_rccl(c, s.kind, s.parity_mode, s.prec, 1));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        return s.prec ? p32::launch_stencil_exterior(c, s) : launch_stencil_exterior(c, s);
    }
    if (c->tun.halo_stream_mode == 2) {
        // the interior is enqueued FIRST on the compute stream (the GPU starts it while the host is still busy issuing the RCCL group)
        // and stays in order with the exterior; pack -> exchange run on the second stream behind an event -- the schedule for an
        // exchange that is shorter than the interior: the fork / join latencies hide behind the interior kernel
        HIPCHK(hipEventRecord(c->ev_pack, c->stream));                 // the inputs of this call are complete
        HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
        LQCHK(s.prec ? p32::launch_stencil_interior(c, s) : launch_stencil_interior(c, s));
        {
            hipStream_t main_stream = c->stream;
            c->stream = c->comm_stream;
            const int st = s.prec ? p32::launch_stencil_pack(c, s) : launch_stencil_pack(c, s);
            c->stream = main_stream;
            LQCHK(st);
        }
        LQCHK(halo_exchange_rccl(c, s.kind, s.parity_mode, s.prec, 2));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        return s.prec ? p32::launch_stencil_exterior(c, s) : launch_stencil_exterior(c, s);
    }
    LQCHK(s.prec ? p32::launch_stencil_pack(c, s) : launch_stencil_pack(c, s));
    LQCHK(halo_exchange_rccl(c, s.kind, s.parity_mode, s.prec, 0));
    LQCHK(s.prec ? p32::launch_stencil_interior(c, s) : launch_stencil_interior(c, s));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    return s.prec ? p32::launch_stencil_exterior(c, s) : launch_stencil_exterior(c, s);
}

// ---------------------------------------------------------------------------------- operator -> stencil calls
static void fill_blocks(const double2* dst[2], lqcd_spinor_s* s) {
    dst[0] = s ? spinor_block(s, 0) : nullptr;
    dst[1] = s ? spinor_block(s, 1) : nullptr;
}

// opt-in 12-real links for the Wilson r = 1 split kernel (tunable gauge_recon = 12): lazily (re)built, used only when every
// link of the current field is unitary to 1e-14, otherwise the 18-real field is read as usual
static const double2* recon12_links(lqcd_op_s* op) {
    lqcd_ctx_s* c = op->ctx;
    c->tun.recon_active = 0;
    if (c->tun.gauge_recon != 12) return nullptr;
    if (op->kind == LQCD_WILSON && (op->r != 1.0 || (c->tun.dslash_variant != 1 && c->tun.dslash_variant < 4))) return nullptr;   // only the direction-split kernels
    if (op->kind == LQCD_STAGGERED && !(c->tun.dslash_variant >= 1 && c->tun.dslash_variant <= 5)) return nullptr;
    if (gauge_ensure_recon12(op->gauge) != LQCD_OK || !op->gauge->recon_ok) return nullptr;
    c->tun.recon_active = 1;
    return op->gauge->data12;
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
    s.gauge12 = recon12_links(op);
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
    s.gauge12 = recon12_links(op);
    return s;
}

void apply_bc(lqcd_ctx_s* c, const int bc[4]) {
    for (int mu = 0; mu < 4; mu++) {
        // unpartitioned direction: this rank owns both ends, a local wrap is a global wrap
        c->geom.bc_fwd[mu] = (double)bc[mu];
        c->geom.bc_bwd[mu] = (double)bc[mu];
    }
}

static int check_full(lqcd_op_s* op, lqcd_spinor_s* a, lqcd_spinor_s* b, const char* who) {
    if (!(op && a && b && a->ctx == op->ctx && b->ctx == op->ctx && a->kind == op->kind && b->kind == op->kind &&
          a->subset == LQCD_FULL && b->subset == LQCD_FULL && a != b)) {
        set_error(std::string(who) + ": need two distinct FULL spinors of the operator's kind on the operator's context");
        return LQCD_ERR_ARG;
    }
    return LQCD_OK;
}

int op_apply_async(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, double* norm_partial, const double* skip_flag) {
    apply_bc(op->ctx, op->bc);
    StencilCall s;
    LQCHK(make_full_call(op, out, in, dagger, s));
    s.norm_partial = norm_partial;
    s.skip_flag = skip_flag;
    return stencil_apply(op->ctx, s);
}

// ---------------------------------------------------------------------------------- CG with device-resident scalars

__global__ void cg_scalar_alpha(double* s) {
    if (s[S_DONE] != 0.0) { s[S_XDONE] = 1.0; return; }   // the iterate of the converging iteration has been written
    s[S_ALPHA] = s[S_RR] / s[S_PQ];
}
__global__ void cg_scalar_beta(double* s) {
    if (s[S_DONE] != 0.0) return;
    const double rrn = s[S_RRNEW];
    s[S_BETA] = rrn / s[S_RR];
    s[S_RR] = rrn;
    s[S_ITERS] += 1.0;
    if (rrn < s[S_EPS]) s[S_DONE] = 1.0;
}

constexpr int UB = 256;
// x += alpha p ; r -= alpha q ; partial |r|^2
__global__ __launch_bounds__(UB) void cg_update_xr(const double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ r,
                                                    const double2* __restrict__ p, const double2* __restrict__ q, size_t n,
                                                    double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    if (s[S_DONE] == 0.0) {
        const double al = s[S_ALPHA];
        for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
            const double2 pv = p[i], qv = q[i];
            double2 xv = x[i], rv = r[i];
            xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
            rv.x = fma(-al, qv.x, rv.x); rv.y = fma(-al, qv.y, rv.y);
            x[i] = xv; r[i] = rv;
            acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// fused tail of an iteration:  x += alpha p ;  p = r + beta p   (x is still updated in the iteration that converges,
// p only while the solve continues; nothing is touched in later, overshooting launches)
__global__ __launch_bounds__(UB) void cg_update_xp(const double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ p,
                                                    const double2* __restrict__ r, size_t n) {
    if (s[S_XDONE] != 0.0) return;
    const double al = s[S_ALPHA], be = s[S_BETA];
    const bool cont = s[S_DONE] == 0.0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        double2 pv = p[i], xv = x[i];
        xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
        x[i] = xv;
        if (cont) {
            const double2 rv = r[i];
            pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
            p[i] = pv;
        }
    }
}
// p = r + beta p
__global__ __launch_bounds__(UB) void cg_update_p(const double* __restrict__ s, double2* __restrict__ p, const double2* __restrict__ r, size_t n) {
    if (s[S_DONE] != 0.0) return;
    const double be = s[S_BETA];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 rv = r[i];
        double2 pv = p[i];
        pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
        p[i] = pv;
    }
}
__global__ __launch_bounds__(UB) void norm2_partial_kernel(const double2* __restrict__ a, size_t n, double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 v = a[i];
        acc = fma(v.x, v.x, acc); acc = fma(v.y, v.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// re(p.q) partials (unfused reference form c1 = p.q)
__global__ __launch_bounds__(UB) void redot_partial_kernel(const double2* __restrict__ a, const double2* __restrict__ b, size_t n, double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = a[i], y = b[i];
        acc = fma(x.x, y.x, acc); acc = fma(x.y, y.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}

struct CgWork {
    lqcd_spinor_s *r, *p, *q, *tmp;
};

// enqueue one CG iteration on the compute stream (no host synchronisation)
static int cg_enqueue_iteration(lqcd_op_s* op, lqcd_spinor_s* x, const CgWork& w) {
    lqcd_ctx_s* c = op->ctx;
    const size_t n = x->elems;
    if (c->tun.cg_fused >= 2) {
        // fully fused form: 10 spinor passes per iteration instead of 13, q = D^+ D p is never written
        //   tmp = D p [+ |tmp|^2 partials] ; alpha = rr / |tmp|^2 ; D^+ tmp with epilogue r -= alpha q [+ |r|^2 partials] ;
        //   beta, convergence ; x += alpha p, p = r + beta p
        const int nbs = stencil_num_partials(c, op->kind, op->r, 2);
        LQCHK(op_apply_async(op, w.tmp, w.p, 0, c->d_partial, c->tun.cg_skip_done ? c->d_scal : nullptr));   // a no-op once the solve has converged inside a burst
        LQCHK(reduce_to_slot(c, nbs, 1, S_PQ, true, 1));      // + alpha = rr / pq
        apply_bc(c, op->bc);
        StencilCall s2;
        LQCHK(make_full_call(op, w.q, w.tmp, 1, s2));
        s2.norm_partial = c->d_partial;
        s2.upd_scal = c->d_scal;
        s2.upd[0] = spinor_block(w.r, 0);
        s2.upd[1] = spinor_block(w.r, 1);
        LQCHK(stencil_apply(c, s2));
        LQCHK(reduce_to_slot(c, nbs, 1, S_RRNEW, true, 2));   // + beta, convergence flag
        const int nbu = stream_grid(c, n);
        hipLaunchKernelGGL(cg_update_xp, dim3(nbu), dim3(UB), 0, c->stream, c->d_scal, x->data, w.p->data, w.r->data, n);
        HIPCHK(hipGetLastError());
        return LQCD_OK;
    }
    const bool fuse = c->tun.cg_fused && !any_partitioned(c);
    // tmp = D p  (|tmp|^2 block partials fused into the stencil when the lattice is not partitioned)
    LQCHK(op_apply_async(op, w.tmp, w.p, 0, fuse ? c->d_partial : nullptr));
    int nb;
    if (fuse) {
        nb = stencil_num_blocks(c, op->kind, op->r, 2);
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    } else if (c->tun.cg_fused) {
        nb = stream_grid(c, n);
        hipLaunchKernelGGL(norm2_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, w.tmp->data, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    } else {
        // reference form: c1 = p . q
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
        nb = stream_grid(c, n);
        hipLaunchKernelGGL(redot_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, w.p->data, w.q->data, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
    }
    hipLaunchKernelGGL(cg_scalar_alpha, dim3(1), dim3(1), 0, c->stream, c->d_scal);
    nb = stream_grid(c, n);
    hipLaunchKernelGGL(cg_update_xr, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, x->data, w.r->data, w.p->data, w.q->data, n, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 1, S_RRNEW, true));
    hipLaunchKernelGGL(cg_scalar_beta, dim3(1), dim3(1), 0, c->stream, c->d_scal);
    hipLaunchKernelGGL(cg_update_p, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, w.p->data, w.r->data, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

static int cg_setup(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, CgWork& w, double eps, double* rr0) {
    lqcd_ctx_s* c = op->ctx;
    const size_t n = x->elems;
    // r = b - D^+ D x ; p = r
    LQCHK(op_apply_async(op, w.tmp, x, 0, nullptr));
    LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    HIPCHK(hipMemcpyAsync(w.r->data, b->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, w.q->data, w.r->data, n));
    HIPCHK(hipMemcpyAsync(w.p->data, w.r->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_norm2(c, w.r->data, n, rr0, true));
    double init[9] = {*rr0, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

int cg_run(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double eps, int maxiter, bool fixed, int* iters, double* final_rr) {
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    CgWork w;
    w.r = scratch_get(c, x->kind, LQCD_FULL);
    w.p = scratch_get(c, x->kind, LQCD_FULL);
    w.q = scratch_get(c, x->kind, LQCD_FULL);
    w.tmp = scratch_get(c, x->kind, LQCD_FULL);
    int st = LQCD_OK;
    double rr = 0;
    int it = 0;
    bool converged = false;
    if (!(w.r && w.p && w.q && w.tmp)) st = LQCD_ERR_HIP;
    if (st == LQCD_OK) st = cg_setup(op, x, b, w, fixed ? -1.0 : eps, &rr);
    if (st == LQCD_OK && !fixed && rr < eps) converged = true;
    const int check_every = 8;
    // tunable "graph": a burst of check_every iterations is captured once into a hipGraph and replayed -- one launch per
    // burst instead of 5 per iteration.  Pays on launch-bound (small) lattices; single-stream (unpartitioned) contexts only.
    const bool use_graph = c->tun.graph != 0 && !any_partitioned(c);
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    while (st == LQCD_OK && !converged && it < maxiter) {
        int burst = std::min(check_every, maxiter - it);
        if (fixed && !use_graph) burst = maxiter - it;
        if (use_graph && burst == check_every) {
            if (!gexec) {
                if (c->tun.halo_stream_mode < 0) c->tun.halo_stream_mode = 0;    // the auto-tuning pass synchronises: not inside a capture
                hipError_t ge = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
                if (ge != hipSuccess) { st = hip_fail(ge, "hipStreamBeginCapture", __FILE__, __LINE__); break; }
                for (int k = 0; k < burst && st == LQCD_OK; k++) st = cg_enqueue_iteration(op, x, w);
                ge = hipStreamEndCapture(c->stream, &graph);
                if (st == LQCD_OK && ge != hipSuccess) st = hip_fail(ge, "hipStreamEndCapture", __FILE__, __LINE__);
                if (st == LQCD_OK) {
                    ge = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
                    if (ge != hipSuccess) st = hip_fail(ge, "hipGraphInstantiate", __FILE__, __LINE__);
                }
                if (st != LQCD_OK) break;
            }
            hipError_t ge = hipGraphLaunch(gexec, c->stream);
            if (ge != hipSuccess) { st = hip_fail(ge, "hipGraphLaunch", __FILE__, __LINE__); break; }
        } else {
            for (int k = 0; k < burst && st == LQCD_OK; k++) st = cg_enqueue_iteration(op, x, w);
        }
        if (st != LQCD_OK) break;
        if (fixed && use_graph && it + burst < maxiter) { it += burst; continue; }   // timing window: no readback between bursts
        hipError_t e = hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { st = hip_fail(e, "cg scalar readback", __FILE__, __LINE__); break; }
        rr = c->h_scal[S_RR - S_RR];
        it = (int)c->h_scal[S_ITERS - S_RR];
        if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
        if (!std::isfinite(rr)) { set_error("CG: residual is not finite"); st = LQCD_ERR_NOT_CONVERGED; }
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    scratch_put(w.r); scratch_put(w.p); scratch_put(w.q); scratch_put(w.tmp);
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (st != LQCD_OK) return st;
    if (!fixed && !converged) {
        set_error("The CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- BiCGStab (device-resident scalars)
// One iteration = 2 operator applications + 5 streaming kernels + 4 single-block reductions, all enqueued without a host
// round trip; complex alpha/omega/beta live in d_scal[B_*] (scalar steps: blas.hip cg_scalar_step ops 3..6).  The host
// polls the done flag every few iterations.  Same recurrences, stopping rule (|s|^2 < eps half-step exit, |r|^2 < eps) and
// iteration count as the textbook van der Vorst loop the parity tests compare against.
typedef std::function<int(double2* out, const double2* in)> ApplyFn;

template <int NV>
__device__ inline void block_reduce_nv(double (&a)[NV], double* partial) {
    __shared__ double red[NV][UB / 64];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[v] += __shfl_down(a[v], off, 64);
        if ((threadIdx.x & 63) == 0) red[v][threadIdx.x >> 6] = a[v];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < UB / 64; w++) t += red[threadIdx.x][w];
        partial[blockIdx.x * NV + threadIdx.x] = t;
    }
}
// <a,b> = sum conj(a) b
__global__ __launch_bounds__(UB) void bicg_dot(const double* __restrict__ sc, const double2* __restrict__ a, const double2* __restrict__ b, size_t n,
                                                double* partial) {
    if (sc[B_DONE] != 0.0) return;
    double acc[2] = {0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = a[i], y = b[i];
        acc[0] = fma(x.x, y.x, acc[0]); acc[0] = fma(x.y, y.y, acc[0]);
        acc[1] = fma(x.x, y.y, acc[1]); acc[1] = fma(-x.y, y.x, acc[1]);
    }
    block_reduce_nv<2>(acc, partial);
}
// s = r - alpha v ; partial |s|^2
__global__ __launch_bounds__(UB) void bicg_s(const double* __restrict__ sc, double2* __restrict__ s, const double2* __restrict__ r,
                                              const double2* __restrict__ v, size_t n, double* partial) {
    if (sc[B_DONE] != 0.0) return;
    const double ar = sc[B_ALPHA], ai = sc[B_ALPHA + 1];
    double acc[1] = {0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 vv = v[i];
        double2 sv = r[i];
        sv.x = fma(-ar, vv.x, sv.x); sv.x = fma(ai, vv.y, sv.x);
        sv.y = fma(-ar, vv.y, sv.y); sv.y = fma(-ai, vv.x, sv.y);
        s[i] = sv;
        acc[0] = fma(sv.x, sv.x, acc[0]); acc[0] = fma(sv.y, sv.y, acc[0]);
    }
    block_reduce_nv<1>(acc, partial);
}
// partials of <t,s> (2 values) and |t|^2
__global__ __launch_bounds__(UB) void bicg_ts(const double* __restrict__ sc, const double2* __restrict__ t, const double2* __restrict__ s, size_t n,
                                               double* partial) {
    if (sc[B_DONE] != 0.0) return;
    double acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = t[i], y = s[i];
        acc[0] = fma(x.x, y.x, acc[0]); acc[0] = fma(x.y, y.y, acc[0]);
        acc[1] = fma(x.x, y.y, acc[1]); acc[1] = fma(-x.y, y.x, acc[1]);
        acc[2] = fma(x.x, x.x, acc[2]); acc[2] = fma(x.y, x.y, acc[2]);
    }
    block_reduce_nv<3>(acc, partial);
}
// x += alpha p + omega s ; r = s - omega t ; partials |r|^2, <r0,r>
__global__ __launch_bounds__(UB) void bicg_xr(const double* __restrict__ sc, double2* __restrict__ x, double2* __restrict__ r,
                                               const double2* __restrict__ p, const double2* __restrict__ s, const double2* __restrict__ t,
                                               const double2* __restrict__ r0, size_t n, double* partial) {
    if (sc[B_DONE] != 0.0) return;
    const double ar = sc[B_ALPHA], ai = sc[B_ALPHA + 1], wr = sc[B_OMEGA], wi = sc[B_OMEGA + 1];
    double acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 pv = p[i], sv = s[i], tv = t[i], zv = r0[i];
        double2 xv = x[i], rv = sv;
        xv.x = fma(ar, pv.x, xv.x); xv.x = fma(-ai, pv.y, xv.x);
        xv.y = fma(ar, pv.y, xv.y); xv.y = fma(ai, pv.x, xv.y);
        xv.x = fma(wr, sv.x, xv.x); xv.x = fma(-wi, sv.y, xv.x);
        xv.y = fma(wr, sv.y, xv.y); xv.y = fma(wi, sv.x, xv.y);
        rv.x = fma(-wr, tv.x, rv.x); rv.x = fma(wi, tv.y, rv.x);
        rv.y = fma(-wr, tv.y, rv.y); rv.y = fma(-wi, tv.x, rv.y);
        x[i] = xv; r[i] = rv;
        acc[0] = fma(rv.x, rv.x, acc[0]); acc[0] = fma(rv.y, rv.y, acc[0]);
        acc[1] = fma(zv.x, rv.x, acc[1]); acc[1] = fma(zv.y, rv.y, acc[1]);
        acc[2] = fma(zv.x, rv.y, acc[2]); acc[2] = fma(-zv.y, rv.x, acc[2]);
    }
    block_reduce_nv<3>(acc, partial);
}
// p = r + beta (p - omega v)
__global__ __launch_bounds__(UB) void bicg_p(const double* __restrict__ sc, double2* __restrict__ p, const double2* __restrict__ r,
                                              const double2* __restrict__ v, size_t n) {
    if (sc[B_DONE] != 0.0) return;
    const double br = sc[B_BETA], bi = sc[B_BETA + 1], wr = sc[B_OMEGA], wi = sc[B_OMEGA + 1];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 vv = v[i], rv = r[i];
        double2 pv = p[i];
        pv.x = fma(-wr, vv.x, pv.x); pv.x = fma(wi, vv.y, pv.x);
        pv.y = fma(-wr, vv.y, pv.y); pv.y = fma(-wi, vv.x, pv.y);
        double2 o;
        o.x = fma(br, pv.x, rv.x); o.x = fma(-bi, pv.y, o.x);
        o.y = fma(br, pv.y, rv.y); o.y = fma(bi, pv.x, o.y);
        p[i] = o;
    }
}

static int bicgstab_core(lqcd_ctx_s* c, const ApplyFn& A, size_t n, double2* x, const double2* b, double2* const w[6], double eps,
                         int maxiter, int* iters, double* final_rr) {
    double2 *r = w[0], *r0 = w[1], *p = w[2], *v = w[3], *s = w[4], *t = w[5];
    const size_t bytes = n * sizeof(double2);
    LQCHK(A(v, x));
    HIPCHK(hipMemcpyAsync(r, b, bytes, hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, v, r, n));
    HIPCHK(hipMemcpyAsync(r0, r, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(p, r, bytes, hipMemcpyDeviceToDevice, c->stream));
    double rr;
    LQCHK(blas_norm2(c, r, n, &rr, true));
    double init[B_END - B_RHO] = {0};
    init[B_RHO - B_RHO] = rr;
    init[B_EPS - B_RHO] = eps;
    init[B_RES - B_RHO] = rr;
    HIPCHK(hipMemcpyAsync(c->d_scal + B_RHO, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int it = 0, st = LQCD_ERR_NOT_CONVERGED;
    bool breakdown = false;
    if (rr < eps) st = LQCD_OK;
    const int nb = stream_grid(c, n), check_every = 4;
    const double* sc = c->d_scal;
    while (st != LQCD_OK && !breakdown && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            LQCHK(A(v, p));
            hipLaunchKernelGGL(bicg_dot, dim3(nb), dim3(UB), 0, c->stream, sc, r0, v, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 2, B_R0V, true, 3));
            hipLaunchKernelGGL(bicg_s, dim3(nb), dim3(UB), 0, c->stream, sc, s, r, v, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 1, B_SS, true, 4));
            LQCHK(A(t, s));
            hipLaunchKernelGGL(bicg_ts, dim3(nb), dim3(UB), 0, c->stream, sc, t, s, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 3, B_TS, true, 5));
            hipLaunchKernelGGL(bicg_xr, dim3(nb), dim3(UB), 0, c->stream, sc, x, r, p, s, t, r0, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 3, B_RR, true, 6));
            hipLaunchKernelGGL(bicg_p, dim3(nb), dim3(UB), 0, c->stream, sc, p, r, v, n);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + B_RHO, (B_END - B_RHO) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        it = (int)c->h_scal[B_ITERS - B_RHO];
        rr = c->h_scal[B_RES - B_RHO];
        const double done = c->h_scal[B_DONE - B_RHO];
        if (done == 1.0) st = LQCD_OK;
        else if (done != 0.0) breakdown = true;
    }
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (breakdown) { set_error("BiCGStab: residual is not finite (breakdown)"); return LQCD_ERR_NOT_CONVERGED; }
    if (st != LQCD_OK) {
        set_error("The BiCGStab is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

// ---------------------------------------------------------------------------------- C API: operator
extern "C" int lqcd_op_create(lqcd_ctx_t ctx, lqcd_op_t* op, int kind, lqcd_gauge_t g, double km, double r, const int bc[4]) {
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
    delete op;
    return LQCD_OK;
}

// Dirac_operator = "WilsonClover", Clover_coefficient (parameter_structs.jl:125; test/test_wilsonclover.toml:9): D_sw = D + (A - 1),
// A = 1 + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu} (clover.hip).  csw = 0 switches the term off again.
extern "C" int lqcd_op_set_clover(lqcd_op_t op, double csw) {
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
    op->clover_version = 0;   // another field: the clover term is rebuilt at the next application
    op->clover_inv_version = 0;
    return LQCD_OK;
}

extern "C" int lqcd_op_apply(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger) {
    LQCHK(check_full(op, out, in, "lqcd_op_apply"));
    LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return LQCD_OK;
}

extern "C" int lqcd_op_apply_DdagD(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in) {
    LQCHK(check_full(op, out, in, "lqcd_op_apply_DdagD"));
    lqcd_spinor_s* tmp = scratch_get(op->ctx, op->kind, LQCD_FULL);
    if (!tmp) return LQCD_ERR_HIP;
    int st = op_apply_async(op, tmp, in, 0, nullptr);
    if (st == LQCD_OK) st = op_apply_async(op, out, tmp, 1, nullptr);
    hipError_t e = hipStreamSynchronize(op->ctx->stream);
    scratch_put(tmp);
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync DdagD", __FILE__, __LINE__);
    return st;
}

extern "C" int lqcd_op_hop(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger) {
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

// ---------------------------------------------------------------------------------- C API: solvers
extern "C" int lqcd_solve_cg_DdagD(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, double eps, int maxiter, int* iters, double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD"));
    ARGCHK(maxiter >= 0, "lqcd_solve_cg_DdagD: maxiter < 0");
    return cg_run(op, x, b, eps, maxiter, false, iters, final_rr);
}
extern "C" int lqcd_solve_cg_DdagD_fixed(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int niter) {
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD_fixed"));
    return cg_run(op, x, b, 0.0, niter, true, nullptr, nullptr);
}

// scratch fields of one call: returned to the context's pool on every exit path
struct ScratchScope {
    lqcd_ctx_s* c;
    std::vector<lqcd_spinor_s*> held;
    explicit ScratchScope(lqcd_ctx_s* c_) : c(c_) {}
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
    lqcd_spinor_s* get(int kind, int subset) {
        lqcd_spinor_s* s = scratch_get(c, kind, subset);
        if (s) held.push_back(s);
        return s;
    }
    ~ScratchScope() { for (lqcd_spinor_s* s : held) scratch_put(s); }
};

// CG with device-resident scalars for a Hermitian positive operator given as an enqueue function (reference form:
// alpha = rr / <p, A p>); x holds the initial guess, work = three fields of n elements.  Used where the fused full-lattice
// iteration of cg_run does not apply (parity blocks).
static int cg_generic(lqcd_ctx_s* c, const ApplyFn& A, size_t n, double2* x, const double2* b, double2* r, double2* p, double2* q, double eps,
                      int maxiter, int* iters, double* final_rr) {
    LQCHK(A(q, x));
    HIPCHK(hipMemcpyAsync(r, b, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, q, r, n));
    HIPCHK(hipMemcpyAsync(p, r, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    double rr = 0;
    LQCHK(blas_norm2(c, r, n, &rr, true));
    double init[9] = {rr, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int it = 0;
    bool converged = rr < eps;
    const int nb = stream_grid(c, n), check_every = 8;
    while (!converged && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            LQCHK(A(q, p));
            hipLaunchKernelGGL(redot_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, p, q, n, c->d_partial);
            HIPCHK(hipGetLastError());
            LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
            hipLaunchKernelGGL(cg_scalar_alpha, dim3(1), dim3(1), 0, c->stream, c->d_scal);
            hipLaunchKernelGGL(cg_update_xr, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, x, r, p, q, n, c->d_partial);
            HIPCHK(hipGetLastError());
            LQCHK(reduce_to_slot(c, nb, 1, S_RRNEW, true));
            hipLaunchKernelGGL(cg_scalar_beta, dim3(1), dim3(1), 0, c->stream, c->d_scal);
            hipLaunchKernelGGL(cg_update_p, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, p, r, n);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        it = (int)c->h_scal[S_ITERS - S_RR];
        if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
        if (!std::isfinite(rr)) { set_error("CG: residual is not finite"); return LQCD_ERR_NOT_CONVERGED; }
    }
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (!converged) {
        set_error("The CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// Staggered D^+D = m^2 - H^2 is block diagonal in parity: (D^+D)_pp = m^2 - H_pq H_qp.  Solves that block for the parity-p halves of
// the FULL fields x (initial guess / solution) and b with half-lattice vectors -- one Dslash-equivalent per iteration instead of
// two.  The other parity of x is not touched.  This is the solve behind the reference's "4 tastes" (Nf = 4) staggered action, whose
// pseudofermion lives on the even sites (test/test_staggered.toml).
extern "C" int lqcd_solve_cg_DdagD_parity(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int parity, double eps, int maxiter, int* iters,
                                          double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD_parity"));
    ARGCHK(op->kind == LQCD_STAGGERED && (parity == 0 || parity == 1) && maxiter >= 0,
           "lqcd_solve_cg_DdagD_parity: staggered operators only, parity 0 (even) or 1 (odd)");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    apply_bc(c, op->bc);
    const size_t nh = x->elems / 2;
    const int mine = parity ? LQCD_ODD : LQCD_EVEN, other = parity ? LQCD_EVEN : LQCD_ODD;
    ScratchScope pool(c);
    lqcd_spinor_s *r = pool.get(op->kind, mine), *p = pool.get(op->kind, mine), *q = pool.get(op->kind, mine), *t = pool.get(op->kind, other);
    if (!(r && p && q && t)) return LQCD_ERR_HIP;
    lqcd_spinor_s xv = *x, bv = *b;
    xv.subset = bv.subset = mine;
    xv.elems = bv.elems = nh;
    xv.data = x->data + (size_t)parity * nh;
    bv.data = b->data + (size_t)parity * nh;
    lqcd_spinor_s vin = xv, vout = xv;
    const double m2 = op->km * op->km;
    ApplyFn A = [&](double2* out, const double2* in) -> int {
        vin.data = const_cast<double2*>(in);
        vout.data = out;
        StencilCall s1 = make_hop_call(op, t, &vin, nullptr, 0.0, 1.0, 0);          // t = H in (other parity)
        LQCHK(stencil_apply(c, s1));
        StencilCall s2 = make_hop_call(op, &vout, t, &vin, m2, -1.0, 0);            // out = m^2 in - H t
        return stencil_apply(c, s2);
    };
    return cg_generic(c, A, nh, xv.data, bv.data, r->data, p->data, q->data, eps, maxiter, iters, final_rr);
}

extern "C" int lqcd_solve_bicgstab(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters,
                                   double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_bicgstab"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    ScratchScope pool(c);
    double2* wd[6];
    for (int i = 0; i < 6; i++) {
        lqcd_spinor_s* wi = pool.get(op->kind, LQCD_FULL);
        if (!wi) return LQCD_ERR_HIP;
        wd[i] = wi->data;
    }
    // the stencil works on spinor handles; wrap raw pointers of the scratch fields
    lqcd_spinor_s vin = *x, vout = *x;
    ApplyFn A = [&](double2* out, const double2* in) -> int {
        vin.data = const_cast<double2*>(in);
        vout.data = out;
        return op_apply_async(op, &vout, &vin, dagger ? 1 : 0, nullptr);
    };
    return bicgstab_core(c, A, x->elems, x->data, b->data, wd, eps, maxiter, iters, final_rr);
}

// even-odd (Schur) preconditioned BiCGStab, Wilson:
//   (1 - k^2 H_eo H_oe) x_e = b_e + k H_eo b_o ;  x_o = b_o + k H_oe x_e
// Wilson-clover (D_sw = A - k H, A block diagonal in parity): with the packed inverse blocks A^-1 (clover.hip)
//   (1 - k^2 A_ee^-1 H_eo A_oo^-1 H_oe) x_e = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o) ;  x_o = A_oo^-1 (b_o + k H_oe x_e)
// (D_sw^+: H -> H^+ through the dagger flag of the hop, A is Hermitian).
extern "C" int lqcd_solve_bicgstab_eo(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters,
                                      double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_bicgstab_eo"));
    ARGCHK(op->kind == LQCD_WILSON, "lqcd_solve_bicgstab_eo: Wilson only");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    apply_bc(c, op->bc);
    const bool clov = op->csw != 0.0 && op->clover;
    if (clov) {     // A follows the links, A^-1 follows A
        if (op->clover_version != op->gauge->version) {
            LQCHK(clover_build(c, op->gauge, op->clover, op->km, op->csw));
            op->clover_version = op->gauge->version;
        }
        if (!op->clover_inv) HIPCHK(hipMalloc((void**)&op->clover_inv, clover_elems(c->geom) * sizeof(double2)));
        if (op->clover_inv_version != op->clover_version) {
            LQCHK(clover_invert(c, op->clover, op->clover_inv));
            op->clover_inv_version = op->clover_version;
        }
    }
    const double2* Ai = op->clover_inv;
    const double k = op->km;
    const int dg = dagger ? 1 : 0;
    const size_t nh = x->elems / 2;
    ScratchScope pool(c);
    double2* wd[6];
    for (int i = 0; i < 6; i++) {
        lqcd_spinor_s* wi = pool.get(op->kind, LQCD_EVEN);
        if (!wi) return LQCD_ERR_HIP;
        wd[i] = wi->data;
    }
    lqcd_spinor_s* rhs = pool.get(op->kind, LQCD_EVEN);
    lqcd_spinor_s* te = clov ? pool.get(op->kind, LQCD_EVEN) : nullptr;
    lqcd_spinor_s* to = pool.get(op->kind, LQCD_ODD);
    lqcd_spinor_s* uo = clov ? pool.get(op->kind, LQCD_ODD) : nullptr;
    if (!rhs || !to || (clov && (!te || !uo))) return LQCD_ERR_HIP;
    // views of the even/odd halves of b and x
    lqcd_spinor_s be = *b, bo = *b, xe = *x, xo = *x;
    be.subset = xe.subset = LQCD_EVEN; bo.subset = xo.subset = LQCD_ODD;
    be.elems = bo.elems = xe.elems = xo.elems = nh;
    bo.data = b->data + nh; xo.data = x->data + nh;
    int st = LQCD_OK;
    auto run = [&]() -> int {
        lqcd_spinor_s vin = xe, vout = xe;
        ApplyFn A;
        if (!clov) {
            // rhs = b_e + k H_eo b_o
            { StencilCall s = make_hop_call(op, rhs, &bo, &be, 1.0, k, dg); LQCHK(stencil_apply(c, s)); }
            A = [&](double2* out, const double2* in) -> int {
                vin.data = const_cast<double2*>(in);
                vout.data = out;
                StencilCall s1 = make_hop_call(op, to, &vin, nullptr, 0.0, 1.0, dg);         // t_o = H_oe in
                LQCHK(stencil_apply(c, s1));
                StencilCall s2 = make_hop_call(op, &vout, to, &vin, 1.0, -k * k, dg);        // out = in - k^2 H_eo t_o
                return stencil_apply(c, s2);
            };
        } else {
            // rhs = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o)
            LQCHK(clover_apply_parity(c, Ai, 1, uo->data, bo.data, 1.0, nullptr, 0.0));
            { StencilCall s = make_hop_call(op, te, uo, &be, 1.0, k, dg); LQCHK(stencil_apply(c, s)); }
            LQCHK(clover_apply_parity(c, Ai, 0, rhs->data, te->data, 1.0, nullptr, 0.0));
            A = [&](double2* out, const double2* in) -> int {
                vin.data = const_cast<double2*>(in);
                StencilCall s1 = make_hop_call(op, to, &vin, nullptr, 0.0, 1.0, dg);         // t_o = H_oe in
                LQCHK(stencil_apply(c, s1));
                LQCHK(clover_apply_parity(c, Ai, 1, uo->data, to->data, 1.0, nullptr, 0.0)); // u_o = A_oo^-1 t_o
                StencilCall s2 = make_hop_call(op, te, uo, nullptr, 0.0, 1.0, dg);           // t_e = H_eo u_o
                LQCHK(stencil_apply(c, s2));
                return clover_apply_parity(c, Ai, 0, out, te->data, -k * k, in, 1.0);        // out = in - k^2 A_ee^-1 t_e
            };
        }
        const int sc = bicgstab_core(c, A, nh, xe.data, rhs->data, wd, eps, maxiter, iters, final_rr);
        // the odd half (also on non-convergence, so x is a consistent best effort)
        if (!clov) {
            StencilCall s = make_hop_call(op, &xo, &xe, &bo, 1.0, k, dg);                    // x_o = b_o + k H_oe x_e
            LQCHK(stencil_apply(c, s));
        } else {
            StencilCall s = make_hop_call(op, to, &xe, &bo, 1.0, k, dg);
            LQCHK(stencil_apply(c, s));
            LQCHK(clover_apply_parity(c, Ai, 1, xo.data, to->data, 1.0, nullptr, 0.0));      // x_o = A_oo^-1 (b_o + k H_oe x_e)
        }
        return sc;
    };
    st = run();
    hipError_t e = hipStreamSynchronize(c->stream);      // before the scratch fields go back to the pool
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync bicgstab_eo", __FILE__, __LINE__);
    return st;
}

// ---------------------------------------------------------------------------------- multi-shift CG (RHMC solver)
namespace lqcd {
// per-shift coefficient block in device memory: [sigma | zeta_{n-1} | zeta_n | a | b | z] (ns doubles each), alpha_{n-1}, beta_{n-1}
// zeta recurrence (Jegerlehner hep-lat/9612014) after the base system's alpha_n, beta_n are known:
//   zeta_{n+1} = zeta_n zeta_{n-1} alpha_{n-1} / (zeta_{n-1} alpha_{n-1} (1 + alpha_n sigma) + alpha_n beta_{n-1} (zeta_{n-1} - zeta_n))
//   x_j += (zeta_{n+1}/zeta_n) alpha_n p_j ;  p_j = (zeta_{n+1}/zeta_n)^2 beta_n p_j + zeta_{n+1} r
__global__ void ms_zeta(const double* __restrict__ sc, double* __restrict__ ms, int ns) {
    if (sc[S_XDONE] != 0.0) return;
    const double alpha = sc[S_ALPHA], beta = sc[S_BETA], alpha_m = ms[6 * ns], beta_m = ms[6 * ns + 1];
    for (int j = threadIdx.x; j < ns; j += blockDim.x) {
        const double sigma = ms[j], zm = ms[ns + j], z0 = ms[2 * ns + j];
        if (fabs(z0) < 1e-100) {      // this shift converged long ago (its residual is zeta^2 |r|^2): freeze it before zeta underflows to 0/0
            ms[3 * ns + j] = 0.0; ms[4 * ns + j] = 0.0; ms[5 * ns + j] = 0.0;
            continue;
        }
        const double den = zm * alpha_m * (1.0 + alpha * sigma) + alpha * beta_m * (zm - z0);
        const double zp = z0 * zm * alpha_m / den, ratio = zp / z0;
        ms[3 * ns + j] = ratio * alpha;
        if (zp * zp * sc[S_RR] < sc[S_EPS]) {
            // the residual of this shift, zeta^2 |r|^2, is below the target once x_j has taken this step: last update, then the
            // shift is frozen (p_j = 0, no further traffic) -- large shifts drop out after a few tens of iterations
            ms[4 * ns + j] = 0.0; ms[5 * ns + j] = 0.0; ms[ns + j] = 0.0; ms[2 * ns + j] = 0.0;
            continue;
        }
        ms[4 * ns + j] = ratio * ratio * beta;
        ms[5 * ns + j] = zp;
        ms[ns + j] = z0;
        ms[2 * ns + j] = zp;
    }
    __syncthreads();
    if (threadIdx.x == 0) { ms[6 * ns] = alpha; ms[6 * ns + 1] = beta; }
}
// base system (x += alpha p ; p = r + beta p) and every active shifted system j (x_j += a_j p_j ; p_j = b_j p_j + z_j r) in one pass:
// r is read once per element, frozen shifts cost nothing.  x is still updated in the iteration that converges; nothing is touched
// afterwards.
__global__ __launch_bounds__(UB) void ms_update_all(const double* __restrict__ sc, const double* __restrict__ ms, double2* const* __restrict__ ptr,
                                                     double2* __restrict__ x0, double2* __restrict__ p0, const double2* __restrict__ r, size_t n,
                                                     int ns) {
    if (sc[S_XDONE] != 0.0) return;
    const double al = sc[S_ALPHA], be = sc[S_BETA];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 rv = r[i];
        {
            double2 pv = p0[i];
            if (x0) {          // the unshifted solution is optional (a rational action only wants the shifted ones)
                double2 xv = x0[i];
                xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
                x0[i] = xv;
            }
            pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
            p0[i] = pv;
        }
        for (int j = 0; j < ns; j++) {
            const double a = ms[3 * ns + j], bb = ms[4 * ns + j], z = ms[5 * ns + j];
            if (a == 0.0 && bb == 0.0 && z == 0.0) continue;       // frozen shift
            double2* __restrict__ x = ptr[j];
            double2* __restrict__ p = ptr[ns + j];
            double2 pv = p[i], xv = x[i];
            xv.x = fma(a, pv.x, xv.x); xv.y = fma(a, pv.y, xv.y);
            pv.x = fma(bb, pv.x, z * rv.x); pv.y = fma(bb, pv.y, z * rv.y);
            x[i] = xv; p[i] = pv;
        }
    }
}
}  // namespace lqcd

// (D^+D + sigma_j) x_j = b for all j < ns, plus the unshifted solution x0 (may be NULL): one Krylov space, the shifted
// iterates follow from the zeta recurrences, which run on the device next to the CG scalars (no host round trip inside an
// iteration; the host polls the convergence flag every 8 iterations).  Zero initial guesses.  Stops when |r|^2 < eps
// (for sigma_j >= 0 every |zeta_j| <= 1, so the shifted residuals zeta_j r are then below eps as well).
extern "C" int lqcd_solve_multishift_cg(lqcd_op_t op, lqcd_spinor_t x0, lqcd_spinor_t* xs, lqcd_spinor_t b, const double* sigma, int ns,
                                        double eps, int maxiter, int* iters, double* final_rr) {
    ARGCHK(op && b && ns >= 0 && ns <= 1024 && (ns == 0 || (xs && sigma)), "lqcd_solve_multishift_cg: null argument or more than 1024 shifts");
    ARGCHK(b->ctx == op->ctx && b->kind == op->kind && b->subset == LQCD_FULL, "lqcd_solve_multishift_cg: b must be a FULL spinor of the operator");
    for (int j = 0; j < ns; j++) {
        ARGCHK(xs[j] && xs[j]->ctx == op->ctx && xs[j]->kind == op->kind && xs[j]->subset == LQCD_FULL && xs[j] != b,
               "lqcd_solve_multishift_cg: xs[j] must be distinct FULL spinors of the operator");
        ARGCHK(sigma[j] >= 0.0, "lqcd_solve_multishift_cg: shifts must be non-negative");
    }
    if (x0) LQCHK(check_full(op, x0, b, "lqcd_solve_multishift_cg"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t n = b->elems, bytes = n * sizeof(double2);
    lqcd_spinor_s* xbase = x0;          // may stay null: then the base system only drives the Krylov space
    lqcd_spinor_s* r = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* p = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* q = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* tmp = scratch_get(c, op->kind, LQCD_FULL);
    std::vector<lqcd_spinor_s*> ps(ns, nullptr);
    bool ok = r && p && q && tmp;
    for (int j = 0; j < ns && ok; j++) { ps[j] = scratch_get(c, op->kind, LQCD_FULL); ok = ps[j] != nullptr; }
    const size_t ms_doubles = 6 * (size_t)ns + 2, ms_bytes = ms_doubles * sizeof(double) + 2 * (size_t)ns * sizeof(double2*);
    char* d_blk = nullptr;
    if (ok && hipMalloc((void**)&d_blk, ms_bytes) != hipSuccess) ok = false;
    auto release = [&]() {
        scratch_put(r); scratch_put(p); scratch_put(q); scratch_put(tmp);
        for (auto* s : ps) scratch_put(s);
        if (d_blk) (void)hipFree(d_blk);
    };
    if (!ok) { release(); set_error("lqcd_solve_multishift_cg: out of device memory"); return LQCD_ERR_HIP; }
    double* d_ms = (double*)d_blk;
    double2** d_ptr = (double2**)(d_blk + ms_doubles * sizeof(double));
    auto run = [&]() -> int {
        if (xbase) HIPCHK(hipMemsetAsync(xbase->data, 0, bytes, c->stream));
        HIPCHK(hipMemcpyAsync(r->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(p->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
        std::vector<double> hms(ms_doubles, 1.0);    // zeta_{-1} = zeta_0 = 1, alpha_{-1} = 1
        std::vector<double2*> hptr(2 * (size_t)ns);
        for (int j = 0; j < ns; j++) {
            HIPCHK(hipMemsetAsync(xs[j]->data, 0, bytes, c->stream));
            HIPCHK(hipMemcpyAsync(ps[j]->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
            hms[j] = sigma[j];
            hptr[j] = xs[j]->data;
            hptr[ns + j] = ps[j]->data;
        }
        hms[6 * (size_t)ns + 1] = 0.0;               // beta_{-1} = 0
        HIPCHK(hipMemcpyAsync(d_ms, hms.data(), ms_doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (ns) HIPCHK(hipMemcpyAsync(d_ptr, hptr.data(), 2 * (size_t)ns * sizeof(double2*), hipMemcpyHostToDevice, c->stream));
        double rr = 0.0;
        LQCHK(blas_norm2(c, r->data, n, &rr, true));
        double init[9] = {rr, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
        HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        int it = 0;
        bool converged = rr < eps;
        const int nbs = stencil_num_partials(c, op->kind, op->r, 2), nbu = stream_grid(c, n), check_every = 8;
        while (!converged && it < maxiter) {
            const int burst = std::min(check_every, maxiter - it);
            for (int k = 0; k < burst; k++) {
                // tmp = D p, alpha = rr / |tmp|^2 ; r -= alpha D^+ tmp in the stencil epilogue, beta = rr'/rr
                LQCHK(op_apply_async(op, tmp, p, 0, c->d_partial, c->d_scal));
                LQCHK(reduce_to_slot(c, nbs, 1, S_PQ, true, 1));
                apply_bc(c, op->bc);
                StencilCall s2;
                LQCHK(make_full_call(op, q, tmp, 1, s2));
                s2.norm_partial = c->d_partial;
                s2.upd_scal = c->d_scal;
                s2.upd[0] = spinor_block(r, 0);
                s2.upd[1] = spinor_block(r, 1);
                LQCHK(stencil_apply(c, s2));
                LQCHK(reduce_to_slot(c, nbs, 1, S_RRNEW, true, 2));
                if (ns) hipLaunchKernelGGL(ms_zeta, dim3(1), dim3(64), 0, c->stream, c->d_scal, d_ms, ns);
                hipLaunchKernelGGL(ms_update_all, dim3(nbu), dim3(UB), 0, c->stream, c->d_scal, d_ms, d_ptr, xbase ? xbase->data : (double2*)nullptr, p->data,
                                   r->data, n, ns);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            rr = c->h_scal[S_RR - S_RR];
            it = (int)c->h_scal[S_ITERS - S_RR];
            if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
            if (!std::isfinite(rr)) { set_error("multi-shift CG: residual is not finite"); break; }
        }
        if (iters) *iters = it;
        if (final_rr) *final_rr = rr;
        if (!converged) {
            if (std::isfinite(rr))
                set_error("The shifted CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
            return LQCD_ERR_NOT_CONVERGED;
        }
        return LQCD_OK;
    };
    const int st = run();
    (void)hipStreamSynchronize(c->stream);
    release();
    return st;
}

// ---------------------------------------------------------------------------------- pseudofermion action and force
// (evaluate_FermiAction standardHMC.jl:71, calc_UdSfdU! AbstractMD.jl:129)
static int mdom_check(int n, lqcd_ctx_s* c0);
static int force_check(lqcd_op_s* op, const char* who) {
    if (any_partitioned(op->ctx) && !op->ctx->local_peers.empty()) {
        set_error(std::string(who) + ": this context belongs to an in-process PE grid: use lqcd_mdom_fermion_force");
        return LQCD_ERR_ARG;
    }
    return LQCD_OK;
}

// S_f = eta^+ (D^+D)^-1 eta by CG from a zero guess; X = (D^+D)^-1 eta is returned, Y = D X if Y != NULL
extern "C" int lqcd_fermi_action(lqcd_op_t op, lqcd_spinor_t eta, lqcd_spinor_t X, lqcd_spinor_t Y, double eps, int maxiter, double* Sf,
                                 int* iters) {
    LQCHK(check_full(op, X, eta, "lqcd_fermi_action"));
    if (Y) LQCHK(check_full(op, Y, eta, "lqcd_fermi_action"));
    ARGCHK(X != eta && Y != eta && X != Y, "lqcd_fermi_action: eta, X and Y must be distinct fields");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemsetAsync(X->data, 0, X->elems * sizeof(double2), c->stream));
    bool even_only = false;
    if (op->kind == LQCD_STAGGERED && c->tun.staggered_parity_solve) {
        // a pseudofermion that lives on the even sites (the reference's 4-taste action): D^+D is block diagonal in parity, X stays
        // on the even sites and the half-lattice CG does the same solve at half the cost
        double odd2 = 0;
        LQCHK(blas_norm2(c, eta->data + eta->elems / 2, eta->elems / 2, &odd2, true));
        even_only = odd2 == 0.0;
    }
    if (even_only) LQCHK(lqcd_solve_cg_DdagD_parity(op, X, eta, 0, eps, maxiter, iters, nullptr));
    else if (c->tun.mixed_action_solver) LQCHK(lqcd_solve_mixed_cg_DdagD(op, X, eta, eps, maxiter, 0.0, iters, nullptr, nullptr));
    else LQCHK(cg_run(op, X, eta, eps, maxiter, false, iters, nullptr));
    if (Y) LQCHK(op_apply_async(op, Y, X, 0, nullptr));
    double re = 0, im = 0;
    LQCHK(blas_dot(c, eta->data, X->data, eta->elems, &re, &im, true));
    if (Sf) *Sf = re;
    return LQCD_OK;
}

// G_mu(n) = "U dS_f/dU" from resident X = (D^+D)^-1 eta and Y = D X (force.hip); out is a link-shaped field.
// out = (accumulate ? out : 0) + scale * G: the sum over the poles of a rational action is built in place.
extern "C" int lqcd_fermion_force_acc(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t X, lqcd_spinor_t Y, double scale, int accumulate) {
    LQCHK(check_full(op, X, Y, "lqcd_fermion_force"));
    ARGCHK(out && out->ctx == op->ctx && out != op->gauge, "lqcd_fermion_force: out must be a gauge-shaped field of the same context, not the operator's links");
    LQCHK(force_check(op, "lqcd_fermion_force"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    apply_bc(c, op->bc);
    if (any_partitioned(c)) {   // one exchange step: lower-face X, Y -> the -mu neighbours
        LQCHK(launch_force_pack(c, op->kind, X, Y));
        LQCHK(force_halo_exchange_rccl(c, op->kind));
    }
    LQCHK(launch_fermion_force(c, op->kind, op->gauge, out, X, Y, op->km, op->r, scale, accumulate ? 1 : 0));
    if (op->csw != 0.0 && op->clover) {      // Wilson-clover: + the derivative of the clover term (clover.hip), added in place
        if (!op->clover_lambda) HIPCHK(hipMalloc((void**)&op->clover_lambda, clover_lambda_elems(c->geom) * sizeof(double2)));
        LQCHK(clover_force(c, op->gauge, out, X, Y, op->clover_lambda, op->km, op->csw, scale, 1));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}
extern "C" int lqcd_fermion_force(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t X, lqcd_spinor_t Y) {
    return lqcd_fermion_force_acc(op, out, X, Y, 1.0, 0);
}

// in-process PE-grid emulation of the same sequence (tests): ops/outs/X/Y ordered by rank
extern "C" int lqcd_mdom_fermion_force(int n, lqcd_op_t* ops, lqcd_gauge_t* outs, lqcd_spinor_t* X, lqcd_spinor_t* Y) {
    ARGCHK(ops && outs && X && Y && n >= 1, "lqcd_mdom_fermion_force: null");
    LQCHK(mdom_check(n, ops[0]->ctx));
    std::vector<lqcd_ctx_s*> ctxs(n);
    for (int r = 0; r < n; r++) {
        LQCHK(check_full(ops[r], X[r], Y[r], "lqcd_mdom_fermion_force"));
        ARGCHK(outs[r] && outs[r]->ctx == ops[r]->ctx && outs[r] != ops[r]->gauge, "lqcd_mdom_fermion_force: bad output field");
        ctxs[r] = ops[r]->ctx;
        ARGCHK(ctxs[r]->rank == r, "lqcd_mdom_fermion_force: ops must be ordered by rank");
        apply_bc(ctxs[r], ops[r]->bc);
    }
    for (int r = 0; r < n; r++) LQCHK(launch_force_pack(ctxs[r], ops[r]->kind, X[r], Y[r]));
    LQCHK(force_halo_exchange_local_all(ctxs.data(), n, ops[0]->kind));
    for (int r = 0; r < n; r++) LQCHK(launch_fermion_force(ctxs[r], ops[r]->kind, ops[r]->gauge, outs[r], X[r], Y[r], ops[r]->km, ops[r]->r));
    for (int r = 0; r < n; r++) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    return LQCD_OK;
}

// calc_UdSfdU!: solve, Y = D X and the outer-product sweep back to back on the device
extern "C" int lqcd_calc_UdSfdU(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t eta, double eps, int maxiter, double* Sf, int* iters) {
    ARGCHK(op && eta, "lqcd_calc_UdSfdU: null argument");
    LQCHK(force_check(op, "lqcd_calc_UdSfdU"));
    lqcd_ctx_s* c = op->ctx;
    lqcd_spinor_s* X = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* Y = scratch_get(c, op->kind, LQCD_FULL);
    int st = (X && Y) ? LQCD_OK : LQCD_ERR_HIP;
    if (st == LQCD_OK) st = lqcd_fermi_action(op, eta, X, Y, eps, maxiter, Sf, iters);
    if (st == LQCD_OK) st = lqcd_fermion_force(op, out, X, Y);
    scratch_put(X); scratch_put(Y);
    return st;
}

// ---------------------------------------------------------------------------------- C API: rational (RHMC) action
// The general-Nf pseudofermion action of the reference's staggered runs (test/test_Nf2.toml:8, test/test_Nf3.toml:8, README.md:132):
// S_f = phi^+ (D^+D)^(-alpha) phi with (D^+D)^(-alpha) ~= a0 + sum_k res_k / (D^+D + pole_k).  One multi-shift solve gives every
// X_k = (D^+D + pole_k)^-1 phi; the fields live in the context's scratch pool, nothing leaves the device.
static int rational_solve(lqcd_op_s* op, lqcd_spinor_s* b, int n, const double* poles, double eps, int maxiter, std::vector<lqcd_spinor_s*>& xs,
                          int* iters) {
    lqcd_ctx_s* c = op->ctx;
    xs.assign(n, nullptr);
    for (int k = 0; k < n; k++) {
        xs[k] = scratch_get(c, op->kind, LQCD_FULL);
        if (!xs[k]) { set_error("rational action: out of device memory"); return LQCD_ERR_HIP; }
    }
    if (c->tun.mixed_action_solver && op->kind == LQCD_STAGGERED) {
        // staggered: D^+D + sigma = (m^2 + sigma) - D_hop^2 is the operator of mass sqrt(m^2 + sigma), so every pole is a plain
        // mixed-precision solve (fp32 inner CG, fp64 defect correction, true-residual stopping rule) -- the shifted iterates of a
        // multi-shift CG cost 240 B/site per pole against 2 x 672 for the two Dslashes, so sharing the Krylov space buys little here
        int total = 0;
        for (int k = 0; k < n; k++) {
            lqcd_op_s shifted = *op;        // a view: same links, other mass; nothing is owned
            shifted.km = std::sqrt(op->km * op->km + poles[k]);
            LQCHK(lqcd_spinor_zero(xs[k]));
            int it = 0;
            LQCHK(lqcd_solve_mixed_cg_DdagD(&shifted, xs[k], b, eps, maxiter, 0.0, &it, nullptr, nullptr));
            total += it;
        }
        if (iters) *iters = total;
        return LQCD_OK;
    }
    return lqcd_solve_multishift_cg(op, nullptr, xs.data(), b, poles, n, eps, maxiter, iters, nullptr);
}

// y = a0 x + sum_k res_k (D^+D + pole_k)^-1 x      (action: S_f = Re <phi, y>; heat bath: phi = D^+D y with the 1 - Nf/16 fit)
extern "C" int lqcd_rational_apply(lqcd_op_t op, lqcd_spinor_t y, lqcd_spinor_t x, double a0, int n, const double* res, const double* poles,
                                   double eps, int maxiter, int* iters) {
    LQCHK(check_full(op, y, x, "lqcd_rational_apply"));
    ARGCHK(n >= 1 && res && poles && y != x, "lqcd_rational_apply: need n >= 1 residues and poles and distinct fields");
    lqcd_ctx_s* c = op->ctx;
    std::vector<lqcd_spinor_s*> xs;
    int st = rational_solve(op, x, n, poles, eps, maxiter, xs, iters);
    if (st == LQCD_OK) {
        st = lqcd_spinor_copy(y, x);
        if (st == LQCD_OK) st = lqcd_scale(a0, 0.0, y);
        for (int k = 0; k < n && st == LQCD_OK; k++) st = lqcd_axpy(res[k], 0.0, xs[k], y);
    }
    for (auto* s : xs) scratch_put(s);
    (void)c;
    return st;
}

// out = sum_k res_k G[X_k, D X_k]: the force of S_f = phi^+ r(D^+D) phi (d(A + p)^-1 = -(A + p)^-1 dA (A + p)^-1 term by term)
extern "C" int lqcd_rational_force(lqcd_op_t op, lqcd_gauge_t out, lqcd_spinor_t phi, int n, const double* res, const double* poles, double eps,
                                   int maxiter, int* iters) {
    ARGCHK(op && phi && out && n >= 1 && res && poles, "lqcd_rational_force: null argument");
    LQCHK(force_check(op, "lqcd_rational_force"));
    lqcd_ctx_s* c = op->ctx;
    std::vector<lqcd_spinor_s*> xs;
    lqcd_spinor_s* Y = scratch_get(c, op->kind, LQCD_FULL);
    int st = Y ? rational_solve(op, phi, n, poles, eps, maxiter, xs, iters) : LQCD_ERR_HIP;
    for (int k = 0; k < n && st == LQCD_OK; k++) {
        st = op_apply_async(op, Y, xs[k], 0, nullptr);
        if (st == LQCD_OK) st = lqcd_fermion_force_acc(op, out, xs[k], Y, res[k], k > 0);
    }
    for (auto* s : xs) scratch_put(s);
    scratch_put(Y);
    return st;
}

// ---------------------------------------------------------------------------------- C API: timing
extern "C" int lqcd_bench_dslash(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int warm, int reps, double* ms) {
    LQCHK(check_full(op, out, in, "lqcd_bench_dslash"));
    ARGCHK(reps > 0 && ms, "lqcd_bench_dslash: bad reps");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    for (int i = 0; i < warm; i++) LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
    HIPCHK(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; i++) LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
    HIPCHK(hipEventRecord(c->ev_t1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev_t1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->ev_t0, c->ev_t1));
    *ms = (double)t / reps;
    return LQCD_OK;
}

// SURVEY.md 8(d) timing protocol: every application bracketed by its own HIP events, median (and mean) over `reps`
extern "C" int lqcd_bench_dslash_median(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int warm, int reps, double* median_ms,
                                        double* mean_ms) {
    LQCHK(check_full(op, out, in, "lqcd_bench_dslash_median"));
    ARGCHK(reps > 0 && reps <= 4096 && median_ms, "lqcd_bench_dslash_median: bad reps");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    std::vector<hipEvent_t> ev(reps + 1);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    for (int i = 0; i < warm; i++) LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
    HIPCHK(hipEventRecord(ev[0], c->stream));
    for (int i = 0; i < reps; i++) {
        LQCHK(op_apply_async(op, out, in, dagger ? 1 : 0, nullptr));
        HIPCHK(hipEventRecord(ev[i + 1], c->stream));
    }
    HIPCHK(hipEventSynchronize(ev[reps]));
    std::vector<double> t(reps);
    double sum = 0;
    for (int i = 0; i < reps; i++) {
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
        t[i] = ms;
        sum += ms;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    std::sort(t.begin(), t.end());
    *median_ms = reps % 2 ? t[reps / 2] : 0.5 * (t[reps / 2 - 1] + t[reps / 2]);
    if (mean_ms) *mean_ms = sum / reps;
    return LQCD_OK;
}

// Phase breakdown of ONE partitioned operator application (mean over reps, each rep synchronised):
//   ms[0] pack, ms[1] interior kernel (overlapping the exchange), ms[2] exchange = pack end -> last halo byte received (RCCL on the
//   communication stream, incl. its launch latency), ms[3] compute stream idle waiting for the exchange after the interior,
//   ms[4] exterior, ms[5] whole application.  What the N > 1 lines of bench.py report, so that the critical path of the halo
//   exchange on real xGMI links is visible from the driver's multi-GPU runs.
extern "C" int lqcd_bench_halo_phases(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int reps, double* ms) {
    LQCHK(check_full(op, out, in, "lqcd_bench_halo_phases"));
    ARGCHK(reps > 0 && ms, "lqcd_bench_halo_phases: bad arguments");
    lqcd_ctx_s* c = op->ctx;
    ARGCHK(any_partitioned(c) && c->has_comm && c->local_peers.empty(), "lqcd_bench_halo_phases: needs a partitioned context with RCCL communicators");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t e[6];
    for (auto& ev : e) HIPCHK(hipEventCreate(&ev));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    apply_bc(c, op->bc);
    StencilCall s;
    LQCHK(make_full_call(op, out, in, dagger ? 1 : 0, s));
    for (int r = 0; r < reps + 2; r++) {          // two untimed warm-up applications
        HIPCHK(hipEventRecord(e[0], c->stream));
        LQCHK(launch_stencil_pack(c, s));
        HIPCHK(hipEventRecord(e[1], c->stream));
        LQCHK(halo_exchange_rccl(c, s.kind, s.parity_mode, 0, 0));
        HIPCHK(hipEventRecord(e[5], c->comm_stream));
        LQCHK(launch_stencil_interior(c, s));
        HIPCHK(hipEventRecord(e[2], c->stream));
        HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
        HIPCHK(hipEventRecord(e[3], c->stream));
        LQCHK(launch_stencil_exterior(c, s));
        HIPCHK(hipEventRecord(e[4], c->stream));
        HIPCHK(hipEventSynchronize(e[4]));
        HIPCHK(hipEventSynchronize(e[5]));
        if (r < 2) continue;
        float t;
        HIPCHK(hipEventElapsedTime(&t, e[0], e[1])); acc[0] += t;
        HIPCHK(hipEventElapsedTime(&t, e[1], e[2])); acc[1] += t;
        HIPCHK(hipEventElapsedTime(&t, e[1], e[5])); acc[2] += t;
        HIPCHK(hipEventElapsedTime(&t, e[2], e[3])); acc[3] += t;
        HIPCHK(hipEventElapsedTime(&t, e[3], e[4])); acc[4] += t;
        HIPCHK(hipEventElapsedTime(&t, e[0], e[4])); acc[5] += t;
    }
    for (int k = 0; k < 6; k++) ms[k] = acc[k] / reps;
    for (auto& ev : e) (void)hipEventDestroy(ev);
    return LQCD_OK;
}

// mean latency (microseconds) of the stream-ordered one-double all-reduce the solvers issue twice per CG iteration
extern "C" int lqcd_bench_allreduce(lqcd_ctx_t c, int reps, double* us) {
    ARGCHK(c && reps > 0 && us, "lqcd_bench_allreduce: bad arguments");
    ARGCHK(c->has_comm, "lqcd_bench_allreduce: communicators not initialised");
    HIPCHK(hipSetDevice(c->device));
    double* d = c->d_scal + S_RED0;
    HIPCHK(hipMemsetAsync(d, 0, sizeof(double), c->stream));   // 0 + 0 + ... stays finite however often it is summed
    for (int i = 0; i < 5; i++) NCCLCHK(ncclAllReduce(d, d, 1, ncclDouble, ncclSum, c->comm_red, c->stream));
    HIPCHK(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; i++) NCCLCHK(ncclAllReduce(d, d, 1, ncclDouble, ncclSum, c->comm_red, c->stream));
    HIPCHK(hipEventRecord(c->ev_t1, c->stream));
    HIPCHK(hipEventSynchronize(c->ev_t1));
    float t = 0;
    HIPCHK(hipEventElapsedTime(&t, c->ev_t0, c->ev_t1));
    *us = 1e3 * (double)t / reps;
    return LQCD_OK;
}

extern "C" int lqcd_bench_cg(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int warm, int niter, double* ms_per_iter) {
    LQCHK(check_full(op, x, b, "lqcd_bench_cg"));
    ARGCHK(niter > 0 && ms_per_iter, "lqcd_bench_cg: bad niter");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    CgWork w;
    w.r = scratch_get(c, x->kind, LQCD_FULL); w.p = scratch_get(c, x->kind, LQCD_FULL);
    w.q = scratch_get(c, x->kind, LQCD_FULL); w.tmp = scratch_get(c, x->kind, LQCD_FULL);
    if (!(w.r && w.p && w.q && w.tmp)) return LQCD_ERR_HIP;
    double rr;
    int st = cg_setup(op, x, b, w, -1.0, &rr);
    for (int i = 0; i < warm && st == LQCD_OK; i++) st = cg_enqueue_iteration(op, x, w);
    if (st == LQCD_OK) {
        (void)hipEventRecord(c->ev_t0, c->stream);
        for (int i = 0; i < niter && st == LQCD_OK; i++) st = cg_enqueue_iteration(op, x, w);
        (void)hipEventRecord(c->ev_t1, c->stream);
        hipError_t e = hipEventSynchronize(c->ev_t1);
        float t = 0;
        if (e == hipSuccess) e = hipEventElapsedTime(&t, c->ev_t0, c->ev_t1);
        if (e != hipSuccess && st == LQCD_OK) st = hip_fail(e, "bench_cg timing", __FILE__, __LINE__);
        *ms_per_iter = (double)t / niter;
    }
    scratch_put(w.r); scratch_put(w.p); scratch_put(w.q); scratch_put(w.tmp);
    return st;
}

// CG session (externally timed windows); the state lives in the context, one open session per context
namespace lqcd {
struct CgSession { lqcd_op_s* op = nullptr; lqcd_spinor_s* x = nullptr; CgWork w; };
}
extern "C" int lqcd_cg_session_begin(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b) {
    LQCHK(check_full(op, x, b, "lqcd_cg_session_begin"));
    lqcd_ctx_s* c = op->ctx;
    ARGCHK(c->cg_session == nullptr, "lqcd_cg_session_begin: a session is already open on this context");
    HIPCHK(hipSetDevice(c->device));
    CgSession* ses = new CgSession;
    CgWork& w = ses->w;
    w.r = scratch_get(c, x->kind, LQCD_FULL); w.p = scratch_get(c, x->kind, LQCD_FULL);
    w.q = scratch_get(c, x->kind, LQCD_FULL); w.tmp = scratch_get(c, x->kind, LQCD_FULL);
    int st = (w.r && w.p && w.q && w.tmp) ? LQCD_OK : LQCD_ERR_HIP;
    double rr;
    if (st == LQCD_OK) st = cg_setup(op, x, b, w, -1.0, &rr);
    if (st != LQCD_OK) {
        scratch_put(w.r); scratch_put(w.p); scratch_put(w.q); scratch_put(w.tmp);
        delete ses;
        return st;
    }
    ses->op = op;
    ses->x = x;
    c->cg_session = ses;
    return LQCD_OK;
}
extern "C" int lqcd_cg_session_iterate(lqcd_op_t op, int n) {
    ARGCHK(op && op->ctx->cg_session && static_cast<CgSession*>(op->ctx->cg_session)->op == op, "lqcd_cg_session_iterate: no open session for this operator");
    CgSession* ses = static_cast<CgSession*>(op->ctx->cg_session);
    for (int i = 0; i < n; i++) LQCHK(cg_enqueue_iteration(op, ses->x, ses->w));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return LQCD_OK;
}
extern "C" int lqcd_cg_session_end(lqcd_op_t op) {
    ARGCHK(op && op->ctx->cg_session && static_cast<CgSession*>(op->ctx->cg_session)->op == op, "lqcd_cg_session_end: no open session for this operator");
    CgSession* ses = static_cast<CgSession*>(op->ctx->cg_session);
    CgWork& w = ses->w;
    scratch_put(w.r); scratch_put(w.p); scratch_put(w.q); scratch_put(w.tmp);
    delete ses;
    op->ctx->cg_session = nullptr;
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- in-process multi-domain collectives
namespace lqcd {
int plaquette_local_sum(lqcd_gauge_s* g, const double2* const ghost[4], double* sum);
int gauge_pack_face(lqcd_gauge_s* g, int mu, double2* dst);
}

static int mdom_check(int n, lqcd_ctx_s* c0) {
    ARGCHK(n >= 1 && c0 && (int)c0->local_peers.size() == n, "lqcd_mdom_*: contexts are not linked with lqcd_ctx_link_local (or wrong n)");
    return LQCD_OK;
}

extern "C" int lqcd_mdom_op_apply(int n, lqcd_op_t* ops, lqcd_spinor_t* outs, lqcd_spinor_t* ins, int dagger) {
    ARGCHK(ops && outs && ins && n >= 1, "lqcd_mdom_op_apply: null");
    LQCHK(mdom_check(n, ops[0]->ctx));
    std::vector<lqcd_ctx_s*> ctxs(n);
    std::vector<StencilCall> calls(n);
    for (int r = 0; r < n; r++) {
        LQCHK(check_full(ops[r], outs[r], ins[r], "lqcd_mdom_op_apply"));
        ctxs[r] = ops[r]->ctx;
        ARGCHK(ctxs[r]->rank == r, "lqcd_mdom_op_apply: ops must be ordered by rank");
        apply_bc(ctxs[r], ops[r]->bc);
        LQCHK(make_full_call(ops[r], outs[r], ins[r], dagger ? 1 : 0, calls[r]));
        if (ops[r]->kind == LQCD_WILSON && ops[r]->r != 1.0) { set_error("r != 1 unsupported on a partitioned lattice"); return LQCD_ERR_UNSUPPORTED; }
    }
    for (int r = 0; r < n; r++) LQCHK(launch_stencil_pack(ctxs[r], calls[r]));
    LQCHK(halo_exchange_local_all(ctxs.data(), n, ops[0]->kind, 2));
    for (int r = 0; r < n; r++) LQCHK(launch_stencil_interior(ctxs[r], calls[r]));
    for (int r = 0; r < n; r++) LQCHK(launch_stencil_exterior(ctxs[r], calls[r]));
    for (int r = 0; r < n; r++) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    return LQCD_OK;
}

extern "C" int lqcd_mdom_dot(int n, lqcd_spinor_t* a, lqcd_spinor_t* b, double* re, double* im) {
    ARGCHK(a && b && re && im && n >= 1, "lqcd_mdom_dot: null");
    double sr = 0, si = 0;
    for (int r = 0; r < n; r++) {
        double x, y;
        LQCHK(blas_dot(a[r]->ctx, a[r]->data, b[r]->data, a[r]->elems, &x, &y, false));
        sr += x; si += y;
    }
    *re = sr; *im = si;
    return LQCD_OK;
}

extern "C" int lqcd_mdom_plaquette(int n, lqcd_gauge_t* g, double* plaq) {
    ARGCHK(g && plaq && n >= 1, "lqcd_mdom_plaquette: null");
    LQCHK(mdom_check(n, g[0]->ctx));
    // exchange forward gauge faces: ghost[mu] of rank r = x_mu = 0 slice of rank nbr_fwd[mu]
    std::vector<std::vector<double2*>> ghost(n, std::vector<double2*>(4, nullptr));
    int st = LQCD_OK;
    for (int r = 0; r < n && st == LQCD_OK; r++) {
        lqcd_ctx_s* c = g[r]->ctx;
        for (int mu = 0; mu < 4 && st == LQCD_OK; mu++) {
            if (!c->geom.part[mu]) continue;
            const size_t elems = (size_t)2 * 4 * 9 * face_half_sites(c->geom, mu);
            if (hipMalloc((void**)&ghost[r][mu], elems * sizeof(double2)) != hipSuccess) { st = LQCD_ERR_HIP; break; }
            st = gauge_pack_face(g[c->nbr_fwd[mu]], mu, ghost[r][mu]);
        }
    }
    if (st == LQCD_OK && hipDeviceSynchronize() != hipSuccess) st = LQCD_ERR_HIP;
    double total = 0;
    for (int r = 0; r < n && st == LQCD_OK; r++) {
        double s;
        const double2* gp[4] = {ghost[r][0], ghost[r][1], ghost[r][2], ghost[r][3]};
        st = plaquette_local_sum(g[r], gp, &s);
        total += s;
    }
    for (int r = 0; r < n; r++) for (int mu = 0; mu < 4; mu++) if (ghost[r][mu]) (void)hipFree(ghost[r][mu]);
    if (st != LQCD_OK) return st;
    lqcd_ctx_s* c0 = g[0]->ctx;
    const double V = (double)c0->gL[0] * c0->gL[1] * c0->gL[2] * c0->gL[3];
    *plaq = total / (6.0 * V * 3.0);
    return LQCD_OK;
}

// plain host-driven CG over the linked domains (tests the halo path inside a solver)
extern "C" int lqcd_mdom_solve_cg_DdagD(int n, lqcd_op_t* ops, lqcd_spinor_t* x, lqcd_spinor_t* b, double eps, int maxiter, int* iters,
                                        double* final_rr) {
    ARGCHK(ops && x && b && n >= 1, "lqcd_mdom_solve_cg_DdagD: null");
    LQCHK(mdom_check(n, ops[0]->ctx));
    std::vector<lqcd_spinor_t> r(n), p(n), q(n), tmp(n);
    for (int k = 0; k < n; k++) {
        lqcd_ctx_s* c = ops[k]->ctx;
        r[k] = scratch_get(c, ops[k]->kind, LQCD_FULL); p[k] = scratch_get(c, ops[k]->kind, LQCD_FULL);
        q[k] = scratch_get(c, ops[k]->kind, LQCD_FULL); tmp[k] = scratch_get(c, ops[k]->kind, LQCD_FULL);
        if (!(r[k] && p[k] && q[k] && tmp[k])) return LQCD_ERR_HIP;
    }
    auto release = [&]() { for (int k = 0; k < n; k++) { scratch_put(r[k]); scratch_put(p[k]); scratch_put(q[k]); scratch_put(tmp[k]); } };
    auto sync_all = [&]() { for (int k = 0; k < n; k++) (void)hipStreamSynchronize(ops[k]->ctx->stream); };
    int st = lqcd_mdom_op_apply(n, ops, tmp.data(), x, 0);
    if (st == LQCD_OK) st = lqcd_mdom_op_apply(n, ops, q.data(), tmp.data(), 1);
    for (int k = 0; k < n && st == LQCD_OK; k++) {
        st = lqcd_spinor_copy(r[k], b[k]);
        if (st == LQCD_OK) st = lqcd_axpy(-1.0, 0.0, q[k], r[k]);
        if (st == LQCD_OK) st = lqcd_spinor_copy(p[k], r[k]);
    }
    double rr = 0, im;
    if (st == LQCD_OK) st = lqcd_mdom_dot(n, r.data(), r.data(), &rr, &im);
    int it = 0;
    bool conv = st == LQCD_OK && rr < eps;
    while (st == LQCD_OK && !conv && it < maxiter) {
        it++;
        st = lqcd_mdom_op_apply(n, ops, tmp.data(), p.data(), 0);
        if (st == LQCD_OK) st = lqcd_mdom_op_apply(n, ops, q.data(), tmp.data(), 1);
        double pq = 0;
        if (st == LQCD_OK) st = lqcd_mdom_dot(n, p.data(), q.data(), &pq, &im);
        const double alpha = rr / pq;
        for (int k = 0; k < n && st == LQCD_OK; k++) {
            st = blas_axpy(ops[k]->ctx, alpha, 0, p[k]->data, x[k]->data, x[k]->elems);
            if (st == LQCD_OK) st = blas_axpy(ops[k]->ctx, -alpha, 0, q[k]->data, r[k]->data, x[k]->elems);
        }
        sync_all();
        double rrn = 0;
        if (st == LQCD_OK) st = lqcd_mdom_dot(n, r.data(), r.data(), &rrn, &im);
        if (rrn < eps) { rr = rrn; conv = true; break; }
        const double beta = rrn / rr;
        for (int k = 0; k < n && st == LQCD_OK; k++) st = blas_axpby(ops[k]->ctx, 1.0, 0, r[k]->data, beta, 0, p[k]->data, x[k]->elems);
        sync_all();
        rr = rrn;
    }
    release();
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (st != LQCD_OK) return st;
    if (!conv) { set_error("The CG is not converged! (mdom)"); return LQCD_ERR_NOT_CONVERGED; }
    return LQCD_OK;
}
