// bench_api.hip -- timing entry points (HIP events on the library's own stream) and the externally timed CG session used by bench.py.
#include "ops_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>

using namespace lqcd;

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
    ARGCHK(any_partitioned(c) && c->has_comm && c->local_peers.empty(), "lqcd_bench_halo_phases: needs a partitioned context with a communicator (RCCL or peer-mapped)");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t e[6];
    for (auto& ev : e) HIPCHK(hipEventCreate(&ev));
    double acc[6] = {0, 0, 0, 0, 0, 0};
    apply_bc(c, op->bc);
    StencilCall s;
    LQCHK(make_full_call(op, out, in, dagger ? 1 : 0, s));
    LQCHK(halo_schedule_settle(op));
    // the folded one-stream schedule (round 5; what the solvers run when the collective timing picked schedule 3 and halo_fold applies): pack -> exchange ->
    // ONE stencil launch that takes the boundary hops from the ghost buffers.  ms[1] = that launch, ms[2] = the exchange, ms[3] = ms[4] = 0: nothing waits and
    // there is no exterior kernel.  (Inside the fused CG the pack launch is gone as well: the x/p update and the reduction launch write the faces.)
    if (halo_fold_applies(c, s.kind, s.r, s.parity_mode, s.prec, s.clover != nullptr)) {
        StencilCall f = s;
        f.fold = 1;
        for (int r = 0; r < reps + 2; r++) {
            HIPCHK(hipEventRecord(e[0], c->stream));
            LQCHK(launch_stencil_pack(c, s));
            HIPCHK(hipEventRecord(e[1], c->stream));
            LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, 0, 1));
            HIPCHK(hipEventRecord(e[2], c->stream));
            LQCHK(launch_stencil_interior(c, f));
            HIPCHK(hipEventRecord(e[3], c->stream));
            HIPCHK(hipEventSynchronize(e[3]));
            if (r < 2) continue;
            float t;
            HIPCHK(hipEventElapsedTime(&t, e[0], e[1])); acc[0] += t;
            HIPCHK(hipEventElapsedTime(&t, e[1], e[2])); acc[2] += t;
            HIPCHK(hipEventElapsedTime(&t, e[2], e[3])); acc[1] += t;
            HIPCHK(hipEventElapsedTime(&t, e[0], e[3])); acc[5] += t;
        }
        for (int k = 0; k < 6; k++) ms[k] = acc[k] / reps;
        for (auto& ev : e) (void)hipEventDestroy(ev);
        return LQCD_OK;
    }
    for (int r = 0; r < reps + 2; r++) {          // two untimed warm-up applications
        HIPCHK(hipEventRecord(e[0], c->stream));
        LQCHK(launch_stencil_pack(c, s));
        HIPCHK(hipEventRecord(e[1], c->stream));
        LQCHK(comm_halo_exchange(c, s.kind, s.parity_mode, 0, 0));
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
    for (int i = 0; i < 5; i++) LQCHK(comm_allreduce(c, d, 1));
    HIPCHK(hipEventRecord(c->ev_t0, c->stream));
    for (int i = 0; i < reps; i++) LQCHK(comm_allreduce(c, d, 1));
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
    LQCHK(cg_work_get(c, x->kind, w));
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
    if (st == LQCD_OK) st = cg_flush_x(op, x, w);
    (void)hipStreamSynchronize(c->stream);
    cg_work_put(w);
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
    int st = cg_work_get(c, x->kind, w);
    double rr;
    if (st == LQCD_OK) st = cg_setup(op, x, b, w, -1.0, &rr);
    if (st != LQCD_OK) {
        cg_work_put(w);
        delete ses;
        return st;
    }
    ses->op = op;
    ses->x = x;
    c->cg_session = ses;
    return LQCD_OK;
}
extern "C" int lqcd_cg_session_iterate(lqcd_op_t op, int n) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
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
    const int st = cg_flush_x(op, ses->x, w);       // a pending deferred x update; its status is the status of the session
    (void)hipStreamSynchronize(op->ctx->stream);
    cg_work_put(w);
    delete ses;
    op->ctx->cg_session = nullptr;
    return st;
}
