// mdom.hip -- in-process emulation of the PE grid (N contexts on one device linked by lqcd_ctx_link_local, halos moved by device
// copies): the collectives the tests use to check the partitioned kernels and geometry without RCCL.
#include "ops_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>

using namespace lqcd;

// ---------------------------------------------------------------------------------- in-process multi-domain collectives
namespace lqcd {
int plaquette_local_sum(lqcd_gauge_s* g, const double2* const ghost[4], double* sum);
int gauge_pack_face(lqcd_gauge_s* g, int mu, double2* dst);
}

int lqcd::mdom_check(int n, lqcd_ctx_s* c0) {
    ARGCHK(n >= 1 && c0 && (int)c0->local_peers.size() == n, "lqcd_mdom_*: contexts are not linked with lqcd_ctx_link_local (or wrong n)");
    return LQCD_OK;
}

extern "C" int lqcd_mdom_op_apply(int n, lqcd_op_t* ops, lqcd_spinor_t* outs, lqcd_spinor_t* ins, int dagger) {
    ARGCHK(ops && outs && ins && n >= 1, "lqcd_mdom_op_apply: null");
    LQCHK(mdom_check(n, ops[0]->ctx));
    std::vector<lqcd_ctx_s*> ctxs(n);
    std::vector<StencilCall> calls(n), calls2(n);
    bool general_r = false;
    for (int r = 0; r < n; r++) {
        LQCHK(check_full(ops[r], outs[r], ins[r], "lqcd_mdom_op_apply"));
        ctxs[r] = ops[r]->ctx;
        ARGCHK(ctxs[r]->rank == r, "lqcd_mdom_op_apply: ops must be ordered by rank");
        apply_bc(ctxs[r], ops[r]->bc);
        LQCHK(make_full_call(ops[r], outs[r], ins[r], dagger ? 1 : 0, calls[r]));
        if (ops[r]->kind == LQCD_WILSON && ops[r]->r != 1.0) {      // two r = 1 passes (apply.hip, split_general_r)
            const StencilCall full = calls[r];
            split_general_r(full, calls[r], calls2[r]);
            general_r = true;
        }
    }
    for (int pass = 0; pass < (general_r ? 2 : 1); pass++) {
        std::vector<StencilCall>& cs = pass ? calls2 : calls;
        for (int r = 0; r < n; r++) LQCHK(launch_stencil_pack(ctxs[r], cs[r]));
        LQCHK(halo_exchange_local_all(ctxs.data(), n, ops[0]->kind, 2));
        for (int r = 0; r < n; r++) LQCHK(launch_stencil_interior(ctxs[r], cs[r]));
        for (int r = 0; r < n; r++) LQCHK(launch_stencil_exterior(ctxs[r], cs[r]));
        for (int r = 0; r < n; r++) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    }
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
