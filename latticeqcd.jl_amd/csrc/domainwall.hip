// domainwall.hip -- Dirac_operator = "Domainwall" (src/system/universe.jl:116-128: params "mass", "L5", "M"; test/test_domainwallhmc.toml:
// Domainwall_M = -1, Domainwall_L5 = 4, Domainwall_m = 1; the fifth HMC fermion test of test/runtests.jl:132-137).
//
// The operator's arithmetic lives in LatticeDiracOperators.jl, which is not under /root/reference: what follows is the textbook Shamir operator in the
// conventions of SURVEY.md Appendix A [EXT-RECALL -- parity unpinned, as the Wilson operator it is built from]:
//
//     (D5 psi)(s) = D4 psi(s) + psi(s) - P_- psi(s+1) - P_+ psi(s-1),      psi(L5+1) := -m psi(1),  psi(0) := -m psi(L5),
//     D4 = (4 + M) - 1/2 sum_nu [ (1 - g_nu) U_nu(n) psi(n+nu) + (1 + g_nu) U_nu(n-nu)^+ psi(n-nu) ]      (Wilson operator of mass M, r = 1),
//     P_+- = (1 +- g5)/2,   D5^+ : P_+ <-> P_- in the fifth-direction hops (D5 is g5 R hermitian).
//
// A five-dimensional field is L5 Wilson fields in ONE allocation [s][parity][chunk][12][lane]: BLAS-1 sees a flat array, a slice is a Wilson field the
// four-dimensional kernels take as it is.  Where the scalar-addressing Wilson kernel applies (fp64, one GPU, z-planes of whole chunks, links on the group) an
// application is ONE launch over all slices (stencil.hip, the DW5 instance of wilson_dirsplit_s): block -> (chunk, slice) with the block's XCD kept and the slices of
// a chunk following each other, out = (5 + M) in - 1/2 H in per slice and the fifth-direction hops added in the epilogue from 12 coalesced loads per lane
// (32^3x64 x 8: 2.87 ms against 4.99; 16^3x32 x 8: 0.167 against 0.35).  Elsewhere (partitioned lattices, reference-format links, small planes): L5 launches of
// the Wilson Dslash and one streaming kernel for the fifth direction (tunable dw_batched = 0 forces this form; read-only dw_active says which ran).
//
// Action (two flavours, Pauli-Villars field of mass 1):  S = phi^+ D_PV (D^+D)^-1 D_PV^+ phi,  D = D5(m), D_PV = D5(1).
//     heat bath:  phi = D_PV^-+ D^+ xi  (one CG on D_PV^+ D_PV)                          -> S = xi^+ xi
//     force:      X = (D^+D)^-1 D_PV^+ phi, Y = D X:  dS = -2 Re[(Y - phi)^+ dD X]        (dD_PV = dD: only the 4-D hops carry links)
//                 = the Wilson outer-product sweep (force.hip) of the pairs (X(s), Y(s) - phi(s)), summed over s.
// With m = 1 (the reference's test) D = D_PV, S = phi^+ phi and the force vanishes identically -- the run is a quenched HMC with a spectator field.
// Partitioned lattices (RCCL ranks): the slices go through the halo path of the Wilson operator as they are, the fifth direction is local to a site, the CG's
// inner products are summed over ranks by the BLAS layer and the force sweep exchanges its faces per slice.
#include "ops_internal.h"

using namespace lqcd;

namespace {

constexpr int DW_WORK = 8;      // r, p, q, tmp of the CG; psi = D_PV^+ phi; X, Y of the force; one spare

// out(s) += cu P_A in(s+1) + cd P_B in(s-1); (P_- psi)[e] = (psi[e] + psi[e'])/2, (P_+ psi)[e] = (psi[e] - psi[e'])/2 with e' the element two spins away
// (g5 psi)_s = -psi_(s+2): SURVEY Appendix A).  One thread per element of a slice, walking s.
__global__ __launch_bounds__(256) void dw_fifth_kernel(double2* __restrict__ out, const double2* __restrict__ in, size_t slice, int L5, double mass, int dagger) {
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= slice) return;
    const int comp = (int)((e >> 6) % 12);
    const size_t ep = comp < 6 ? e + 6 * 64 : e - 6 * 64;
    const double sa = dagger ? -1.0 : 1.0;      // P_A = P_- (P_+ in the adjoint): + (-) the partner element
    for (int s = 0; s < L5; s++) {
        const int su = s + 1 < L5 ? s + 1 : 0, sd = s >= 1 ? s - 1 : L5 - 1;
        const double cu = 0.5 * (s + 1 < L5 ? -1.0 : mass), cd = 0.5 * (s >= 1 ? -1.0 : mass);
        const double2 u0 = in[(size_t)su * slice + e], u1 = in[(size_t)su * slice + ep];
        const double2 d0 = in[(size_t)sd * slice + e], d1 = in[(size_t)sd * slice + ep];
        double2 o = out[(size_t)s * slice + e];
        o.x += cu * (u0.x + sa * u1.x) + cd * (d0.x - sa * d1.x);
        o.y += cu * (u0.y + sa * u1.y) + cd * (d0.y - sa * d1.y);
        out[(size_t)s * slice + e] = o;
    }
}

lqcd_spinor_s slice_view(const lqcd_spinor_s* s5, int i5) {
    lqcd_spinor_s v = *s5;
    v.kind = LQCD_WILSON;
    v.ls = 1;
    v.owner = false;
    v.view_of = nullptr; v.nviews = 0; v.zombie = false;
    v.elems = s5->elems / s5->ls;
    v.data = s5->data + (size_t)i5 * v.elems;
    return v;
}

int dw_check(lqcd_op_s* op, lqcd_spinor_s* a, lqcd_spinor_s* b, const char* who) {
    LQCHK(check_full(op, a, b, (std::string("dw:") + who).c_str()));
    if (!(op->kind == LQCD_DOMAINWALL && a->ls == op->L5 && b->ls == op->L5)) {
        set_error(std::string(who) + ": the fields do not have the operator's L5");
        return LQCD_ERR_ARG;
    }
    return LQCD_OK;
}

// out = D5(mass) in or its adjoint on raw five-dimensional buffers (enqueued, no synchronisation)
int dw_apply_raw(lqcd_op_s* op, double2* out, const double2* in, int dagger, double mass) {
    lqcd_ctx_s* c = op->ctx;
    lqcd_op_s* w = op->dw_wilson;
    w->gauge = op->gauge;
    const size_t slice = (size_t)12 * c->geom.Vs * 2;
    lqcd_spinor_s vi, vo;
    vi.ctx = vo.ctx = c; vi.kind = vo.kind = LQCD_WILSON; vi.subset = vo.subset = LQCD_FULL; vi.ncomp = vo.ncomp = 12;
    vi.elems = vo.elems = slice; vi.owner = vo.owner = false;
    apply_bc(c, op->bc);
    if (c->tun.dw_batched) {      // all slices in ONE launch with the fifth-direction hops in its epilogue, where the scalar-addressing kernel applies (stencil.hip DW5)
        vi.data = const_cast<double2*>(in);
        vo.data = out;
        StencilCall sc;
        LQCHK(make_full_call(w, &vo, &vi, dagger, sc));
        sc.a = 5.0 + op->dw_M;
        sc.b = -0.5;
        sc.dw_ls = op->L5; sc.dw_slice = slice; sc.dw_mass = mass;
        if (stencil_dw5_applies(c, sc)) { c->tun.dw_active = 1; return stencil_apply(c, sc); }
    }
    c->tun.dw_active = 0;
    for (int s = 0; s < op->L5; s++) {
        vi.data = const_cast<double2*>(in) + (size_t)s * slice;
        vo.data = out + (size_t)s * slice;
        StencilCall sc;
        LQCHK(make_full_call(w, &vo, &vi, dagger, sc));
        sc.a = 5.0 + op->dw_M;      // (4 + M) psi + psi
        sc.b = -0.5;
        LQCHK(stencil_apply(c, sc));
    }
    hipLaunchKernelGGL(dw_fifth_kernel, dim3((unsigned)((slice + 255) / 256)), dim3(256), 0, c->stream, out, in, slice, op->L5, mass, dagger);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int dw_work(lqcd_op_s* op, int i, lqcd_spinor_s** out) {
    if (!op->dw_work[i]) LQCHK(lqcd_spinor_create_5d(op->ctx, &op->dw_work[i], op->L5));
    *out = op->dw_work[i];
    return LQCD_OK;
}

// the batched call of dw_apply_raw (all slices in one launch), or false where it does not apply
bool dw_batched_call(lqcd_op_s* op, double2* out, const double2* in, int dagger, double mass, StencilCall& sc) {
    lqcd_ctx_s* c = op->ctx;
    if (!c->tun.dw_batched) return false;
    lqcd_op_s* w = op->dw_wilson;
    w->gauge = op->gauge;
    const size_t slice = (size_t)12 * c->geom.Vs * 2;
    lqcd_spinor_s vi, vo;
    vi.ctx = vo.ctx = c; vi.kind = vo.kind = LQCD_WILSON; vi.subset = vo.subset = LQCD_FULL; vi.ncomp = vo.ncomp = 12;
    vi.elems = vo.elems = slice; vi.owner = vo.owner = false;
    vi.data = const_cast<double2*>(in);
    vo.data = out;
    if (make_full_call(w, &vo, &vi, dagger, sc) != LQCD_OK) return false;
    sc.a = 5.0 + op->dw_M;
    sc.b = -0.5;
    sc.dw_ls = op->L5; sc.dw_slice = slice; sc.dw_mass = mass;
    return stencil_dw5_applies(c, sc);
}

// x = (D5(mass)^+ D5(mass))^-1 b from a zero guess.
// Where the five-dimensional launch applies (tunable dw_fused_cg): the fused iteration of the four-dimensional CG (solvers.hip cg_enqueue_iteration, its plain form) --
//     t = D p with |t|^2 partials in the epilogue (one per workgroup = chunk x slice) ;  alpha = rr / |t|^2 in the reduction launch ;
//     D^+ t in update mode: r -= alpha (D^+ t), |r|^2 partials, q = D^+D p never written ;  beta and the stopping test in the reduction launch ;
//     x += alpha p, p = r + beta p in one pass
// = 5 launches and 5 vector passes beside the operator per iteration where cg_generic has 11 and 11.  Same recurrences and stopping rule (r.r < eps).
int dw_solve(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double mass, double eps, int maxiter, int* iters, double* rr_out) {
    lqcd_ctx_s* c = op->ctx;
    lqcd_spinor_s *r, *p, *q, *t;
    LQCHK(dw_work(op, 0, &r)); LQCHK(dw_work(op, 1, &p)); LQCHK(dw_work(op, 2, &q)); LQCHK(dw_work(op, 3, &t));
    StencilCall probe;
    const bool fused = c->tun.dw_fused_cg && c->tun.cg_fused >= 2 && !c->has_comm && dw_batched_call(op, t->data, p->data, 0, mass, probe);
    if (!fused) {
        ApplyFn A = [&](double2* out, const double2* in) -> int {
            LQCHK(dw_apply_raw(op, t->data, in, 0, mass));
            return dw_apply_raw(op, out, t->data, 1, mass);
        };
        return cg_generic(c, A, x->elems, x->data, b->data, r->data, p->data, q->data, eps, maxiter, iters, rr_out);
    }
    const size_t n = x->elems, slice = (size_t)12 * c->geom.Vs * 2;
    const int nparts = stencil_num_blocks(c, LQCD_WILSON, 1.0, 2, 0, false) * op->L5;
    if (op->dw_partial_n < (size_t)nparts) {
        if (op->dw_partial) HIPCHK(hipFree(op->dw_partial));
        op->dw_partial = nullptr; op->dw_partial_n = 0;
        HIPCHK(hipMalloc((void**)&op->dw_partial, (size_t)nparts * sizeof(double)));
        op->dw_partial_n = (size_t)nparts;
    }
    // r = b - D^+D x ; p = r
    LQCHK(dw_apply_raw(op, t->data, x->data, 0, mass));
    LQCHK(dw_apply_raw(op, q->data, t->data, 1, mass));
    HIPCHK(hipMemcpyAsync(r->data, b->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, q->data, r->data, n));
    HIPCHK(hipMemcpyAsync(p->data, r->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    double rr = 0;
    LQCHK(blas_norm2(c, r->data, n, &rr, true));
    double init[9] = {rr, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int it = 0;
    bool converged = rr < eps;
    const int check_every = 8;
    auto upd_ptr = [&](lqcd_spinor_s* f, int par) { return f->data + (size_t)par * (slice / 2); };
    while (!converged && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            apply_bc(c, op->bc);
            StencilCall s1;
            if (!dw_batched_call(op, t->data, p->data, 0, mass, s1)) { set_error("Domainwall CG: the five-dimensional launch stopped applying inside a solve"); return LQCD_ERR_UNSUPPORTED; }
            s1.norm_partial = op->dw_partial;
            s1.skip_flag = c->d_scal;
            LQCHK(stencil_apply(c, s1));
            LQCHK(reduce_to_slot(c, nparts, 1, S_PQ, true, 1, op->dw_partial));          // + alpha = rr / pq
            StencilCall s2;
            if (!dw_batched_call(op, q->data, t->data, 1, mass, s2)) { set_error("Domainwall CG: the five-dimensional launch stopped applying inside a solve"); return LQCD_ERR_UNSUPPORTED; }
            s2.norm_partial = op->dw_partial;
            s2.upd_scal = c->d_scal;
            s2.upd[0] = upd_ptr(r, 0); s2.upd[1] = upd_ptr(r, 1);                         // slice 0 of r; the launch adds the slice offsets
            LQCHK(stencil_apply(c, s2));
            LQCHK(reduce_to_slot(c, nparts, 1, S_RRNEW, true, 2, op->dw_partial));       // + beta, stopping test, iteration count
            LQCHK(cg_launch_update_xp(c, x->data, p->data, r->data, n));
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        it = (int)c->h_scal[S_ITERS - S_RR];
        if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
        if (!std::isfinite(rr)) { set_error("CG: residual is not finite"); return LQCD_ERR_NOT_CONVERGED; }
    }
    if (iters) *iters = it;
    if (rr_out) *rr_out = rr;
    if (!converged) {
        set_error("The CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

}  // namespace

namespace lqcd {

int dw_op_apply(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger) {
    LQCHK(dw_check(op, out, in, "lqcd_op_apply"));
    HIPCHK(hipSetDevice(op->ctx->device));
    LQCHK(dw_apply_raw(op, out->data, in->data, dagger ? 1 : 0, op->km));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return LQCD_OK;
}
int dw_op_apply_DdagD(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in) {
    LQCHK(dw_check(op, out, in, "lqcd_op_apply_DdagD"));
    HIPCHK(hipSetDevice(op->ctx->device));
    lqcd_spinor_s* t;
    LQCHK(dw_work(op, 3, &t));
    LQCHK(dw_apply_raw(op, t->data, in->data, 0, op->km));
    LQCHK(dw_apply_raw(op, out->data, t->data, 1, op->km));
    HIPCHK(hipStreamSynchronize(op->ctx->stream));
    return LQCD_OK;
}
// solve_DinvX!(x, DdagD, b): x holds the initial guess
int dw_solve_cg(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double eps, int maxiter, int* iters, double* rr) {
    LQCHK(dw_check(op, x, b, "lqcd_solve_cg_DdagD"));
    HIPCHK(hipSetDevice(op->ctx->device));
    return dw_solve(op, x, b, op->km, eps, maxiter, iters, rr);
}

// sample_pseudofermions!(phi, U, fa, xi): phi = D_PV^-+ D^+ xi = D_PV (D_PV^+ D_PV)^-1 D^+ xi
int dw_sample(lqcd_op_s* op, lqcd_spinor_s* phi, lqcd_spinor_s* xi, double eps, int maxiter) {
    LQCHK(dw_check(op, phi, xi, "sample_pseudofermions (Domainwall)"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    lqcd_spinor_s *w, *z;
    LQCHK(dw_work(op, 4, &w)); LQCHK(dw_work(op, 5, &z));
    LQCHK(dw_apply_raw(op, w->data, xi->data, 1, op->km));
    HIPCHK(hipMemsetAsync(z->data, 0, z->elems * sizeof(double2), c->stream));
    LQCHK(dw_solve(op, z, w, 1.0, eps, maxiter, nullptr, nullptr));
    LQCHK(dw_apply_raw(op, phi->data, z->data, 0, 1.0));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// evaluate_FermiAction(fa, U, phi): S = psi^+ (D^+D)^-1 psi, psi = D_PV^+ phi.  X / Y (may be null) receive (D^+D)^-1 psi / D X
int dw_action(lqcd_op_s* op, lqcd_spinor_s* phi, lqcd_spinor_s* X, lqcd_spinor_s* Y, double eps, int maxiter, double* Sf, int* iters) {
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    lqcd_spinor_s* psi;
    LQCHK(dw_work(op, 4, &psi));
    if (!X) LQCHK(dw_work(op, 5, &X));
    LQCHK(dw_check(op, X, phi, "evaluate_FermiAction (Domainwall)"));
    if (Y) LQCHK(dw_check(op, Y, phi, "evaluate_FermiAction (Domainwall)"));
    ARGCHK(X != Y, "evaluate_FermiAction (Domainwall): X and Y must be distinct fields");
    LQCHK(dw_apply_raw(op, psi->data, phi->data, 1, 1.0));
    HIPCHK(hipMemsetAsync(X->data, 0, X->elems * sizeof(double2), c->stream));
    LQCHK(dw_solve(op, X, psi, op->km, eps, maxiter, iters, nullptr));
    if (Y) LQCHK(dw_apply_raw(op, Y->data, X->data, 0, op->km));
    double re = 0, im = 0;
    LQCHK(blas_dot(c, psi->data, X->data, X->elems, &re, &im, true));
    if (Sf) *Sf = re;
    return LQCD_OK;
}

// calc_UdSfdU!(UdSfdU, fa, U, phi): out = G in the convention of lqcd_fermion_force
int dw_force(lqcd_op_s* op, lqcd_gauge_s* out, lqcd_spinor_s* phi, double eps, int maxiter, double* Sf, int* iters) {
    lqcd_ctx_s* c = op->ctx;
    lqcd_spinor_s *X, *Y;
    LQCHK(dw_work(op, 5, &X)); LQCHK(dw_work(op, 6, &Y));
    LQCHK(dw_action(op, phi, X, Y, eps, maxiter, Sf, iters));
    LQCHK(blas_axpy(c, -1.0, 0.0, phi->data, Y->data, Y->elems));      // Z = Y - phi
    HIPCHK(hipStreamSynchronize(c->stream));
    lqcd_op_s* w = op->dw_wilson;
    w->gauge = op->gauge;
    for (int s = 0; s < op->L5; s++) {
        lqcd_spinor_s xs = slice_view(X, s), zs = slice_view(Y, s);
        LQCHK(lqcd_fermion_force_acc(w, out, &xs, &zs, 1.0, s > 0));
    }
    return LQCD_OK;
}

}  // namespace lqcd

// Initialize_pseudofermion_fields(U[1], "Domainwall", L5 = L5, nowing = true) (universe.jl:128)
extern "C" int lqcd_spinor_create_5d(lqcd_ctx_t ctx, lqcd_spinor_t* s, int L5) {
    ARGCHK(ctx && s && L5 >= 2, "lqcd_spinor_create_5d: need a context and L5 >= 2");
    HIPCHK(hipSetDevice(ctx->device));
    lqcd_spinor_s* x = new lqcd_spinor_s;
    x->ctx = ctx;
    x->kind = LQCD_DOMAINWALL;
    x->subset = LQCD_FULL;
    x->ncomp = 12;
    x->ls = L5;
    x->elems = (size_t)L5 * 12 * ctx->geom.Vs * 2;
    x->data = nullptr;
    hipError_t e = hipMalloc((void**)&x->data, x->elems * sizeof(double2));
    if (e != hipSuccess) { delete x; return hip_fail(e, "hipMalloc(5-d spinor)", __FILE__, __LINE__); }
    e = hipMemsetAsync(x->data, 0, x->elems * sizeof(double2), ctx->stream);
    if (e != hipSuccess) { (void)hipFree(x->data); delete x; return hip_fail(e, "memset(5-d spinor)", __FILE__, __LINE__); }
    *s = x;
    return LQCD_OK;
}

// x.w[i5] of the reference's five-dimensional field: a Wilson field that ALIASES slice i5 (0-based) of s5 -- upload / download / fill / BLAS go through it.
// The view owns nothing; a parent destroyed while views exist keeps its storage until the last view is destroyed (finalizers of a garbage collector run in any order).
extern "C" int lqcd_spinor_slice(lqcd_spinor_t s5, int i5, lqcd_spinor_t* view) {
    ARGCHK(s5 && view && s5->kind == LQCD_DOMAINWALL && i5 >= 0 && i5 < s5->ls, "lqcd_spinor_slice: need a five-dimensional field and 0 <= i5 < L5");
    std::lock_guard<std::mutex> lk(lqcd::view_mutex());
    ARGCHK(!s5->zombie, "lqcd_spinor_slice: the five-dimensional field has been destroyed");
    lqcd_spinor_s* v = new lqcd_spinor_s;
    *v = slice_view(s5, i5);
    v->view_of = s5;
    v->nviews = 0;
    v->zombie = false;
    s5->nviews++;
    *view = v;
    return LQCD_OK;
}

// Dirac_operator(U, x, Dict("Dirac_operator" => "Domainwall", "mass" => m, "L5" => L5, "M" => M, ...)) (universe.jl:116-128, 137)
extern "C" int lqcd_op_create_domainwall(lqcd_ctx_t ctx, lqcd_op_t* op, lqcd_gauge_t g, double M, double mass, int L5, const int bc[4]) {
    ARGCHK(ctx && op && g && bc && L5 >= 2, "lqcd_op_create_domainwall: need a context, a gauge field, boundary conditions and L5 >= 2");
    ARGCHK(ctx->local_peers.empty(), "lqcd_op_create_domainwall: not available on an in-process PE grid (RCCL ranks only)");
    lqcd_op_s* w = nullptr;
    LQCHK(lqcd_op_create(ctx, &w, LQCD_WILSON, g, 0.5, 1.0, bc));      // hop coefficient -1/2: D4 = (4 + M) - H/2
    lqcd_op_s* o = new lqcd_op_s;
    o->ctx = ctx; o->kind = LQCD_DOMAINWALL; o->gauge = g; o->km = mass; o->r = 1.0;
    for (int mu = 0; mu < 4; mu++) o->bc[mu] = bc[mu];
    o->L5 = L5; o->dw_M = M; o->dw_wilson = w;
    *op = o;
    return LQCD_OK;
}
