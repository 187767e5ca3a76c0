// actions.hip -- pseudofermion action and force entry points (2-flavour and rational), all fields resident.
// evaluate_FermiAction /root/reference/src/updates/standardHMC.jl:71, calc_UdSfdU!
// src/md/AbstractMD.jl:129, the general-Nf actions of test/test_Nf2.toml:8, test/test_Nf3.toml:8.
#include "ops_internal.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <functional>

using namespace lqcd;

// ---------------------------------------------------------------------------------- pseudofermion action and force
// (evaluate_FermiAction standardHMC.jl:71, calc_UdSfdU! AbstractMD.jl:129)
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
    bool have_Y = false;
    if (even_only) LQCHK(lqcd_solve_cg_DdagD_parity(op, X, eta, 0, eps, maxiter, iters, nullptr));
    // mixed_action_solver: Wilson and Wilson-clover (r = 1, one rank) take the even-odd route below with the fp32 inner chain (bicg_mixed); everything else the mixed-precision CG
    else if (c->tun.mixed_action_solver && !(op->kind == LQCD_WILSON && c->tun.action_eo_solver && op->r == 1.0 && c->tun.dslash_variant == 1 && !any_partitioned(c) && !c->has_comm))
        LQCHK(lqcd_solve_mixed_cg_DdagD(op, X, eta, eps, maxiter, 0.0, iters, nullptr, nullptr));
    else {
        // Wilson(-clover), tunable action_eo_solver: X = (D^+D)^-1 eta as TWO even-odd preconditioned BiCGStab solves, Y = D^-+ eta and X = D^-1 Y --
        // the Schur complement converges in a fraction of the normal equations' iterations (32^3x64 hot start: 2 x ~13 iterations of 2 Dslash
        // against 75 of 2), and Y = D X, which the force needs anyway, comes out of the first solve.  The reference's stopping rule is kept for the
        // system it states it for: with |eta - D^+ Y|^2 < eps/4 and |Y - D X|^2 < eps / (4 |D|^2), |D| <= 1 + 8 kappa (+ 6 kappa c_sw),
        // |eta - D^+D X| <= |eta - D^+ Y| + |D^+| |Y - D X| < sqrt(eps).  A BiCGStab that breaks down or stalls falls back to the CG.
        lqcd_spinor_s* Yw = Y;
        const bool try_eo = op->kind == LQCD_WILSON && c->tun.action_eo_solver && c->local_peers.empty();
        if (try_eo && !Yw) Yw = scratch_get(c, op->kind, LQCD_FULL);
        int st = LQCD_ERR_NOT_CONVERGED, it1 = 0, it2 = 0;
        if (try_eo && Yw) {
            const double nD = 1.0 + 8.0 * std::fabs(op->km) + (op->csw != 0.0 ? 6.0 * std::fabs(op->km * op->csw) : 0.0);
            HIPCHK(hipMemsetAsync(Yw->data, 0, Yw->elems * sizeof(double2), c->stream));
            const int mixed0 = c->tun.bicg_mixed;
            if (c->tun.mixed_action_solver) c->tun.bicg_mixed = 1;
            // Wilson-clover: the even-odd solver iterates on the A_ee^-1-preconditioned Schur system, whose residual is A_ee^-1 times the residual of the
            // system the rule is stated for -- the true residual can exceed the solver's by |A| <= 1 + 6 kappa c_sw: both targets tighten by |A|^2
            const double nA = op->csw != 0.0 ? 1.0 + 6.0 * std::fabs(op->km * op->csw) : 1.0;
            c->zero_guess_hint = true;      // Yw and X were cleared above: the solver need neither test that nor apply M to zero
            st = lqcd_solve_bicgstab_eo(op, Yw, eta, 1, 0.25 * eps / (nA * nA), maxiter, &it1, nullptr);
            if (st == LQCD_OK) st = lqcd_solve_bicgstab_eo(op, X, Yw, 0, 0.25 * eps / (nD * nD * nA * nA), maxiter, &it2, nullptr);
            c->zero_guess_hint = false;
            c->tun.bicg_mixed = mixed0;
            if (st != LQCD_OK && st != LQCD_ERR_NOT_CONVERGED) { if (Yw != Y) scratch_put(Yw); return st; }
        }
        if (Yw && Yw != Y) scratch_put(Yw);
        if (st == LQCD_OK) { have_Y = Y != nullptr; if (iters) *iters = it1 + it2; }
        else {
            HIPCHK(hipMemsetAsync(X->data, 0, X->elems * sizeof(double2), c->stream));
            LQCHK(cg_run(op, X, eta, eps, maxiter, false, iters, nullptr));
        }
    }
    if (Y && !have_Y) LQCHK(op_apply_async(op, Y, X, 0, nullptr));
    if (Sf) {      // (the force evaluation asks for X and Y only: no inner product, no read-back)
        double re = 0, im = 0;
        LQCHK(blas_dot(c, eta->data, X->data, eta->elems, &re, &im, true));
        *Sf = re;
    }
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
    if (c->tun.mixed_action_solver == 2)     // one fp32 multi-shift pass for all poles, fp64 defect correction per pole (mixed.hip)
        return lqcd_solve_multishift_mixed_cg(op, nullptr, xs.data(), b, poles, n, eps, maxiter, 0.0, iters, nullptr, nullptr);
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
