/*
 * c_abi_smoke.c -- drives liblqcd_hip.so from plain C (no Python, no torch): the drop-in boundary is a C ABI.
 * Built and run by tests/test_gpu_parity.py::test_c_abi_from_plain_c on the GPU box:
 *   gcc -std=c99 -I include tests/c_abi_smoke.c -o /tmp/c_abi_smoke -L latticeqcd.jl_amd/csrc -llqcd_hip -lm
 * Checks, without any oracle: cold-start plaquette = 1, free-field Wilson D on a constant spinor
 * (D psi = (1 - 8 kappa) psi for periodic BC and U = 1), gamma5-hermiticity on a hot start, CG true residual; and the reference's general-Nf
 * staggered action (FermiAction(D, Dict("Nf" => 3)), universe.jl:106-110,138, test/test_Nf3.toml:8) through the action handle alone: the fit
 * verifies on its interval with the right signs, the heat bath gives S_f = xi^+ xi, the force is finite.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "lqcd_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int st_ = (call);                                                           \
        if (st_ != LQCD_OK) {                                                       \
            fprintf(stderr, "%s failed (%d): %s\n", #call, st_, lqcd_last_error()); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

int main(void) {
    const int L[4] = {8, 8, 8, 8}, pe[4] = {1, 1, 1, 1}, bc_per[4] = {1, 1, 1, 1}, bc_apbc[4] = {1, 1, 1, -1};
    const double kappa = 0.141139;
    const long V = 8L * 8 * 8 * 8, n = 12 * V;
    if (lqcd_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }

    lqcd_ctx_t ctx;
    lqcd_gauge_t U;
    lqcd_spinor_t a, b, Da, Ddb, x, r;
    lqcd_op_t D;
    CHECK(lqcd_ctx_create(&ctx, 0, L, pe, 0));
    CHECK(lqcd_gauge_create(ctx, &U));
    CHECK(lqcd_gauge_unit(U));
    double plaq;
    CHECK(lqcd_gauge_plaquette(U, &plaq));
    if (fabs(plaq - 1.0) > 1e-14) { fprintf(stderr, "cold plaquette %.17g\n", plaq); return 1; }

    CHECK(lqcd_spinor_create(ctx, &a, LQCD_WILSON, LQCD_FULL));
    CHECK(lqcd_spinor_create(ctx, &b, LQCD_WILSON, LQCD_FULL));
    CHECK(lqcd_spinor_create(ctx, &Da, LQCD_WILSON, LQCD_FULL));
    CHECK(lqcd_spinor_create(ctx, &Ddb, LQCD_WILSON, LQCD_FULL));
    CHECK(lqcd_spinor_create(ctx, &x, LQCD_WILSON, LQCD_FULL));
    CHECK(lqcd_spinor_create(ctx, &r, LQCD_WILSON, LQCD_FULL));

    /* free field, constant spinor, periodic BC: D psi = (1 - 8 kappa) psi  (r = 1: sum_nu 2 r cos 0 = 8) */
    double* host = (double*)malloc(sizeof(double) * 2 * n);
    for (long i = 0; i < n; i++) { host[2 * i] = 1.0 + (double)(i % 12); host[2 * i + 1] = -0.5 * (double)(i % 12); }
    /* reference layout is ic + 3*(site + V*is): make the value depend on (ic,is) only -> constant over sites */
    for (int is = 0; is < 4; is++)
        for (long s = 0; s < V; s++)
            for (int ic = 0; ic < 3; ic++) {
                long idx = ic + 3 * (s + V * is);
                host[2 * idx] = 1.0 + ic + 3 * is;
                host[2 * idx + 1] = 0.25 * (ic - is);
            }
    CHECK(lqcd_spinor_upload(a, host));
    CHECK(lqcd_op_create(ctx, &D, LQCD_WILSON, U, kappa, 1.0, bc_per));
    CHECK(lqcd_op_apply(D, Da, a, 0));
    double* out = (double*)malloc(sizeof(double) * 2 * n);
    CHECK(lqcd_spinor_download(Da, out));
    double maxerr = 0;
    for (long i = 0; i < 2 * n; i++) {
        double e = fabs(out[i] - (1.0 - 8.0 * kappa) * host[i]);
        if (e > maxerr) maxerr = e;
    }
    if (maxerr > 1e-13) { fprintf(stderr, "free-field check failed: %.3e\n", maxerr); return 1; }
    CHECK(lqcd_op_destroy(D));

    /* hot start: <a, D b> = conj <b, D^+ a>, then CG with an independent residual check */
    CHECK(lqcd_gauge_hot_start(U, 111));
    CHECK(lqcd_op_create(ctx, &D, LQCD_WILSON, U, kappa, 1.0, bc_apbc));
    CHECK(lqcd_spinor_gaussian(a, 1));
    CHECK(lqcd_spinor_gaussian(b, 2));
    CHECK(lqcd_op_apply(D, Da, b, 0));      /* Da := D b   */
    CHECK(lqcd_op_apply(D, Ddb, a, 1));     /* Ddb := D^+ a */
    double l_re, l_im, r_re, r_im;
    CHECK(lqcd_dot(a, Da, &l_re, &l_im));
    CHECK(lqcd_dot(b, Ddb, &r_re, &r_im));
    if (fabs(l_re - r_re) > 1e-10 * fabs(l_re) || fabs(l_im + r_im) > 1e-10 * (fabs(l_re) + fabs(l_im))) {
        fprintf(stderr, "gamma5-hermiticity failed: (%.15g,%.15g) vs conj(%.15g,%.15g)\n", l_re, l_im, r_re, r_im);
        return 1;
    }
    int iters = 0;
    double rr = 0, res2 = 0;
    CHECK(lqcd_spinor_zero(x));
    CHECK(lqcd_solve_cg_DdagD(D, x, b, 1e-19, 3000, &iters, &rr));
    CHECK(lqcd_op_apply_DdagD(D, r, x));
    CHECK(lqcd_axpy(-1.0, 0.0, b, r));
    CHECK(lqcd_norm2(r, &res2));
    if (!(rr < 1e-19) || !(res2 < 1e-18)) { fprintf(stderr, "CG failed: rr %.3e true %.3e\n", rr, res2); return 1; }
    /* error path: maxiter too small -> LQCD_ERR_NOT_CONVERGED with a message */
    CHECK(lqcd_spinor_zero(x));
    if (lqcd_solve_cg_DdagD(D, x, b, 1e-19, 2, &iters, &rr) != LQCD_ERR_NOT_CONVERGED) { fprintf(stderr, "expected non-convergence\n"); return 1; }

    /* FermiAction(D, Dict("Nf" => 3)) with the staggered operator of the reference's test_Nf3.toml (mass 0.5): everything through lqcd_action_* */
    {
        lqcd_op_t Ds;
        lqcd_action_t fa;
        lqcd_spinor_t xi, eta;
        lqcd_gauge_t G;
        double v = 0, a0 = 0, res[40], poles[40], err = 0, S = 0, xx = 0, xim = 0;
        int np = 0;
        CHECK(lqcd_op_create(ctx, &Ds, LQCD_STAGGERED, U, 0.5, 1.0, bc_apbc));
        CHECK(lqcd_action_create(Ds, 3.0, 1e-20, 3000, 0, NULL, NULL, &fa));
        CHECK(lqcd_action_get(fa, "rational", &v));
        if (v != 1.0) { fprintf(stderr, "Nf = 3 staggered must be the rational action\n"); return 1; }
        CHECK(lqcd_action_get(fa, "alpha", &v));
        if (fabs(v - 3.0 / 8.0) > 1e-15) { fprintf(stderr, "alpha %.17g\n", v); return 1; }
        CHECK(lqcd_action_coefficients(fa, 0, &a0, res, poles, 40, &np, &err));
        if (!(np >= 6 && np <= 30 && err <= 1e-12 && a0 >= 0)) { fprintf(stderr, "fit: n %d err %.2e a0 %.3e\n", np, err, a0); return 1; }
        for (int k = 0; k < np; k++)
            if (!(res[k] > 0 && poles[k] > 0)) { fprintf(stderr, "fit: residue / pole %d has the wrong sign\n", k); return 1; }
        for (int i = 0; i <= 1000; i++) {       /* x^(-3/8) on [0.25, 16.25] */
            const double xv = 0.25 * pow(16.25 / 0.25, i / 1000.0);
            double rv = a0;
            for (int k = 0; k < np; k++) rv += res[k] / (xv + poles[k]);
            if (fabs(rv * pow(xv, 0.375) - 1.0) > 1e-11) { fprintf(stderr, "fit off by %.2e at %.4f\n", fabs(rv * pow(xv, 0.375) - 1.0), xv); return 1; }
        }
        CHECK(lqcd_spinor_create(ctx, &xi, LQCD_STAGGERED, LQCD_FULL));
        CHECK(lqcd_spinor_create(ctx, &eta, LQCD_STAGGERED, LQCD_FULL));
        CHECK(lqcd_gauge_create(ctx, &G));
        CHECK(lqcd_action_gauss_sampling(fa, xi, 77));
        CHECK(lqcd_action_sample_pseudofermions(fa, U, eta, xi));
        CHECK(lqcd_action_evaluate(fa, U, eta, NULL, NULL, &S, NULL));
        CHECK(lqcd_dot(xi, xi, &xx, &xim));
        if (fabs(S / xx - 1.0) > 1e-9) { fprintf(stderr, "heat bath: S_f %.15g, xi.xi %.15g\n", S, xx); return 1; }
        CHECK(lqcd_action_force(fa, U, G, eta, NULL, NULL));
        /* Nf outside (0, 8) is refused with the reference-style message; Nf = 4 is the exact even-site action */
        lqcd_action_t bad;
        if (lqcd_action_create(Ds, 9.0, 1e-20, 3000, 0, NULL, NULL, &bad) != LQCD_ERR_UNSUPPORTED) { fprintf(stderr, "Nf = 9 accepted\n"); return 1; }
        CHECK(lqcd_action_create(Ds, 4.0, 1e-20, 3000, 0, NULL, NULL, &bad));
        CHECK(lqcd_action_get(bad, "rational", &v));
        if (v != 0.0) { fprintf(stderr, "Nf = 4 staggered is an exact action\n"); return 1; }
        lqcd_action_destroy(bad);
        lqcd_action_destroy(fa);
        lqcd_spinor_destroy(xi); lqcd_spinor_destroy(eta);
        lqcd_gauge_destroy(G);
        lqcd_op_destroy(Ds);
    }

    /* The reference's U_update! (AbstractMD.jl:89-93) calls exptU! -> mul! -> substitute_U! per direction: the library records the first two and
     * fuses at the third (tunable lazy_links).  An INTERRUPTED triple -- the temporary is read in between -- must hold what the eager calls put
     * there, and a completed one must update the links like the eager sequence. */
    {
        lqcd_gauge_t P, T1, T2, Ua, Ub;
        const long ng = 4L * V * 9;
        double* g1 = (double*)malloc(sizeof(double) * 2 * ng);
        double* g2 = (double*)malloc(sizeof(double) * 2 * ng);
        int open = -1, deferred = -1;
        CHECK(lqcd_gauge_create(ctx, &P)); CHECK(lqcd_gauge_create(ctx, &T1)); CHECK(lqcd_gauge_create(ctx, &T2));
        CHECK(lqcd_gauge_create(ctx, &Ua)); CHECK(lqcd_gauge_create(ctx, &Ub));
        CHECK(lqcd_momentum_gaussian(P, 9));
        CHECK(lqcd_gauge_copy(Ua, U)); CHECK(lqcd_gauge_copy(Ub, U));
        CHECK(lqcd_gauge_unit(T1)); CHECK(lqcd_gauge_unit(T2));
        CHECK(lqcd_ctx_get_param(ctx, "lazy_links", &open));
        if (open != 0) { fprintf(stderr, "the plain C ABI must be eager by default (lazy_links = %d)\n", open); return 1; }
        CHECK(lqcd_ctx_set_param(ctx, "lazy_links", 1));              /* what the Julia / Python bindings do when they create a context */
        CHECK(lqcd_link_exp(T1, 0, 0.3, P, 1));                       /* recorded */
        CHECK(lqcd_ctx_get_param(ctx, "lazy_open", &open));
        if (open != 1) { fprintf(stderr, "lqcd_link_exp was not recorded (lazy_open = %d)\n", open); return 1; }
        CHECK(lqcd_link_mul(T1, 1, T1, 0, Ua, 2));                    /* recorded: W = expU U[2] */
        CHECK(lqcd_gauge_download(T1, g1, LQCD_LAYOUT_REFERENCE));    /* interruption: both run now, in order */
        CHECK(lqcd_ctx_get_param(ctx, "lazy_open", &open));
        if (open != 0) { fprintf(stderr, "reading a temporary did not flush the record\n"); return 1; }
        CHECK(lqcd_ctx_set_param(ctx, "lazy_links", 0));
        CHECK(lqcd_link_exp(T2, 0, 0.3, P, 1));
        CHECK(lqcd_link_mul(T2, 1, T2, 0, Ub, 2));
        CHECK(lqcd_gauge_download(T2, g2, LQCD_LAYOUT_REFERENCE));
        for (long i = 0; i < 2 * ng; i++)
            if (g1[i] != g2[i]) { fprintf(stderr, "interrupted triple differs from the eager calls at %ld\n", i); return 1; }
        /* three directions eagerly on Ub; lazily on Ua (deferred, waiting for a fourth), then the plaquette asks for Ua: they run */
        for (int mu = 0; mu < 3; mu++) {
            CHECK(lqcd_link_exp(T2, 0, 0.1, P, mu)); CHECK(lqcd_link_mul(T2, 1, T2, 0, Ub, mu)); CHECK(lqcd_link_copy(Ub, mu, T2, 1));
        }
        CHECK(lqcd_ctx_set_param(ctx, "lazy_links", 1));
        for (int mu = 0; mu < 3; mu++) {
            CHECK(lqcd_link_exp(T1, 0, 0.1, P, mu)); CHECK(lqcd_link_mul(T1, 1, T1, 0, Ua, mu)); CHECK(lqcd_link_copy(Ua, mu, T1, 1));
        }
        CHECK(lqcd_ctx_get_param(ctx, "lazy_deferred", &deferred));
        if (deferred != 3) { fprintf(stderr, "expected three deferred triples, found %d\n", deferred); return 1; }
        double pa = 0, pb = 0;
        CHECK(lqcd_gauge_plaquette(Ua, &pa));
        CHECK(lqcd_gauge_plaquette(Ub, &pb));
        CHECK(lqcd_ctx_get_param(ctx, "lazy_deferred", &deferred));
        if (deferred != 0 || fabs(pa - pb) > 1e-13) { fprintf(stderr, "deferred triples: %d left, plaquettes %.16g vs %.16g\n", deferred, pa, pb); return 1; }
        free(g1); free(g2);
        lqcd_gauge_destroy(P); lqcd_gauge_destroy(T1); lqcd_gauge_destroy(T2); lqcd_gauge_destroy(Ua); lqcd_gauge_destroy(Ub);
    }

    /* Dirac_operator = "Domainwall" (universe.jl:116-128) from plain C, no oracle: on unit links with periodic boundaries a field that is constant in all
     * five directions has H psi = 8 psi, so D4 psi = (4 + M) psi - 4 psi = M psi; with m = -1 the fifth direction is periodic too and
     * D5 psi = M psi + psi - P_- psi - P_+ psi = M psi.  Then the pseudofermion action through the handle: at m = 1 (the reference's test) D = D_PV and
     * S = phi^+ phi on any links. */
    {
        const int L5 = 4;
        lqcd_gauge_t Uc; lqcd_spinor_t p5, q5, v; lqcd_op_t D5; lqcd_action_t fa5;
        CHECK(lqcd_gauge_create(ctx, &Uc)); CHECK(lqcd_gauge_unit(Uc));
        CHECK(lqcd_spinor_create_5d(ctx, &p5, L5)); CHECK(lqcd_spinor_create_5d(ctx, &q5, L5));
        /* `host` still holds the site-constant spinor of the free-field check above */
        for (int i5 = 0; i5 < L5; i5++) { CHECK(lqcd_spinor_slice(p5, i5, &v)); CHECK(lqcd_spinor_upload(v, host)); CHECK(lqcd_spinor_destroy(v)); }
        CHECK(lqcd_op_create_domainwall(ctx, &D5, Uc, -1.3, -1.0, L5, bc_per));
        CHECK(lqcd_op_apply(D5, q5, p5, 0));
        double dwerr = 0;
        for (int i5 = 0; i5 < L5; i5++) {
            CHECK(lqcd_spinor_slice(q5, i5, &v)); CHECK(lqcd_spinor_download(v, out)); CHECK(lqcd_spinor_destroy(v));
            for (long i = 0; i < 2 * n; i++) { double e = fabs(out[i] + 1.3 * host[i]); if (e > dwerr) dwerr = e; }
        }
        if (dwerr > 1e-13) { fprintf(stderr, "Domainwall free-field check failed: %.3e\n", dwerr); return 1; }
        CHECK(lqcd_op_destroy(D5));
        CHECK(lqcd_op_create_domainwall(ctx, &D5, U, -1.0, 1.0, L5, bc_apbc));      /* hot links, m = 1 */
        CHECK(lqcd_action_create(D5, 0.0, 1e-20, 3000, 0, NULL, NULL, &fa5));
        CHECK(lqcd_spinor_gaussian(p5, 77));
        double S5 = 0, n5 = 0; int it5 = 0;
        CHECK(lqcd_action_evaluate(fa5, U, p5, NULL, NULL, &S5, &it5));
        CHECK(lqcd_norm2(p5, &n5));
        if (fabs(S5 - n5) > 1e-9 * n5) { fprintf(stderr, "Domainwall action at the Pauli-Villars mass: %.15g vs phi.phi %.15g\n", S5, n5); return 1; }
        if (lqcd_solve_bicgstab(D5, q5, p5, 0, 1e-19, 100, NULL, NULL) != LQCD_ERR_UNSUPPORTED) { fprintf(stderr, "expected LQCD_ERR_UNSUPPORTED\n"); return 1; }
        lqcd_action_destroy(fa5); lqcd_op_destroy(D5);
        lqcd_spinor_destroy(p5); lqcd_spinor_destroy(q5); lqcd_gauge_destroy(Uc);
    }

    printf("C_ABI_OK plaquette=1 free-field maxerr=%.2e CG iters ok true-res=%.2e msg=\"%s\"\n", maxerr, res2, lqcd_last_error());
    free(host); free(out);
    lqcd_op_destroy(D);
    lqcd_spinor_destroy(a); lqcd_spinor_destroy(b); lqcd_spinor_destroy(Da); lqcd_spinor_destroy(Ddb);
    lqcd_spinor_destroy(x); lqcd_spinor_destroy(r);
    lqcd_gauge_destroy(U);
    lqcd_ctx_destroy(ctx);
    return 0;
}
