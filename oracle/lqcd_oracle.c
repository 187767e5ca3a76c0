/*
 * lqcd_oracle.c -- CPU oracle (plain C99).  TEST INFRASTRUCTURE ONLY; see lqcd_oracle.h.
 *
 * PARITY UNPINNED at the Dslash/CG level (the reference's arithmetic is in un-vendored Julia
 * packages); pinned against the reference's gauge fixtures for formats, index
 * order and plaquette.  Deliberately written differently from the device kernels: explicit 4x4
 * gamma matrices (no spin-projection trick), reference host layout, one serial site loop in the
 * reference's (it,iz,iy,ix) order with colour innermost, D^dagger literally as
 * gamma5 D gamma5.
 */
#include "lqcd_oracle.h"
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double complex cplx;

static int g_threads = 1;
void orc_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int orc_get_threads(void) { return g_threads; }

/* gamma matrices, SURVEY.md Appendix A (LTK basis, all Hermitian, gamma5 = g1 g2 g3 g4) */
static void gamma_mat(int nu, cplx g[4][4]) {
    memset(g, 0, 16 * sizeof(cplx));
    switch (nu) {
    case 0: g[0][3] = -I; g[1][2] = -I; g[2][1] = I;  g[3][0] = I;  break;
    case 1: g[0][3] = -1; g[1][2] = 1;  g[2][1] = 1;  g[3][0] = -1; break;
    case 2: g[0][2] = -I; g[1][3] = I;  g[2][0] = I;  g[3][1] = -I; break;
    case 3: g[0][0] = 1;  g[1][1] = 1;  g[2][2] = -1; g[3][3] = -1; break;
    default: /* gamma5 */
        g[0][2] = -1; g[1][3] = -1; g[2][0] = -1; g[3][1] = -1; break;
    }
}

static inline long site_of(const int L[4], int x, int y, int z, int t) {
    return x + (long)L[0] * (y + (long)L[1] * (z + (long)L[2] * t));
}
static inline long vol(const int L[4]) { return (long)L[0] * L[1] * L[2] * L[3]; }

/* neighbour site in direction nu (sign = +1/-1); *wrapped set when the global boundary is crossed */
static inline long neigh(const int L[4], const int c[4], int nu, int sign, int* wrapped) {
    int d[4] = {c[0], c[1], c[2], c[3]};
    d[nu] += sign;
    *wrapped = 0;
    if (d[nu] >= L[nu]) { d[nu] -= L[nu]; *wrapped = 1; }
    if (d[nu] < 0)      { d[nu] += L[nu]; *wrapped = 1; }
    return site_of(L, d[0], d[1], d[2], d[3]);
}

#define UIDX(V, mu, s, a, b) ((a) + 3 * ((b) + 3 * ((s) + (V) * (long)(mu))))
#define PIDX(V, s, c, sp) ((c) + 3 * ((s) + (V) * (long)(sp)))

/* ------------------------------------------------------------------ plaquette / unitarity */
static void load_link(cplx M[3][3], const cplx* U, long V, int mu, long s) {
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) M[a][b] = U[UIDX(V, mu, s, a, b)];
}
static void mm(cplx C[3][3], cplx A[3][3], cplx B[3][3]) {
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            cplx t = 0;
            for (int k = 0; k < 3; k++) t += A[a][k] * B[k][b];
            C[a][b] = t;
        }
}
static void mmd(cplx C[3][3], cplx A[3][3], cplx B[3][3]) { /* C = A B^dagger */
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            cplx t = 0;
            for (int k = 0; k < 3; k++) t += A[a][k] * conj(B[b][k]);
            C[a][b] = t;
        }
}

double orc_plaquette(const double* Ud, const int L[4]) {
    const cplx* U = (const cplx*)Ud;
    long V = vol(L);
    double sum = 0;
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t}, w;
                    long s = site_of(L, x, y, z, t);
                    for (int mu = 0; mu < 4; mu++)
                        for (int nu = mu + 1; nu < 4; nu++) {
                            cplx A[3][3], B[3][3], C[3][3], D[3][3], T1[3][3], T2[3][3], T3[3][3];
                            load_link(A, U, V, mu, s);
                            load_link(B, U, V, nu, neigh(L, c, mu, 1, &w));
                            load_link(C, U, V, mu, neigh(L, c, nu, 1, &w));
                            load_link(D, U, V, nu, s);
                            mm(T1, A, B);
                            mmd(T2, T1, C);
                            mmd(T3, T2, D);
                            sum += creal(T3[0][0] + T3[1][1] + T3[2][2]);
                        }
                }
    return sum / (6.0 * (double)V * 3.0);
}

double orc_unitarity_dev(const double* Ud, const int L[4]) {
    const cplx* U = (const cplx*)Ud;
    long V = vol(L);
    double dev = 0;
    for (int mu = 0; mu < 4; mu++)
        for (long s = 0; s < V; s++) {
            cplx A[3][3], T[3][3];
            load_link(A, U, V, mu, s);
            mmd(T, A, A);
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) {
                    double d = cabs(T[a][b] - (a == b ? 1.0 : 0.0));
                    if (d > dev) dev = d;
                }
        }
    return dev;
}

/* ------------------------------------------------------------------ Wilson */
/* hop(n) = sum_nu [ (r - sg*g_nu) U_nu(n) x(n+nu) + (r + sg*g_nu) U_nu^+(n-nu) x(n-nu) ],  sg = +1 */
static void wilson_hop_site(cplx acc[4][3], const cplx* U, const cplx* in, const int L[4], long V,
                            const int c[4], double r, const int bc[4], cplx G[4][4][4]) {
    long s = site_of(L, c[0], c[1], c[2], c[3]);
    for (int sp = 0; sp < 4; sp++)
        for (int a = 0; a < 3; a++) acc[sp][a] = 0;
    for (int nu = 0; nu < 4; nu++) {
        int w;
        cplx tmp[4][3];
        /* forward */
        long np = neigh(L, c, nu, 1, &w);
        double sgn = w ? (double)bc[nu] : 1.0;
        for (int sp = 0; sp < 4; sp++)
            for (int a = 0; a < 3; a++) {
                cplx t = 0;
                for (int b = 0; b < 3; b++) t += U[UIDX(V, nu, s, a, b)] * in[PIDX(V, np, b, sp)];
                tmp[sp][a] = sgn * t;
            }
        for (int sp = 0; sp < 4; sp++)
            for (int a = 0; a < 3; a++) {
                cplx t = r * tmp[sp][a];
                for (int s2 = 0; s2 < 4; s2++) t -= G[nu][sp][s2] * tmp[s2][a];
                acc[sp][a] += t;
            }
        /* backward */
        long nm = neigh(L, c, nu, -1, &w);
        sgn = w ? (double)bc[nu] : 1.0;
        for (int sp = 0; sp < 4; sp++)
            for (int a = 0; a < 3; a++) {
                cplx t = 0;
                for (int b = 0; b < 3; b++) t += conj(U[UIDX(V, nu, nm, b, a)]) * in[PIDX(V, nm, b, sp)];
                tmp[sp][a] = sgn * t;
            }
        for (int sp = 0; sp < 4; sp++)
            for (int a = 0; a < 3; a++) {
                cplx t = r * tmp[sp][a];
                for (int s2 = 0; s2 < 4; s2++) t += G[nu][sp][s2] * tmp[s2][a];
                acc[sp][a] += t;
            }
    }
}

static void apply_gamma5(cplx* out, const cplx* in, long V) {
    /* gamma5 = [[0,0,-1,0],[0,0,0,-1],[-1,0,0,0],[0,-1,0,0]] */
    for (long s = 0; s < V; s++)
        for (int a = 0; a < 3; a++) {
            cplx p0 = in[PIDX(V, s, a, 0)], p1 = in[PIDX(V, s, a, 1)], p2 = in[PIDX(V, s, a, 2)],
                 p3 = in[PIDX(V, s, a, 3)];
            out[PIDX(V, s, a, 0)] = -p2;
            out[PIDX(V, s, a, 1)] = -p3;
            out[PIDX(V, s, a, 2)] = -p0;
            out[PIDX(V, s, a, 3)] = -p1;
        }
}

static void wilson_D_plain(cplx* out, const cplx* U, const cplx* in, const int L[4], double kappa, double r,
                           const int bc[4]) {
    long V = vol(L);
    cplx G[4][4][4];
    for (int nu = 0; nu < 4; nu++) gamma_mat(nu, G[nu]);
#ifdef _OPENMP
#pragma omp parallel for collapse(3) num_threads(g_threads) schedule(static)      /* rows of x: the partition par_range() reproduces for the vectors */
#endif
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t};
                    long s = site_of(L, x, y, z, t);
                    cplx acc[4][3];
                    wilson_hop_site(acc, U, in, L, V, c, r, bc, G);
                    for (int sp = 0; sp < 4; sp++)
                        for (int a = 0; a < 3; a++)
                            out[PIDX(V, s, a, sp)] = in[PIDX(V, s, a, sp)] - kappa * acc[sp][a];
                }
}

void orc_wilson_D(double* outd, const double* Ud, const double* ind, const int L[4], double kappa, double r,
                  const int bc[4], int dagger) {
    long V = vol(L);
    cplx* out = (cplx*)outd;
    const cplx* U = (const cplx*)Ud;
    const cplx* in = (const cplx*)ind;
    if (!dagger) {
        wilson_D_plain(out, U, in, L, kappa, r, bc);
        return;
    }
    /* D^dagger = gamma5 D gamma5 (Appendix A) */
    cplx* t1 = (cplx*)malloc(sizeof(cplx) * 12 * V);
    cplx* t2 = (cplx*)malloc(sizeof(cplx) * 12 * V);
    apply_gamma5(t1, in, V);
    wilson_D_plain(t2, U, t1, L, kappa, r, bc);
    apply_gamma5(out, t2, V);
    free(t1);
    free(t2);
}

void orc_wilson_hop_parity(double* outd, const double* Ud, const double* ind, const int L[4], double r,
                           const int bc[4], int dagger, int out_parity) {
    long V = vol(L);
    cplx* out = (cplx*)outd;
    const cplx* U = (const cplx*)Ud;
    const cplx* in0 = (const cplx*)ind;
    cplx G[4][4][4];
    for (int nu = 0; nu < 4; nu++) gamma_mat(nu, G[nu]);
    cplx* in = (cplx*)in0;
    cplx* t1 = NULL;
    if (dagger) { /* gamma5 H gamma5 */
        t1 = (cplx*)malloc(sizeof(cplx) * 12 * V);
        apply_gamma5(t1, in0, V);
        in = t1;
    }
    cplx* res = dagger ? (cplx*)malloc(sizeof(cplx) * 12 * V) : out;
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t};
                    long s = site_of(L, x, y, z, t);
                    cplx acc[4][3];
                    if (((x + y + z + t) & 1) == out_parity)
                        wilson_hop_site(acc, U, in, L, V, c, r, bc, G);
                    else
                        memset(acc, 0, sizeof(acc));
                    for (int sp = 0; sp < 4; sp++)
                        for (int a = 0; a < 3; a++) res[PIDX(V, s, a, sp)] = acc[sp][a];
                }
    if (dagger) {
        apply_gamma5(out, res, V);
        free(res);
        free(t1);
    }
}

/* ------------------------------------------------------------------ staggered */
void orc_staggered_D(double* outd, const double* Ud, const double* ind, const int L[4], double mass,
                     const int bc[4], int dagger) {
    long V = vol(L);
    cplx* out = (cplx*)outd;
    const cplx* U = (const cplx*)Ud;
    const cplx* in = (const cplx*)ind;
    double hs = dagger ? -0.5 : 0.5; /* D_hop^dagger = -D_hop */
#ifdef _OPENMP
#pragma omp parallel for collapse(3) num_threads(g_threads) schedule(static)
#endif
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t}, w;
                    long s = site_of(L, x, y, z, t);
                    cplx acc[3] = {0, 0, 0};
                    for (int nu = 0; nu < 4; nu++) {
                        /* eta_1 = 1, eta_nu = (-1)^(x_1+...+x_{nu-1}), 0-based global coordinates */
                        int e = 0;
                        for (int k = 0; k < nu; k++) e += c[k];
                        double eta = (e & 1) ? -1.0 : 1.0;
                        long np = neigh(L, c, nu, 1, &w);
                        double sp = w ? (double)bc[nu] : 1.0;
                        long nm = neigh(L, c, nu, -1, &w);
                        double sm = w ? (double)bc[nu] : 1.0;
                        for (int a = 0; a < 3; a++) {
                            cplx f = 0, b_ = 0;
                            for (int b = 0; b < 3; b++) {
                                f += U[UIDX(V, nu, s, a, b)] * in[b + 3 * np];
                                b_ += conj(U[UIDX(V, nu, nm, b, a)]) * in[b + 3 * nm];
                            }
                            acc[a] += eta * (sp * f - sm * b_);
                        }
                    }
                    for (int a = 0; a < 3; a++) out[a + 3 * s] = mass * in[a + 3 * s] + hs * acc[a];
                }
}

/* ------------------------------------------------------------------ BLAS-1 */
void orc_dot(const double* ad, const double* bd, long n, double* re, double* im) {
    const cplx* a = (const cplx*)ad;
    const cplx* b = (const cplx*)bd;
    cplx s = 0;
    for (long i = 0; i < n; i++) s += conj(a[i]) * b[i];
    *re = creal(s);
    *im = cimag(s);
}
void orc_axpy(double ar, double ai, const double* xd, double* yd, long n) {
    const cplx* x = (const cplx*)xd;
    cplx* y = (cplx*)yd;
    cplx a = ar + I * ai;
    for (long i = 0; i < n; i++) y[i] += a * x[i];
}
static double norm2(const cplx* a, long n) {
    double s = 0;
    for (long i = 0; i < n; i++) s += creal(a[i]) * creal(a[i]) + cimag(a[i]) * cimag(a[i]);
    return s;
}
static cplx cdot(const cplx* a, const cplx* b, long n) {
    cplx s = 0;
    for (long i = 0; i < n; i++) s += conj(a[i]) * b[i];
    return s;
}

/* ---- the same loops on several threads (the "all host cores" leg of bench.py's cpu_baseline; one thread = the plain loops above, bit for bit).
 * A vector is nblk blocks (the spin components of the reference layout) of m = n / nblk numbers; thread k owns the SAME index range of every block -- the
 * sites the stencil's static (t,z,y) partition gives it, so that the pages a thread touches first (NUMA) are the pages it works on in every loop.
 * Sums: one partial per thread over its ranges, added in thread order -- deterministic for a given thread count. */
static void par_range(long m, int nth, int k, long* j0, long* j1) { *j0 = m * k / nth; *j1 = m * (k + 1) / nth; }
static int vec_blocks(long n, const int L[4]) { return (n == 12 * vol(L)) ? 4 : 1; }
static double par_norm2(const cplx* a, long n, int nblk) {
    if (g_threads <= 1) return norm2(a, n);
    double part[1024];
    const int nth = g_threads > 1024 ? 1024 : g_threads;
    const long m = n / nblk;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
    {
#ifdef _OPENMP
        const int k = omp_get_thread_num();
#else
        const int k = 0;
#endif
        long j0, j1;
        par_range(m, nth, k, &j0, &j1);
        double sum = 0;
        for (int bk = 0; bk < nblk; bk++)
            for (long j = j0; j < j1; j++) { const cplx v = a[bk * m + j]; sum += creal(v) * creal(v) + cimag(v) * cimag(v); }
        part[k] = sum;
    }
    double tot = 0;
    for (int k = 0; k < nth; k++) tot += part[k];
    return tot;
}
static cplx par_cdot(const cplx* a, const cplx* b, long n, int nblk) {
    if (g_threads <= 1) return cdot(a, b, n);
    cplx part[1024];
    const int nth = g_threads > 1024 ? 1024 : g_threads;
    const long m = n / nblk;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
    {
#ifdef _OPENMP
        const int k = omp_get_thread_num();
#else
        const int k = 0;
#endif
        long j0, j1;
        par_range(m, nth, k, &j0, &j1);
        cplx sum = 0;
        for (int bk = 0; bk < nblk; bk++)
            for (long j = j0; j < j1; j++) sum += conj(a[bk * m + j]) * b[bk * m + j];
        part[k] = sum;
    }
    cplx tot = 0;
    for (int k = 0; k < nth; k++) tot += part[k];
    return tot;
}
/* scale_y 0: y += xa x;  1: y = ya y + xa x;  2: y = xa x (y is not read: first touch of a fresh vector) */
static void par_axpby(cplx xa, const cplx* x, double ya, int scale_y, cplx* y, long n, int nblk) {
    const int nth = g_threads > 1024 ? 1024 : (g_threads < 1 ? 1 : g_threads);
    const long m = n / nblk;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
    {
#ifdef _OPENMP
        const int k = omp_get_thread_num();
#else
        const int k = 0;
#endif
        long j0, j1;
        par_range(m, nth, k, &j0, &j1);
        for (int bk = 0; bk < nblk; bk++)
            for (long j = j0; j < j1; j++) {
                const long i = bk * m + j;
                if (scale_y == 2) y[i] = xa * x[i]; else if (scale_y) y[i] = ya * y[i] + xa * x[i]; else y[i] += xa * x[i];
            }
    }
}
/* first touch by the owning thread: dst <- src (nblk blocks of n / nblk numbers), for the arrays the caller hands to the all-cores leg */
void orc_numa_copy(double* dstd, const double* srcd, long n, int nblk) {
    cplx* dst = (cplx*)dstd;
    const cplx* src = (const cplx*)srcd;
    const int nth = g_threads > 1024 ? 1024 : (g_threads < 1 ? 1 : g_threads);
    const long m = n / nblk;
#ifdef _OPENMP
#pragma omp parallel num_threads(nth)
#endif
    {
#ifdef _OPENMP
        const int k = omp_get_thread_num();
#else
        const int k = 0;
#endif
        long j0, j1;
        par_range(m, nth, k, &j0, &j1);
        for (int bk = 0; bk < nblk; bk++) memcpy(dst + bk * m + j0, src + bk * m + j0, sizeof(cplx) * (size_t)(j1 - j0));
    }
}

/* ------------------------------------------------------------------ operator dispatch */
typedef struct {
    int kind;
    const cplx* U;
    const int* L;
    double km, r;
    const int* bc;
    long n; /* complex numbers per vector */
} op_t;

static void op_D(const op_t* o, cplx* out, const cplx* in, int dagger) {
    if (o->kind == ORC_WILSON)
        orc_wilson_D((double*)out, (const double*)o->U, (const double*)in, o->L, o->km, o->r, o->bc, dagger);
    else
        orc_staggered_D((double*)out, (const double*)o->U, (const double*)in, o->L, o->km, o->bc, dagger);
}
static op_t mk_op(int kind, const double* U, const int L[4], double km, double r, const int bc[4]) {
    op_t o = {kind, (const cplx*)U, L, km, r, bc, (kind == ORC_WILSON ? 12 : 3) * vol(L)};
    return o;
}

/* CG on D^dagger D: the loop of SURVEY.md 3.3
 *   mul!(q,A,p); c1=p.q; alpha=rho/c1; x+=alpha p; r-=alpha q; rho'=r.r; beta=rho'/rho; p=beta p+r */
static int cg_core(const op_t* o, cplx* x, const cplx* b, double eps, int maxiter, int fixed, int* iters,
                   double* final_rr) {
    long n = o->n;
    cplx* res = (cplx*)malloc(sizeof(cplx) * n);
    cplx* p = (cplx*)malloc(sizeof(cplx) * n);
    cplx* q = (cplx*)malloc(sizeof(cplx) * n);
    cplx* tmp = (cplx*)malloc(sizeof(cplx) * n);
    int status = 1, it = 0;
    const int nblk = vec_blocks(n, o->L), par = g_threads > 1;      /* several threads: the same loops, each thread on its own sites (par_* above) */
    op_D(o, tmp, x, 0);
    op_D(o, q, tmp, 1);
    if (par) { par_axpby(1.0, b, 0.0, 2, res, n, nblk); par_axpby(-1.0, q, 1.0, 0, res, n, nblk); par_axpby(1.0, res, 0.0, 2, p, n, nblk); }
    else { for (long i = 0; i < n; i++) res[i] = b[i] - q[i]; memcpy(p, res, sizeof(cplx) * n); }
    double rnorm = par ? par_norm2(res, n, nblk) : norm2(res, n);
    if (!fixed && rnorm < eps) { status = 0; goto done; }
    for (it = 1; it <= maxiter; it++) {
        op_D(o, tmp, p, 0);
        op_D(o, q, tmp, 1);
        cplx c1 = par ? par_cdot(p, q, n, nblk) : cdot(p, q, n);
        cplx alpha = rnorm / c1;
        if (par) { par_axpby(alpha, p, 1.0, 0, x, n, nblk); par_axpby(-alpha, q, 1.0, 0, res, n, nblk); }
        else { for (long i = 0; i < n; i++) x[i] += alpha * p[i]; for (long i = 0; i < n; i++) res[i] -= alpha * q[i]; }
        double c3 = par ? par_norm2(res, n, nblk) : norm2(res, n);
        if (!fixed && c3 < eps) { rnorm = c3; status = 0; break; }
        double beta = c3 / rnorm;
        if (par) par_axpby(1.0, res, beta, 1, p, n, nblk);
        else for (long i = 0; i < n; i++) p[i] = beta * p[i] + res[i];
        rnorm = c3;
    }
    if (it > maxiter) it = maxiter;
done:
    if (iters) *iters = it;
    if (final_rr) *final_rr = rnorm;
    free(res); free(p); free(q); free(tmp);
    return status;
}

int orc_cg_DdagD(int kind, double* x, const double* U, const double* b, const int L[4], double km, double r,
                 const int bc[4], double eps, int maxiter, int* iters, double* final_rr) {
    op_t o = mk_op(kind, U, L, km, r, bc);
    return cg_core(&o, (cplx*)x, (const cplx*)b, eps, maxiter, 0, iters, final_rr);
}
void orc_cg_DdagD_fixed(int kind, double* x, const double* U, const double* b, const int L[4], double km,
                        double r, const int bc[4], int niter) {
    op_t o = mk_op(kind, U, L, km, r, bc);
    cg_core(&o, (cplx*)x, (const cplx*)b, 0.0, niter, 1, NULL, NULL);
}

/* BiCGStab (van der Vorst) for A x = b with a generic apply callback */
typedef void (*apply_fn)(void* ctx, cplx* out, const cplx* in);

static int bicgstab_core(apply_fn A, void* ctx, long n, cplx* x, const cplx* b, double eps, int maxiter,
                         int* iters, double* final_rr) {
    cplx* r = (cplx*)malloc(sizeof(cplx) * n);
    cplx* r0 = (cplx*)malloc(sizeof(cplx) * n);
    cplx* p = (cplx*)malloc(sizeof(cplx) * n);
    cplx* v = (cplx*)malloc(sizeof(cplx) * n);
    cplx* s = (cplx*)malloc(sizeof(cplx) * n);
    cplx* t = (cplx*)malloc(sizeof(cplx) * n);
    int status = 1, it = 0;
    A(ctx, v, x);
    for (long i = 0; i < n; i++) r[i] = b[i] - v[i];
    memcpy(r0, r, sizeof(cplx) * n);
    memcpy(p, r, sizeof(cplx) * n);
    double rr = norm2(r, n);
    cplx rho = cdot(r0, r, n);
    if (rr < eps) { status = 0; goto done; }
    for (it = 1; it <= maxiter; it++) {
        A(ctx, v, p);
        cplx alpha = rho / cdot(r0, v, n);
        for (long i = 0; i < n; i++) s[i] = r[i] - alpha * v[i];
        double ss = norm2(s, n);
        if (ss < eps) {
            for (long i = 0; i < n; i++) x[i] += alpha * p[i];
            rr = ss; status = 0; break;
        }
        A(ctx, t, s);
        cplx omega = cdot(t, s, n) / norm2(t, n);
        for (long i = 0; i < n; i++) x[i] += alpha * p[i] + omega * s[i];
        for (long i = 0; i < n; i++) r[i] = s[i] - omega * t[i];
        rr = norm2(r, n);
        if (rr < eps) { status = 0; break; }
        cplx rho1 = cdot(r0, r, n);
        cplx beta = (rho1 / rho) * (alpha / omega);
        for (long i = 0; i < n; i++) p[i] = r[i] + beta * (p[i] - omega * v[i]);
        rho = rho1;
    }
    if (it > maxiter) it = maxiter;
done:
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    free(r); free(r0); free(p); free(v); free(s); free(t);
    return status;
}

typedef struct { const op_t* o; int dagger; } full_ctx;
static void apply_full(void* c, cplx* out, const cplx* in) {
    full_ctx* f = (full_ctx*)c;
    op_D(f->o, out, in, f->dagger);
}

int orc_bicgstab(int kind, double* x, const double* U, const double* b, const int L[4], double km, double r,
                 const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr) {
    op_t o = mk_op(kind, U, L, km, r, bc);
    full_ctx f = {&o, dagger};
    return bicgstab_core(apply_full, &f, o.n, (cplx*)x, (const cplx*)b, eps, maxiter, iters, final_rr);
}

/* BiCG (Fletcher) for A x = b -- `bicg`, the default method_CG of solve_DinvX!(y, D, x) (SURVEY.md 3.3, Appendix A "Solvers"):
 * two coupled recurrences with A and A^+, shadow residual r~_0 = r_0, stop on real(r.r) < eps. */
int orc_bicg(int kind, double* xd, const double* U, const double* bd, const int L[4], double km, double r_w, const int bc[4], int dagger,
             double eps, int maxiter, int* iters, double* final_rr) {
    op_t o = mk_op(kind, U, L, km, r_w, bc);
    const long n = o.n;
    cplx* x = (cplx*)xd;
    const cplx* b = (const cplx*)bd;
    cplx *r = malloc(sizeof(cplx) * n), *rt = malloc(sizeof(cplx) * n), *p = malloc(sizeof(cplx) * n), *pt = malloc(sizeof(cplx) * n),
         *q = malloc(sizeof(cplx) * n), *qt = malloc(sizeof(cplx) * n);
    int status = 1, it = 0;
    op_D(&o, q, x, dagger);
    for (long i = 0; i < n; i++) { r[i] = b[i] - q[i]; rt[i] = r[i]; p[i] = r[i]; pt[i] = r[i]; }
    double rr = norm2(r, n);
    cplx rho = cdot(rt, r, n);
    if (rr < eps) { status = 0; goto done; }
    for (it = 1; it <= maxiter; it++) {
        op_D(&o, q, p, dagger);
        op_D(&o, qt, pt, !dagger);
        cplx alpha = rho / cdot(pt, q, n);
        for (long i = 0; i < n; i++) { x[i] += alpha * p[i]; r[i] -= alpha * q[i]; rt[i] -= conj(alpha) * qt[i]; }
        rr = norm2(r, n);
        if (rr < eps) { status = 0; break; }
        cplx rho1 = cdot(rt, r, n);
        cplx beta = rho1 / rho;
        for (long i = 0; i < n; i++) { p[i] = r[i] + beta * p[i]; pt[i] = rt[i] + conj(beta) * pt[i]; }
        rho = rho1;
    }
    if (it > maxiter) it = maxiter;
done:
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    free(r); free(rt); free(p); free(pt); free(q); free(qt);
    return status;
}

/* even-odd preconditioning, Wilson:  D = [[1, -k H_eo], [-k H_oe, 1]]
 *   (1 - k^2 H_eo H_oe) x_e = b_e + k H_eo b_o ;   x_o = b_o + k H_oe x_e
 * Vectors stay full-lattice sized here; the "even" system lives on even sites with odd sites zero. */
typedef struct { const double* U; const int* L; double kappa, r; const int* bc; int dagger; long V; cplx *w1, *w2; } eo_ctx;
static void apply_schur(void* c, cplx* out, const cplx* in) {
    eo_ctx* e = (eo_ctx*)c;
    long n = 12 * e->V;
    orc_wilson_hop_parity((double*)e->w1, e->U, (const double*)in, e->L, e->r, e->bc, e->dagger, 1); /* H_oe in */
    orc_wilson_hop_parity((double*)e->w2, e->U, (const double*)e->w1, e->L, e->r, e->bc, e->dagger, 0); /* H_eo */
    double k2 = e->kappa * e->kappa;
    for (long i = 0; i < n; i++) out[i] = in[i] - k2 * e->w2[i];
}
static void mask_parity(cplx* v, const int L[4], int keep_parity) {
    long V = vol(L);
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++)
                    if (((x + y + z + t) & 1) != keep_parity) {
                        long s = site_of(L, x, y, z, t);
                        for (int sp = 0; sp < 4; sp++)
                            for (int a = 0; a < 3; a++) v[PIDX(V, s, a, sp)] = 0;
                    }
}

int orc_wilson_bicgstab_eo(double* xd, const double* U, const double* bd, const int L[4], double kappa, double r,
                           const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr) {
    long V = vol(L), n = 12 * V;
    cplx* x = (cplx*)xd;
    const cplx* b = (const cplx*)bd;
    cplx* be = (cplx*)malloc(sizeof(cplx) * n);
    cplx* bo = (cplx*)malloc(sizeof(cplx) * n);
    cplx* xe = (cplx*)calloc(n, sizeof(cplx));
    cplx* w = (cplx*)malloc(sizeof(cplx) * n);
    eo_ctx e = {U, L, kappa, r, bc, dagger, V, (cplx*)malloc(sizeof(cplx) * n), (cplx*)malloc(sizeof(cplx) * n)};
    memcpy(be, b, sizeof(cplx) * n); mask_parity(be, L, 0);
    memcpy(bo, b, sizeof(cplx) * n); mask_parity(bo, L, 1);
    /* rhs_e = b_e + k H_eo b_o */
    orc_wilson_hop_parity((double*)w, U, (const double*)bo, L, r, bc, dagger, 0);
    for (long i = 0; i < n; i++) be[i] += kappa * w[i];
    /* initial guess: even part of x */
    memcpy(xe, x, sizeof(cplx) * n); mask_parity(xe, L, 0);
    int st = bicgstab_core(apply_schur, &e, n, xe, be, eps, maxiter, iters, final_rr);
    /* x_o = b_o + k H_oe x_e */
    orc_wilson_hop_parity((double*)w, U, (const double*)xe, L, r, bc, dagger, 1);
    for (long i = 0; i < n; i++) x[i] = xe[i] + bo[i] + kappa * w[i];
    free(be); free(bo); free(xe); free(w); free(e.w1); free(e.w2);
    return st;
}

/* ------------------------------------------------------------------ multi-shift CG  (RHMC)
 * Solves (D^+D + sigma_j) x_j = b for all shifts at the cost of one Krylov space (Jegerlehner's shifted CG, the algorithm
 * of LatticeDiracOperators' `shiftedcg`, [EXT-RECALL]): CG on the unshifted system, shifted iterates by the zeta recurrences.
 * x0 = unshifted solution.  Zero initial guesses.  Stop when rr * max_j zeta_j^2 < eps.  Return 0 = converged. */
int orc_multishift_cg(int kind, double* x0d, double* xsd, const double* U, const double* bd, const int L[4], double km, double r,
                      const int bc[4], const double* sigma, int ns, double eps, int maxiter, int* iters, double* final_rr) {
    op_t o = mk_op(kind, U, L, km, r, bc);
    long n = o.n;
    const cplx* b = (const cplx*)bd;
    cplx* x0 = (cplx*)x0d;
    cplx* xs = (cplx*)xsd;
    cplx* res = (cplx*)malloc(sizeof(cplx) * n);
    cplx* p = (cplx*)malloc(sizeof(cplx) * n);
    cplx* q = (cplx*)malloc(sizeof(cplx) * n);
    cplx* tmp = (cplx*)malloc(sizeof(cplx) * n);
    cplx* ps = (cplx*)malloc(sizeof(cplx) * n * (ns > 0 ? ns : 1));
    double* zm = (double*)malloc(sizeof(double) * (ns + 1));
    double* z0 = (double*)malloc(sizeof(double) * (ns + 1));
    double* zp = (double*)malloc(sizeof(double) * (ns + 1));
    memset(x0, 0, sizeof(cplx) * n);
    memset(xs, 0, sizeof(cplx) * n * ns);
    memcpy(res, b, sizeof(cplx) * n);
    memcpy(p, b, sizeof(cplx) * n);
    for (int j = 0; j < ns; j++) { memcpy(ps + j * n, b, sizeof(cplx) * n); zm[j] = 1.0; z0[j] = 1.0; }
    double alpha_m = 1.0, beta_m = 0.0, rr = norm2(res, n), resid = rr;
    int status = 1, it = 0;
    if (rr < eps) { status = 0; goto done; }
    for (it = 1; it <= maxiter; it++) {
        op_D(&o, tmp, p, 0);
        op_D(&o, q, tmp, 1);
        double pAp = creal(cdot(p, q, n));
        double alpha = rr / pAp;
        for (long i = 0; i < n; i++) x0[i] += alpha * p[i];
        for (long i = 0; i < n; i++) res[i] -= alpha * q[i];
        double rrn = norm2(res, n);
        double beta = rrn / rr;
        for (long i = 0; i < n; i++) p[i] = beta * p[i] + res[i];
        double zmax = 0.0;
        for (int j = 0; j < ns; j++) {
            if (fabs(z0[j]) < 1e-100) { zp[j] = z0[j]; continue; }   /* converged long ago; frozen before zeta underflows to 0/0 */
            double den = zm[j] * alpha_m * (1.0 + alpha * sigma[j]) + alpha * beta_m * (zm[j] - z0[j]);
            zp[j] = z0[j] * zm[j] * alpha_m / den;
            double aj = (zp[j] / z0[j]) * alpha;
            double bj = (zp[j] / z0[j]) * (zp[j] / z0[j]) * beta;
            cplx* xj = xs + j * n;
            cplx* pj = ps + j * n;
            for (long i = 0; i < n; i++) xj[i] += aj * pj[i];
            for (long i = 0; i < n; i++) pj[i] = bj * pj[i] + zp[j] * res[i];
            if (fabs(zp[j]) > zmax) zmax = fabs(zp[j]);
            if (zp[j] * zp[j] * rrn < eps) zp[j] = 0.0;   /* residual of this shift below the target: frozen from the next iteration on */
        }
        for (int j = 0; j < ns; j++) { zm[j] = z0[j]; z0[j] = zp[j]; }
        alpha_m = alpha; beta_m = beta; rr = rrn;
        if (ns == 0) zmax = 1.0;
        resid = rr * (zmax > 1.0 ? zmax * zmax : 1.0);
        if (resid < eps) { status = 0; break; }
    }
    if (it > maxiter) it = maxiter;
done:
    if (iters) *iters = it;
    if (final_rr) *final_rr = resid;
    free(res); free(p); free(q); free(tmp); free(ps); free(zm); free(z0); free(zp);
    return status;
}

/* ------------------------------------------------------------------ pseudofermion action and its force
 * S_f = eta^+ (D^+D)^-1 eta  (reference callers: evaluate_FermiAction, src/updates/standardHMC.jl:71;
 * calc_UdSfdU!, src/md/AbstractMD.jl:129).  With X = (D^+D)^-1 eta and Y = D X,
 *   delta S_f = -2 Re( Y^+ (delta D) X ).
 * The force field G_mu(n) ("U dS_f/dU", a general 3x3 matrix per link) is DEFINED by
 *   d/d eps S_f[ U_mu(n) -> exp(i eps T) U_mu(n) ] at eps = 0   =   -2 Im tr( T G_mu(n) )     for every Hermitian T,
 * which tests/test_oracle_identities.py checks by central differences of S_f itself (convention independent).
 *   Wilson:    G = kappa s [ sum_spin (U X(n+mu))_s ((r - g_mu) Y(n))_s^+  -  sum_spin X(n)_s (U (r + g_mu) Y(n+mu))_s^+ ]
 *   staggered: G = -1/2 eta_mu(n) s [ (U X(n+mu)) Y(n)^+ + X(n) (U Y(n+mu))^+ ]
 * s = boundary sign when n -> n+mu crosses the global boundary.  Output in the gauge layout (UIDX). */
double orc_fermi_action(int kind, double* Xd, double* Yd, const double* U, const double* eta, const int L[4], double km, double r,
                        const int bc[4], double eps, int maxiter, int* iters, int* status) {
    op_t o = mk_op(kind, U, L, km, r, bc);
    cplx* X = (cplx*)Xd;
    memset(X, 0, sizeof(cplx) * o.n);
    int st = cg_core(&o, X, (const cplx*)eta, eps, maxiter, 0, iters, NULL);
    if (status) *status = st;
    if (Yd) op_D(&o, (cplx*)Yd, X, 0);
    return creal(cdot((const cplx*)eta, X, o.n));
}

void orc_wilson_force(double* Gd, const double* Ud, const double* Xd, const double* Yd, const int L[4], double kappa, double r,
                      const int bc[4]) {
    const cplx *U = (const cplx*)Ud, *X = (const cplx*)Xd, *Y = (const cplx*)Yd;
    cplx* G = (cplx*)Gd;
    long V = vol(L);
    cplx Gm[4][4][4];
    for (int nu = 0; nu < 4; nu++) gamma_mat(nu, Gm[nu]);
#pragma omp parallel for collapse(2) num_threads(g_threads)
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t}, w;
                    long s = site_of(L, x, y, z, t);
                    for (int mu = 0; mu < 4; mu++) {
                        long np = neigh(L, c, mu, 1, &w);
                        double sg = w ? (double)bc[mu] : 1.0;
                        cplx W[4][3], Z[4][3], Q[4][3], UQ[4][3];
                        for (int sp = 0; sp < 4; sp++)
                            for (int a = 0; a < 3; a++) {
                                cplx tw = 0;
                                for (int b = 0; b < 3; b++) tw += U[UIDX(V, mu, s, a, b)] * X[PIDX(V, np, b, sp)];
                                W[sp][a] = tw;
                                cplx tz = r * Y[PIDX(V, s, a, sp)], tq = r * Y[PIDX(V, np, a, sp)];
                                for (int s2 = 0; s2 < 4; s2++) {
                                    tz -= Gm[mu][sp][s2] * Y[PIDX(V, s, a, s2)];
                                    tq += Gm[mu][sp][s2] * Y[PIDX(V, np, a, s2)];
                                }
                                Z[sp][a] = tz;
                                Q[sp][a] = tq;
                            }
                        for (int sp = 0; sp < 4; sp++)
                            for (int a = 0; a < 3; a++) {
                                cplx tq = 0;
                                for (int b = 0; b < 3; b++) tq += U[UIDX(V, mu, s, a, b)] * Q[sp][b];
                                UQ[sp][a] = tq;
                            }
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) {
                                cplx m = 0;
                                for (int sp = 0; sp < 4; sp++)
                                    m += W[sp][a] * conj(Z[sp][b]) - X[PIDX(V, s, a, sp)] * conj(UQ[sp][b]);
                                G[UIDX(V, mu, s, a, b)] = kappa * sg * m;
                            }
                    }
                }
}

void orc_staggered_force(double* Gd, const double* Ud, const double* Xd, const double* Yd, const int L[4], const int bc[4]) {
    const cplx *U = (const cplx*)Ud, *X = (const cplx*)Xd, *Y = (const cplx*)Yd;
    cplx* G = (cplx*)Gd;
    long V = vol(L);
#pragma omp parallel for collapse(2) num_threads(g_threads)
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t}, w;
                    long s = site_of(L, x, y, z, t);
                    for (int mu = 0; mu < 4; mu++) {
                        int e = 0;
                        for (int k = 0; k < mu; k++) e += c[k];
                        double eta = (e & 1) ? -1.0 : 1.0;
                        long np = neigh(L, c, mu, 1, &w);
                        double sg = w ? (double)bc[mu] : 1.0;
                        cplx UX[3], UY[3];
                        for (int a = 0; a < 3; a++) {
                            cplx tx = 0, ty = 0;
                            for (int b = 0; b < 3; b++) {
                                tx += U[UIDX(V, mu, s, a, b)] * X[b + 3 * np];
                                ty += U[UIDX(V, mu, s, a, b)] * Y[b + 3 * np];
                            }
                            UX[a] = tx;
                            UY[a] = ty;
                        }
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++)
                                G[UIDX(V, mu, s, a, b)] = -0.5 * eta * sg * (UX[a] * conj(Y[b + 3 * s]) + X[a + 3 * s] * conj(UY[b]));
                    }
                }
}

/* ------------------------------------------------------------------ gauge side of the MD step
 * Reference callers: P_update!/U_update! src/md/AbstractMD.jl:78-118 (calc_dSdUmu!, Traceless_antihermitian_add!, exptU!),
 * action bookkeeping S = p.p/2 - S_g/NC + S_f  src/updates/standardHMC.jl:49-56.  The packages that own those generics are
 * not vendored, so the conventions below are this build's own, fixed by requiring dH/dtau = 0:
 *   momenta P_mu(n): traceless anti-Hermitian 3x3 (link layout), kinetic term K = -sum tr P^2  (= p.p/2 for P = i p_a T_a);
 *   links move as dU/dtau = P U  (U <- exp(dt P) U);
 *   gauge action S_g = -(beta/3) sum_plaq Re tr U_p;
 *   every force field G ("U dS/dU") is defined by dS/d eps [U -> exp(i eps T) U] = -2 Im tr(T G), so dS/dtau = 2 Re tr(P G)
 *   and Hamilton's equation reads dP/dtau = TA(G), TA(G) = (G - G^+)/2 - tr(G - G^+)/6. */
static void staple_sum(cplx A[3][3], const cplx* U, const int L[4], long V, const int c[4], int mu) {
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) A[a][b] = 0;
    int w;
    for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        cplx U1[3][3], U2[3][3], U3[3][3], T1[3][3], T2[3][3];
        long s = site_of(L, c[0], c[1], c[2], c[3]);
        long spm = neigh(L, c, mu, 1, &w), spn = neigh(L, c, nu, 1, &w), smn = neigh(L, c, nu, -1, &w);
        /* upper: U_nu(n+mu) U_mu(n+nu)^+ U_nu(n)^+ */
        load_link(U1, U, V, nu, spm);
        load_link(U2, U, V, mu, spn);
        load_link(U3, U, V, nu, s);
        mmd(T1, U1, U2);
        mmd(T2, T1, U3);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) A[a][b] += T2[a][b];
        /* lower: U_nu(n+mu-nu)^+ U_mu(n-nu)^+ U_nu(n-nu) */
        int cm[4] = {c[0], c[1], c[2], c[3]};
        cm[nu] = (c[nu] - 1 + L[nu]) % L[nu];
        long spmn = neigh(L, cm, mu, 1, &w);
        load_link(U1, U, V, nu, spmn);
        load_link(U2, U, V, mu, smn);
        load_link(U3, U, V, nu, smn);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                cplx t = 0;
                for (int k = 0; k < 3; k++) t += conj(U1[k][a]) * conj(U2[b][k]);   /* (U1^+ U2^+)_{ab} */
                T1[a][b] = t;
            }
        mm(T2, T1, U3);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) A[a][b] += T2[a][b];
    }
}

double orc_gauge_action(const double* Ud, const int L[4], double beta) {
    return -beta * 6.0 * (double)vol(L) * orc_plaquette(Ud, L);
}

/* G_mu(n) = -(beta/6) U_mu(n) A_mu(n), A = sum of the six staples */
void orc_gauge_force(double* Gd, const double* Ud, const int L[4], double beta) {
    const cplx* U = (const cplx*)Ud;
    cplx* G = (cplx*)Gd;
    long V = vol(L);
#pragma omp parallel for collapse(2) num_threads(g_threads)       /* same arithmetic per link whatever the thread count */
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    int c[4] = {x, y, z, t};
                    long s = site_of(L, x, y, z, t);
                    for (int mu = 0; mu < 4; mu++) {
                        cplx A[3][3], Um[3][3], T[3][3];
                        staple_sum(A, U, L, V, c, mu);
                        load_link(Um, U, V, mu, s);
                        mm(T, Um, A);
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) G[UIDX(V, mu, s, a, b)] = -(beta / 6.0) * T[a][b];
                    }
                }
}

/* P += c * TA(G) on every link */
void orc_momentum_add_ta(double* Pd, double cf, const double* Gd, const int L[4]) {
    cplx* P = (cplx*)Pd;
    const cplx* G = (const cplx*)Gd;
    long V = vol(L);
    for (int mu = 0; mu < 4; mu++)
#pragma omp parallel for num_threads(g_threads)
        for (long s = 0; s < V; s++) {
            cplx M[3][3], tr = 0;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) M[a][b] = 0.5 * (G[UIDX(V, mu, s, a, b)] - conj(G[UIDX(V, mu, s, b, a)]));
            for (int a = 0; a < 3; a++) tr += M[a][a];
            for (int a = 0; a < 3; a++) M[a][a] -= tr / 3.0;
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) P[UIDX(V, mu, s, a, b)] += cf * M[a][b];
        }
}

/* K = - sum tr P^2 */
double orc_momentum_action(const double* Pd, const int L[4]) {
    const cplx* P = (const cplx*)Pd;
    long V = vol(L);
    double k = 0;
    for (int mu = 0; mu < 4; mu++)
        for (long s = 0; s < V; s++)
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) k -= creal(P[UIDX(V, mu, s, a, b)] * P[UIDX(V, mu, s, b, a)]);
    return k;
}

/* U <- exp(dt P) U : Taylor series in Horner form (30 terms: exact to rounding for |dt P| < 4) */
void orc_link_update(double* Ud, const double* Pd, double dt, const int L[4]) {
    cplx* U = (cplx*)Ud;
    const cplx* P = (const cplx*)Pd;
    long V = vol(L);
    for (int mu = 0; mu < 4; mu++)
#pragma omp parallel for num_threads(g_threads)
        for (long s = 0; s < V; s++) {
            cplx X[3][3], E[3][3], T[3][3], Um[3][3];
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) { X[a][b] = dt * P[UIDX(V, mu, s, a, b)]; E[a][b] = (a == b); }
            for (int k = 30; k >= 1; k--) {      /* E = 1 + X E / k */
                mm(T, X, E);
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) E[a][b] = (a == b ? 1.0 : 0.0) + T[a][b] / (double)k;
            }
            load_link(Um, U, V, mu, s);
            mm(T, E, Um);
            for (int a = 0; a < 3; a++)
                for (int b = 0; b < 3; b++) U[UIDX(V, mu, s, a, b)] = T[a][b];
        }
}

/* ------------------------------------------------------------------ clover (Sheikholeslami-Wohlert) term, SURVEY.md 8(f) rank 2
 * BASELINE.json configs[3] names a Wilson-clover operator; the reference itself rejects it (src/system/universe.jl:129-131,
 * test/runtests.jl:153-158 commented out), so there is NO reference behaviour to match: this is the textbook definition
 * (Luscher, Sint, Sommer, Weisz, hep-lat/9605038 eqs 2.5-2.7) in the hopping normalisation used above,
 *     D_sw = 1 - kappa H + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu},
 *     sigma_{mu nu} = (i/2)[g_mu, g_nu],   F_{mu nu}(x) = (Q_{mu nu}(x) - Q_{mu nu}(x)^+)/8,
 *     Q_{mu nu}(x) = sum of the four plaquette loops in the mu-nu plane that start and end at x (the "clover"),
 * and is checked through identities only (Hermiticity, gamma5-hermiticity, gauge covariance, A = 1 on a pure-gauge field).
 * clov holds the full 12x12 matrix A(x) = 1 + i kappa c_sw sum sigma F per site, index (s*3+c) row-major. */
static void link_dag(cplx D[3][3], cplx A[3][3]) {
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) D[a][b] = conj(A[b][a]);
}
static void clover_leaves(cplx Q[3][3], const cplx* U, const int L[4], long V, const int c[4], int mu, int nu) {
    int w;
    cplx A[3][3], B[3][3], C[3][3], D[3][3], Ad[3][3], Bd[3][3], Cd[3][3], Dd[3][3], T1[3][3], T2[3][3], T3[3][3];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) Q[a][b] = 0;
    long x = site_of(L, c[0], c[1], c[2], c[3]);
    int cm[4] = {c[0], c[1], c[2], c[3]}, cn[4] = {c[0], c[1], c[2], c[3]}, cmn[4];
    cm[mu] = (c[mu] - 1 + L[mu]) % L[mu];                 /* x - mu */
    cn[nu] = (c[nu] - 1 + L[nu]) % L[nu];                 /* x - nu */
    for (int k = 0; k < 4; k++) cmn[k] = cm[k];
    cmn[nu] = (c[nu] - 1 + L[nu]) % L[nu];                /* x - mu - nu */
    long xpm = neigh(L, c, mu, 1, &w), xpn = neigh(L, c, nu, 1, &w);
    long xmm = site_of(L, cm[0], cm[1], cm[2], cm[3]), xmn = site_of(L, cn[0], cn[1], cn[2], cn[3]);
    long xmmpn = neigh(L, cm, nu, 1, &w), xmnpm = neigh(L, cn, mu, 1, &w), xmmmn = site_of(L, cmn[0], cmn[1], cmn[2], cmn[3]);
#define ADDQ(M) for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Q[a][b] += M[a][b]
    /* 1: U_mu(x) U_nu(x+mu) U_mu^+(x+nu) U_nu^+(x) */
    load_link(A, U, V, mu, x); load_link(B, U, V, nu, xpm); load_link(C, U, V, mu, xpn); load_link(D, U, V, nu, x);
    mm(T1, A, B); mmd(T2, T1, C); mmd(T3, T2, D); ADDQ(T3);
    /* 2: U_nu(x) U_mu^+(x-mu+nu) U_nu^+(x-mu) U_mu(x-mu) */
    load_link(A, U, V, nu, x); load_link(B, U, V, mu, xmmpn); load_link(C, U, V, nu, xmm); load_link(D, U, V, mu, xmm);
    mmd(T1, A, B); mmd(T2, T1, C); mm(T3, T2, D); ADDQ(T3);
    /* 3: U_mu^+(x-mu) U_nu^+(x-mu-nu) U_mu(x-mu-nu) U_nu(x-nu) */
    load_link(A, U, V, mu, xmm); load_link(B, U, V, nu, xmmmn); load_link(C, U, V, mu, xmmmn); load_link(D, U, V, nu, xmn);
    link_dag(Ad, A); link_dag(Bd, B); mm(T1, Ad, Bd); mm(T2, T1, C); mm(T3, T2, D); ADDQ(T3);
    /* 4: U_nu^+(x-nu) U_mu(x-nu) U_nu(x+mu-nu) U_mu^+(x) */
    load_link(A, U, V, nu, xmn); load_link(B, U, V, mu, xmn); load_link(C, U, V, nu, xmnpm); load_link(D, U, V, mu, x);
    link_dag(Ad, A); mm(T1, Ad, B); mm(T2, T1, C); mmd(T3, T2, D); ADDQ(T3);
#undef ADDQ
    (void)Cd; (void)Dd; (void)Bd;
}

void orc_clover_build(double* clovd, const double* Ud, const int L[4], double kappa, double csw) {
    const cplx* U = (const cplx*)Ud;
    cplx* clov = (cplx*)clovd;
    long V = vol(L);
    cplx G[4][4][4];
    for (int nu = 0; nu < 4; nu++) gamma_mat(nu, G[nu]);
    /* sites are independent: threads only split the site loop (same arithmetic per site whatever the thread count) */
#pragma omp parallel for collapse(2) num_threads(g_threads) schedule(static)
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    cplx S[4][4];
                    int c[4] = {x, y, z, t};
                    long s = site_of(L, x, y, z, t);
                    cplx* A = clov + 144 * s;
                    for (int i = 0; i < 144; i++) A[i] = 0;
                    for (int i = 0; i < 12; i++) A[i * 12 + i] = 1.0;
                    for (int mu = 0; mu < 4; mu++)
                        for (int nu = mu + 1; nu < 4; nu++) {
                            cplx Q[3][3], F[3][3];
                            clover_leaves(Q, U, L, V, c, mu, nu);
                            for (int a = 0; a < 3; a++)
                                for (int b = 0; b < 3; b++) F[a][b] = (Q[a][b] - conj(Q[b][a])) / 8.0;
                            for (int a = 0; a < 4; a++)       /* sigma = (i/2)(g_mu g_nu - g_nu g_mu) */
                                for (int b = 0; b < 4; b++) {
                                    cplx t1 = 0;
                                    for (int k = 0; k < 4; k++) t1 += G[mu][a][k] * G[nu][k][b] - G[nu][a][k] * G[mu][k][b];
                                    S[a][b] = 0.5 * I * t1;
                                }
                            for (int sa = 0; sa < 4; sa++)
                                for (int sb = 0; sb < 4; sb++)
                                    for (int ca = 0; ca < 3; ca++)
                                        for (int cb = 0; cb < 3; cb++)
                                            A[(sa * 3 + ca) * 12 + (sb * 3 + cb)] += I * kappa * csw * S[sa][sb] * F[ca][cb];
                        }
                }
}

/* out = D_sw in = Wilson D in + (A - 1) in   (dagger: A is Hermitian and commutes with gamma5, so D_sw^+ = gamma5 D_sw gamma5) */
void orc_wilson_clover_D(double* outd, const double* Ud, const double* clovd, const double* ind, const int L[4], double kappa,
                         double r, const int bc[4], int dagger) {
    long V = vol(L);
    orc_wilson_D(outd, Ud, ind, L, kappa, r, bc, dagger);
    cplx* out = (cplx*)outd;
    const cplx* in = (const cplx*)ind;
    const cplx* clov = (const cplx*)clovd;
#pragma omp parallel for num_threads(g_threads) schedule(static)
    for (long s = 0; s < V; s++) {
        const cplx* A = clov + 144 * s;
        for (int i = 0; i < 12; i++) {
            cplx t = 0;
            for (int j = 0; j < 12; j++) t += (A[i * 12 + j] - (i == j ? 1.0 : 0.0)) * in[PIDX(V, s, j % 3, j / 3)];
            out[PIDX(V, s, i % 3, i / 3)] += t;
        }
    }
}

/* CG on D_sw^+ D_sw (same loop and stopping rule as orc_cg_DdagD) */
int orc_cg_clover(double* xd, const double* U, const double* clov, const double* bd, const int L[4], double kappa, double r,
                  const int bc[4], double eps, int maxiter, int* iters, double* final_rr) {
    long n = 12 * vol(L);
    cplx *x = (cplx*)xd, *res = malloc(sizeof(cplx) * n), *p = malloc(sizeof(cplx) * n), *q = malloc(sizeof(cplx) * n),
         *tmp = malloc(sizeof(cplx) * n);
    const cplx* b = (const cplx*)bd;
    int status = 1, it = 0;
    orc_wilson_clover_D((double*)tmp, U, clov, (const double*)x, L, kappa, r, bc, 0);
    orc_wilson_clover_D((double*)q, U, clov, (const double*)tmp, L, kappa, r, bc, 1);
    for (long i = 0; i < n; i++) res[i] = b[i] - q[i];
    memcpy(p, res, sizeof(cplx) * n);
    double rnorm = norm2(res, n);
    if (rnorm < eps) status = 0;
    for (it = 1; status && it <= maxiter; it++) {
        orc_wilson_clover_D((double*)tmp, U, clov, (const double*)p, L, kappa, r, bc, 0);
        orc_wilson_clover_D((double*)q, U, clov, (const double*)tmp, L, kappa, r, bc, 1);
        cplx alpha = rnorm / cdot(p, q, n);
        for (long i = 0; i < n; i++) { x[i] += alpha * p[i]; res[i] -= alpha * q[i]; }
        double c3 = norm2(res, n);
        if (c3 < eps) { rnorm = c3; status = 0; break; }
        double beta = c3 / rnorm;
        for (long i = 0; i < n; i++) p[i] = beta * p[i] + res[i];
        rnorm = c3;
    }
    if (it > maxiter) it = maxiter;
    if (iters) *iters = it;
    if (final_rr) *final_rr = rnorm;
    free(res); free(p); free(q); free(tmp);
    return status;
}

/* ------------------------------------------------------------------ even-odd preconditioning with the clover term
 * D_sw = A - k H with A block diagonal in parity (SURVEY.md 8(f) rank 2 "even-odd clover inverse"; textbook, no reference behaviour):
 *   (1 - k^2 A_ee^-1 H_eo A_oo^-1 H_oe) x_e = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o) ;   x_o = A_oo^-1 (b_o + k H_oe x_e)
 * inv[V][12][12] = A(x)^-1 by Gauss-Jordan elimination with partial pivoting. */
void orc_clover_invert(double* invd, const double* clovd, const int L[4]) {
    long V = vol(L);
    const cplx* clov = (const cplx*)clovd;
    cplx* inv = (cplx*)invd;
    for (long s = 0; s < V; s++) {
        cplx a[12][24];
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 12; j++) { a[i][j] = clov[144 * s + i * 12 + j]; a[i][12 + j] = (i == j) ? 1.0 : 0.0; }
        for (int k = 0; k < 12; k++) {
            int piv = k;
            for (int i = k + 1; i < 12; i++)
                if (cabs(a[i][k]) > cabs(a[piv][k])) piv = i;
            if (piv != k)
                for (int j = 0; j < 24; j++) { cplx t = a[k][j]; a[k][j] = a[piv][j]; a[piv][j] = t; }
            cplx d = 1.0 / a[k][k];
            for (int j = 0; j < 24; j++) a[k][j] *= d;
            for (int i = 0; i < 12; i++) {
                if (i == k) continue;
                cplx f = a[i][k];
                for (int j = 0; j < 24; j++) a[i][j] -= f * a[k][j];
            }
        }
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 12; j++) inv[144 * s + i * 12 + j] = a[i][12 + j];
    }
}
/* out(s) = M(s) in(s) on the sites of one parity, zero on the others */
static void site_mat_parity(cplx* out, const cplx* M, const cplx* in, const int L[4], int parity) {
    long V = vol(L);
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    long s = site_of(L, x, y, z, t);
                    int keep = ((x + y + z + t) & 1) == parity;
                    for (int i = 0; i < 12; i++) {
                        cplx acc = 0;
                        if (keep)
                            for (int j = 0; j < 12; j++) acc += M[144 * s + i * 12 + j] * in[PIDX(V, s, j % 3, j / 3)];
                        out[PIDX(V, s, i % 3, i / 3)] = acc;
                    }
                }
}
typedef struct { const double* U; const cplx* Ainv; const int* L; double kappa, r; const int* bc; int dagger; long V; cplx *w1, *w2; } eoc_ctx;
static void apply_schur_clover(void* c, cplx* out, const cplx* in) {
    eoc_ctx* e = (eoc_ctx*)c;
    long n = 12 * e->V;
    orc_wilson_hop_parity((double*)e->w1, e->U, (const double*)in, e->L, e->r, e->bc, e->dagger, 1);   /* H_oe in */
    site_mat_parity(e->w2, e->Ainv, e->w1, e->L, 1);                                                  /* A_oo^-1 */
    orc_wilson_hop_parity((double*)e->w1, e->U, (const double*)e->w2, e->L, e->r, e->bc, e->dagger, 0); /* H_eo */
    site_mat_parity(e->w2, e->Ainv, e->w1, e->L, 0);                                                  /* A_ee^-1 */
    double k2 = e->kappa * e->kappa;
    for (long i = 0; i < n; i++) out[i] = in[i] - k2 * e->w2[i];
}
int orc_wilson_clover_bicgstab_eo(double* xd, const double* U, const double* clov, const double* bd, const int L[4], double kappa, double r,
                                  const int bc[4], int dagger, double eps, int maxiter, int* iters, double* final_rr) {
    long V = vol(L), n = 12 * V;
    cplx* x = (cplx*)xd;
    const cplx* b = (const cplx*)bd;
    cplx* Ainv = (cplx*)malloc(sizeof(cplx) * 144 * V);
    orc_clover_invert((double*)Ainv, clov, L);
    cplx* be = (cplx*)malloc(sizeof(cplx) * n);
    cplx* bo = (cplx*)malloc(sizeof(cplx) * n);
    cplx* xe = (cplx*)calloc(n, sizeof(cplx));
    cplx* w = (cplx*)malloc(sizeof(cplx) * n);
    cplx* w2 = (cplx*)malloc(sizeof(cplx) * n);
    eoc_ctx e = {U, Ainv, L, kappa, r, bc, dagger, V, (cplx*)malloc(sizeof(cplx) * n), (cplx*)malloc(sizeof(cplx) * n)};
    memcpy(be, b, sizeof(cplx) * n); mask_parity(be, L, 0);
    memcpy(bo, b, sizeof(cplx) * n); mask_parity(bo, L, 1);
    /* rhs_e = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o) */
    site_mat_parity(w2, Ainv, bo, L, 1);
    orc_wilson_hop_parity((double*)w, U, (const double*)w2, L, r, bc, dagger, 0);
    for (long i = 0; i < n; i++) w[i] = be[i] + kappa * w[i];
    site_mat_parity(be, Ainv, w, L, 0);
    memcpy(xe, x, sizeof(cplx) * n); mask_parity(xe, L, 0);
    int st = bicgstab_core(apply_schur_clover, &e, n, xe, be, eps, maxiter, iters, final_rr);
    /* x_o = A_oo^-1 (b_o + k H_oe x_e) */
    orc_wilson_hop_parity((double*)w, U, (const double*)xe, L, r, bc, dagger, 1);
    for (long i = 0; i < n; i++) w[i] = bo[i] + kappa * w[i];
    site_mat_parity(w2, Ainv, w, L, 1);
    for (long i = 0; i < n; i++) x[i] = xe[i] + w2[i];
    free(Ainv); free(be); free(bo); free(xe); free(w); free(w2); free(e.w1); free(e.w2);
    return st;
}

/* ------------------------------------------------------------------ force of the clover term (2-flavour Wilson-clover HMC, BASELINE.json configs[3])
 * S_f = phi^+ (D_sw^+ D_sw)^-1 phi, X = (D_sw^+ D_sw)^-1 phi, Y = D_sw X:  dS_f = -2 Re(Y^+ dD_sw X), dD_sw = -kappa dH + dA.  The hopping
 * part is orc_wilson_force; this adds the part of dA = i kappa c_sw sum_{mu<nu} sigma_{mu nu} dF_{mu nu}:
 *   Y^+ sigma F X = tr_c(F M),  M^{mu nu}(x) = sum_{s s'} sigma_{s s'} X_{s'}(x) Y_s(x)^+,   Lambda = M + M^+  (Hermitian)
 *   dS_clover = (kappa c_sw / 4) sum_x sum_{mu<nu} Im tr( dQ_{mu nu}(x) Lambda^{mu nu}(x) )
 * Every leaf of Q is a closed 4-step path; for the link at step k (leaf = P1 L_k P2), with the convention of the other force fields
 * (dS/d eps[U -> exp(i eps T) U] = -2 Im tr(T G)):
 *   forward  step, L_k = U_rho(z):    G_rho(z) += -i (kappa c_sw / 8) U_rho(z) P2 Lambda P1
 *   backward step, L_k = U_rho(z)^+:  G_rho(z) += +i (kappa c_sw / 8) P2 Lambda P1 U_rho(z)^+
 * Written as a scatter over (site, plane, leaf, step) with generic path stepping -- deliberately not the gather the device uses. */
static const int LEAF_STEPS[4][4][2] = {   /* {0 = mu | 1 = nu, sign} */
    {{0, 1}, {1, 1}, {0, -1}, {1, -1}},
    {{1, 1}, {0, -1}, {1, -1}, {0, 1}},
    {{0, -1}, {1, -1}, {0, 1}, {1, 1}},
    {{1, -1}, {0, 1}, {1, 1}, {0, -1}}};
void orc_clover_force(double* Gd, const double* Ud, const double* Xd, const double* Yd, const int L[4], double kappa, double csw,
                      int accumulate) {
    const cplx* U = (const cplx*)Ud;
    const cplx* X = (const cplx*)Xd;
    const cplx* Y = (const cplx*)Yd;
    cplx* G = (cplx*)Gd;
    long V = vol(L);
    if (!accumulate) memset(G, 0, sizeof(cplx) * 36 * V);
    cplx Gm[4][4][4];
    for (int nu = 0; nu < 4; nu++) gamma_mat(nu, Gm[nu]);
    const double cf = kappa * csw / 8.0;
    cplx (*Lam)[6][3][3] = malloc(sizeof(cplx) * 54 * V);
    for (long s = 0; s < V; s++) {
        int plane = 0;
        for (int mu = 0; mu < 4; mu++)
            for (int nu = mu + 1; nu < 4; nu++, plane++) {
                cplx S[4][4], M[3][3];
                for (int a = 0; a < 4; a++)
                    for (int b = 0; b < 4; b++) {
                        cplx t1 = 0;
                        for (int k = 0; k < 4; k++) t1 += Gm[mu][a][k] * Gm[nu][k][b] - Gm[nu][a][k] * Gm[mu][k][b];
                        S[a][b] = 0.5 * I * t1;
                    }
                for (int b = 0; b < 3; b++)
                    for (int a = 0; a < 3; a++) {
                        cplx t = 0;
                        for (int sa = 0; sa < 4; sa++)
                            for (int sb = 0; sb < 4; sb++) t += S[sa][sb] * X[PIDX(V, s, b, sb)] * conj(Y[PIDX(V, s, a, sa)]);
                        M[b][a] = t;
                    }
                for (int a = 0; a < 3; a++)
                    for (int b = 0; b < 3; b++) Lam[s][plane][a][b] = M[a][b] + conj(M[b][a]);
            }
    }
    for (int t = 0; t < L[3]; t++)
        for (int z = 0; z < L[2]; z++)
            for (int y = 0; y < L[1]; y++)
                for (int x = 0; x < L[0]; x++) {
                    long s = site_of(L, x, y, z, t);
                    int plane = 0;
                    for (int mu = 0; mu < 4; mu++)
                        for (int nu = mu + 1; nu < 4; nu++, plane++)
                            for (int leaf = 0; leaf < 4; leaf++) {
                                cplx Lk[4][3][3];
                                long zs[4];
                                int rho[4], fwd[4];
                                int c[4] = {x, y, z, t};
                                for (int k = 0; k < 4; k++) {
                                    int d = LEAF_STEPS[leaf][k][0] ? nu : mu, sg = LEAF_STEPS[leaf][k][1], w;
                                    rho[k] = d;
                                    fwd[k] = sg > 0;
                                    long here = site_of(L, c[0], c[1], c[2], c[3]);
                                    long next = neigh(L, c, d, sg, &w);
                                    c[d] = (c[d] + sg + L[d]) % L[d];
                                    zs[k] = sg > 0 ? here : next;            /* the site the link U_d lives on */
                                    cplx Um[3][3];
                                    load_link(Um, U, V, d, zs[k]);
                                    if (sg > 0) memcpy(Lk[k], Um, sizeof(Um));
                                    else link_dag(Lk[k], Um);
                                }
                                for (int k = 0; k < 4; k++) {
                                    cplx P1[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, P2[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, T1[3][3], T2[3][3], W[3][3];
                                    for (int j = 0; j < k; j++) { mm(T1, P1, Lk[j]); memcpy(P1, T1, sizeof(T1)); }
                                    for (int j = k + 1; j < 4; j++) { mm(T1, P2, Lk[j]); memcpy(P2, T1, sizeof(T1)); }
                                    mm(T1, P2, Lam[s][plane]);
                                    mm(T2, T1, P1);                           /* P2 Lambda P1 */
                                    if (fwd[k]) mm(W, Lk[k], T2); else mm(W, T2, Lk[k]);
                                    const cplx f = fwd[k] ? -I * cf : I * cf;
                                    for (int a = 0; a < 3; a++)
                                        for (int b = 0; b < 3; b++) G[UIDX(V, rho[k], zs[k], a, b)] += f * W[a][b];
                                }
                            }
                }
    free(Lam);
}
