// md.hip -- gauge side of the molecular-dynamics step on the device: staple force, traceless
// anti-Hermitian momentum update, exponential link update, momentum sampling and kinetic term.  Reference callers:
// P_update! / U_update! /root/reference/src/md/AbstractMD.jl:78-118 (calc_dSdUmu!, Traceless_antihermitian_add!, exptU!),
// gauss_distribution!(p) src/md/standardMD.jl:86, action bookkeeping src/updates/standardHMC.jl:49-56.
// With lqcd_calc_UdSfdU (force.hip) a whole MD step runs without a host transfer: the links are uploaded once per trajectory.
//
// Conventions (the packages that own these generics are not vendored; these are fixed by dH/dtau = 0 and tested as such):
//   momenta P_mu(n): traceless anti-Hermitian 3x3 matrices in a gauge-shaped field, K = -sum tr P^2 (= p.p/2 for P = i p_a T_a);
//   dU/dtau = P U (U <- exp(dt P) U);   S_g = -(beta/3) sum_plaq Re tr U_p;
//   every force field G obeys dS/d eps [U -> exp(i eps T) U] = -2 Im tr(T G), hence dP/dtau = TA(G) = (G - G^+)/2 - tr(.)/3.
#include "lqcd_internal.h"

#include <cmath>

namespace lqcd {

typedef cd m3[9];

__device__ __forceinline__ void load_m3(cd (&u)[9], const double2* __restrict__ base, int stride) {
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = ld(base + (size_t)e * stride);
}
// C = A B
__device__ __forceinline__ void mm3(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma(t, A[a * 3 + k], B[k * 3 + b]);
            C[a * 3 + b] = t;
        }
}
// C = A B^+
__device__ __forceinline__ void mm3_nd(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma_conj(t, B[b * 3 + k], A[a * 3 + k]);
            C[a * 3 + b] = t;
        }
}
// C = A^+ B^+ = (B A)^+
__device__ __forceinline__ void mm3_dd(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma(t, B[b * 3 + k], A[k * 3 + a]);
            C[a * 3 + b] = mk(t.re, -t.im);
        }
}

// link U_mu at the site with local coordinates c (periodic wrap; links carry no boundary sign)
__device__ __forceinline__ const double2* link_at(const Geom& g, const double2* __restrict__ U, const int (&c)[4], int mu) {
    const int p = (c[0] + c[1] + c[2] + c[3]) & 1;
    return U + glink_off(g, p, mu, coords_to_cb(g, c));
}
__device__ __forceinline__ void shift(int (&d)[4], const Geom& g, int mu, int dir) {
    d[mu] += dir;
    if (d[mu] == g.L[mu]) d[mu] = 0;
    if (d[mu] < 0) d[mu] = g.L[mu] - 1;
}

// out_mu(n) = coef * U_mu(n) * sum_{nu != mu} [ U_nu(n+mu) U_mu(n+nu)^+ U_nu(n)^+  +  U_nu(n+mu-nu)^+ U_mu(n-nu)^+ U_nu(n-nu) ]
// workgroup = 64 sites of one parity x 4 waves (wave = mu); the 6 x 3 neighbour links are re-used across waves/sites through L2
// FUSE_TA = false: out = G.   FUSE_TA = true: out (the momenta) += factor * TA(G) -- P_update! in one pass, G never stored.
template <bool FUSE_TA>
__global__ __launch_bounds__(256) void gauge_force_kernel(Geom g, const double2* __restrict__ U, double2* __restrict__ out, double coef, double factor) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    const int Gs = glink_stride(g);
    int c[4];
    cb_to_coords(g, p, i, c);
    cd A[9];
#pragma unroll
    for (int e = 0; e < 9; e++) A[e] = mk(0.0, 0.0);
    int cpm[4] = {c[0], c[1], c[2], c[3]};
    shift(cpm, g, mu, 1);
    for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        cd u1[9], u2[9], u3[9], t1[9], t2[9];
        int cpn[4] = {c[0], c[1], c[2], c[3]}, cmn[4] = {c[0], c[1], c[2], c[3]}, cpmn[4] = {cpm[0], cpm[1], cpm[2], cpm[3]};
        shift(cpn, g, nu, 1);
        shift(cmn, g, nu, -1);
        shift(cpmn, g, nu, -1);
        load_m3(u1, link_at(g, U, cpm, nu), Gs);
        load_m3(u2, link_at(g, U, cpn, mu), Gs);
        load_m3(u3, link_at(g, U, c, nu), Gs);
        mm3_nd(t1, u1, u2);
        mm3_nd(t2, t1, u3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        load_m3(u1, link_at(g, U, cpmn, nu), Gs);
        load_m3(u2, link_at(g, U, cmn, mu), Gs);
        load_m3(u3, link_at(g, U, cmn, nu), Gs);
        mm3_dd(t1, u1, u2);
        mm3(t2, t1, u3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
    }
    cd um[9], r[9];
    load_m3(um, U + glink_off(g, p, mu, i), Gs);
    mm3(r, um, A);
    double2* o = out + glink_off(g, p, mu, i);
    if constexpr (!FUSE_TA) {
#pragma unroll
        for (int e = 0; e < 9; e++) st(o + (size_t)e * Gs, mk(coef * r[e].re, coef * r[e].im));
    } else {
        cd a[9];
        const double f = 0.5 * coef * factor;
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int y = 0; y < 3; y++) a[x * 3 + y] = mk(f * (r[x * 3 + y].re - r[y * 3 + x].re), f * (r[x * 3 + y].im + r[y * 3 + x].im));
        const double tr = (a[0].im + a[4].im + a[8].im) / 3.0;
        a[0].im -= tr; a[4].im -= tr; a[8].im -= tr;
#pragma unroll
        for (int e = 0; e < 9; e++) {
            const cd pv = ld(o + (size_t)e * Gs);
            st(o + (size_t)e * Gs, mk(pv.re + a[e].re, pv.im + a[e].im));
        }
    }
}

// one thread per link; workgroup = 64 consecutive sites of one parity x 4 waves (wave = mu): every access of a wave is one
// contiguous 1 KiB run of the chunk-blocked layout
__device__ __forceinline__ bool link_of_thread(const Geom& g, size_t& off) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return false;
    off = glink_off(g, p, mu, i);
    return true;
}

// P += c * TA(G)
__global__ __launch_bounds__(256) void momentum_add_ta_kernel(Geom g, double2* __restrict__ P, double cf, const double2* __restrict__ G) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd m[9], a[9];
    load_m3(m, G + off, Gs);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) a[r * 3 + q] = mk(0.5 * (m[r * 3 + q].re - m[q * 3 + r].re), 0.5 * (m[r * 3 + q].im + m[q * 3 + r].im));
    const double tr = (a[0].im + a[4].im + a[8].im) / 3.0;    // the anti-Hermitian part has an imaginary trace
    a[0].im -= tr; a[4].im -= tr; a[8].im -= tr;
#pragma unroll
    for (int e = 0; e < 9; e++) {
        cd pv = ld(P + off + (size_t)e * Gs);
        st(P + off + (size_t)e * Gs, mk(fma(cf, a[e].re, pv.re), fma(cf, a[e].im, pv.im)));
    }
}

// U <- exp(dt P) U, Taylor series in Horner form.  Terms: 12 when the max-abs-row-sum norm of dt P is below 0.2
// (0.2^13/13! = 1e-19), otherwise 24 (exact to rounding up to norm 2); an MD step has |dt P| of a few 1e-2.
__global__ __launch_bounds__(256) void link_exp_update_kernel(Geom g, double2* __restrict__ U, double dt, const double2* __restrict__ P) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd x[9], e[9], t[9], u[9];
    load_m3(x, P + off, Gs);
#pragma unroll
    for (int k = 0; k < 9; k++) { x[k] = mk(dt * x[k].re, dt * x[k].im); e[k] = mk((k % 4 == 0) ? 1.0 : 0.0, 0.0); }
    double nrm = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
        nrm = fmax(nrm, (fabs(x[a * 3].re) + fabs(x[a * 3].im)) + (fabs(x[a * 3 + 1].re) + fabs(x[a * 3 + 1].im)) + (fabs(x[a * 3 + 2].re) + fabs(x[a * 3 + 2].im)));
    for (int n = nrm < 0.2 ? 12 : 24; n >= 1; n--) {
        mm3(t, x, e);
        const double inv = 1.0 / (double)n;
#pragma unroll
        for (int k = 0; k < 9; k++) e[k] = mk(((k % 4 == 0) ? 1.0 : 0.0) + inv * t[k].re, inv * t[k].im);
    }
    load_m3(u, U + off, Gs);
    mm3(t, e, u);
#pragma unroll
    for (int k = 0; k < 9; k++) st(U + off + (size_t)k * Gs, t[k]);
}

__device__ inline void gauss2(uint64_t k, double& a, double& b) {
    const double u1 = u01(k), u2 = u01(splitmix64(k));
    const double r = sqrt(-2.0 * log(u1));
    double s, c;
    sincos(6.283185307179586 * u2, &s, &c);
    a = r * c; b = r * s;
}
// P = i sum_a pi_a lambda_a / 2, pi_a ~ N(0,1) keyed by (seed, GLOBAL site, mu, a): identical for every decomposition
__global__ __launch_bounds__(256) void momentum_gaussian_kernel(Geom g, double2* __restrict__ P, uint64_t seed) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    const uint64_t x = c[0] + g.origin[0], y = c[1] + g.origin[1], z = c[2] + g.origin[2], tt = c[3] + g.origin[3];
    const uint64_t gs = x + (uint64_t)g.gL[0] * (y + (uint64_t)g.gL[1] * (z + (uint64_t)g.gL[2] * tt));
    double pi[8];
    for (int a = 0; a < 4; a++) gauss2(rng_key(seed, gs * 4 + mu, 77, a), pi[2 * a], pi[2 * a + 1]);
    const double s3 = 0.5773502691896258;   // 1/sqrt(3)
    cd m[9];
    m[0] = mk(0.0, 0.5 * (pi[2] + s3 * pi[7]));
    m[4] = mk(0.0, 0.5 * (-pi[2] + s3 * pi[7]));
    m[8] = mk(0.0, -s3 * pi[7]);
    m[1] = mk(0.5 * pi[1], 0.5 * pi[0]);  m[3] = mk(-0.5 * pi[1], 0.5 * pi[0]);
    m[2] = mk(0.5 * pi[4], 0.5 * pi[3]);  m[6] = mk(-0.5 * pi[4], 0.5 * pi[3]);
    m[5] = mk(0.5 * pi[6], 0.5 * pi[5]);  m[7] = mk(-0.5 * pi[6], 0.5 * pi[5]);
    const size_t off = glink_off(g, p, mu, i);
    const int Gs = glink_stride(g);
#pragma unroll
    for (int e = 0; e < 9; e++) st(P + off + (size_t)e * Gs, m[e]);
}

// block partials of -Re tr P^2
__global__ __launch_bounds__(256) void momentum_action_kernel(Geom g, const double2* __restrict__ P, double* partial) {
    __shared__ double red[4];
    size_t off;
    double acc = 0.0;
    if (link_of_thread(g, off)) {
        const int Gs = glink_stride(g);
        cd m[9];
        load_m3(m, P + off, Gs);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) acc -= m[a * 3 + b].re * m[b * 3 + a].re - m[a * 3 + b].im * m[b * 3 + a].im;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

static int link_grid(const Geom& g) { return 2 * g.nch; }

}  // namespace lqcd

using namespace lqcd;

static int same_ctx(lqcd_gauge_t a, lqcd_gauge_t b, const char* who) {
    if (!(a && b && a->ctx == b->ctx && a != b)) { set_error(std::string(who) + ": need two distinct gauge-shaped fields of one context"); return LQCD_ERR_ARG; }
    return LQCD_OK;
}

extern "C" int lqcd_gauge_copy(lqcd_gauge_t dst, lqcd_gauge_t src) {      // substitute_U!(Uold, U) (standardHMC.jl:45)
    LQCHK(same_ctx(dst, src, "lqcd_gauge_copy"));
    lqcd_ctx_s* c = dst->ctx;
    HIPCHK(hipSetDevice(c->device));
    dst->version++;
    HIPCHK(hipMemcpyAsync(dst->data, src->data, src->elems * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// S_g = -(beta/3) sum_plaq Re tr U_p = -beta * 6 V_global * plaquette
extern "C" int lqcd_gauge_action(lqcd_gauge_t U, double beta, double* Sg) {
    ARGCHK(U && Sg, "lqcd_gauge_action: null argument");
    double plaq = 0;
    LQCHK(lqcd_gauge_plaquette(U, &plaq));
    const lqcd_ctx_s* c = U->ctx;
    *Sg = -beta * 6.0 * (double)c->gL[0] * c->gL[1] * c->gL[2] * c->gL[3] * plaq;
    return LQCD_OK;
}

// G_mu(n) = -(beta/6) U_mu(n) * (sum of the six staples)      (calc_dSdUmu! + mul!(temp, U, dSdUmu), AbstractMD.jl:108-110)
extern "C" int lqcd_gauge_force(lqcd_gauge_t out, lqcd_gauge_t U, double beta) {
    LQCHK(same_ctx(out, U, "lqcd_gauge_force"));
    lqcd_ctx_s* c = U->ctx;
    if (any_partitioned(c)) {
        set_error("lqcd_gauge_force: not available on a partitioned lattice yet (needs link halos in both directions)");
        return LQCD_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    out->version++;
    hipLaunchKernelGGL(gauge_force_kernel<false>, dim3(2 * c->geom.nch), dim3(256), 0, c->stream, c->geom, U->data, out->data, -beta / 6.0, 0.0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// P_update!(U, p, eps, md) (AbstractMD.jl:99-118) in one pass:  P += factor * TA(-(beta/6) U * staples); the force field is never stored
extern "C" int lqcd_momentum_add_gauge_force(lqcd_gauge_t P, double factor, lqcd_gauge_t U, double beta) {
    LQCHK(same_ctx(P, U, "lqcd_momentum_add_gauge_force"));
    lqcd_ctx_s* c = U->ctx;
    if (any_partitioned(c)) {
        set_error("lqcd_momentum_add_gauge_force: not available on a partitioned lattice yet (needs link halos in both directions)");
        return LQCD_ERR_UNSUPPORTED;
    }
    HIPCHK(hipSetDevice(c->device));
    P->version++;
    hipLaunchKernelGGL(gauge_force_kernel<true>, dim3(2 * c->geom.nch), dim3(256), 0, c->stream, c->geom, U->data, P->data, -beta / 6.0, factor);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// Traceless_antihermitian_add!(p, factor, G) (AbstractMD.jl:110,131):  P += factor * TA(G)
extern "C" int lqcd_momentum_add_ta(lqcd_gauge_t P, double factor, lqcd_gauge_t G) {
    LQCHK(same_ctx(P, G, "lqcd_momentum_add_ta"));
    lqcd_ctx_s* c = P->ctx;
    HIPCHK(hipSetDevice(c->device));
    P->version++;
    hipLaunchKernelGGL(momentum_add_ta_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, P->data, factor, G->data);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// U_update! (AbstractMD.jl:78-97): U <- exp(dt P) U
extern "C" int lqcd_gauge_exp_update(lqcd_gauge_t U, double dt, lqcd_gauge_t P) {
    LQCHK(same_ctx(U, P, "lqcd_gauge_exp_update"));
    lqcd_ctx_s* c = U->ctx;
    HIPCHK(hipSetDevice(c->device));
    U->version++;
    hipLaunchKernelGGL(link_exp_update_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, U->data, dt, P->data);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// gauss_distribution!(p) (standardMD.jl:86)
extern "C" int lqcd_momentum_gaussian(lqcd_gauge_t P, uint64_t seed) {
    ARGCHK(P, "lqcd_momentum_gaussian: null argument");
    lqcd_ctx_s* c = P->ctx;
    HIPCHK(hipSetDevice(c->device));
    P->version++;
    hipLaunchKernelGGL(momentum_gaussian_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, P->data, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// K = -sum tr P^2  (= md.p * md.p / 2, standardHMC.jl:49); summed over ranks
extern "C" int lqcd_momentum_action(lqcd_gauge_t P, double* K) {
    ARGCHK(P && K, "lqcd_momentum_action: null argument");
    lqcd_ctx_s* c = P->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int nb = link_grid(c->geom);
    ARGCHK(nb <= 2 * (2 * c->geom.Vh / 64 + 4096), "lqcd_momentum_action: partial buffer too small");
    hipLaunchKernelGGL(momentum_action_kernel, dim3(nb), dim3(256), 0, c->stream, c->geom, P->data, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 1, S_RED0, true, 0));
    HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *K = c->h_scal[0];
    return LQCD_OK;
}
