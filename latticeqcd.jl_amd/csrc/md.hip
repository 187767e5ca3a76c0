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

#include <algorithm>
#include <cmath>
#include <vector>

namespace lqcd {

typedef cd m3[9];
#ifndef LQCD_STAPLE_TWOROW
#define LQCD_STAPLE_TWOROW 1    // ... and two-row products inside it (r04)
#endif
#ifndef LQCD_STAPLE_NT
#define LQCD_STAPLE_NT 1        // round 6: the one-sweep form streams the momenta (read + written once) and the new links (written once) past the caches: the block
                                // U_update! P_update! U_update! 1.416 -> 1.359 ms at 32^3x64 (profiles/r06_staple_ab.log) -- the sweep is bound by the memory path, not by issue
#endif
typedef double v2d_md __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cd ld_stream(const double2* p) {
#if LQCD_STAPLE_NT
    const v2d_md v = __builtin_nontemporal_load(reinterpret_cast<const v2d_md*>(p));
    return mk(v.x, v.y);
#else
    return ld(p);
#endif
}
__device__ __forceinline__ void st_stream(double2* p, cd v) {
#if LQCD_STAPLE_NT
    const v2d_md t = {v.re, v.im};
    __builtin_nontemporal_store(t, reinterpret_cast<v2d_md*>(p));
#else
    st(p, v);
#endif
}
#ifndef LQCD_STAPLE_TILE_ROWS
#define LQCD_STAPLE_TILE_ROWS 6     // rows of a link in the tile's LDS copy times three: 6 = rows 0, 1 (48 KiB: THREE workgroups per CU at 146..151 VGPRs -- the default, -5 %),
                                    // 9 = all three rows (72 KiB, two workgroups per CU, no row-2 rebuild for the operands from LDS: -1.7 %; profiles/r06_staple_ab.log)
#endif
#ifndef LQCD_STAPLE_TILE_OCC
#define LQCD_STAPLE_TILE_OCC 3
#endif
#ifndef LQCD_STAPLE_TILE_NBR
#define LQCD_STAPLE_TILE_NBR 1      // 1: the other parity's links are in LDS too and the x neighbours come from there; 0 (experiment): own parity only (24 KiB)
#endif
#ifndef LQCD_STAPLE_TILE_Y
#define LQCD_STAPLE_TILE_Y 0        // 1: the y rows of the tile too, through generic pointers (flat loads): 256 VGPRs + 19..51 spilled, 1.483 ms per block against 1.329 -- off
#endif
#ifndef LQCD_STAPLE_TILE_BURST
#define LQCD_STAPLE_TILE_BURST 0    // tile form: 1 = all five neighbour links of a plane in one burst (214 VGPRs, two workgroups per CU: no gain); 0 = the lower staple's loads
                                    // behind the upper staple's sum -- one staple in registers at a time is what lets three workgroups share a CU
#endif
#ifndef LQCD_STAPLE_ROWS3
#define LQCD_STAPLE_ROWS3 0
#endif
#ifndef LQCD_STAPLE_BURST
#define LQCD_STAPLE_BURST 1     // two-row staple sweep: the five neighbour links of a plane in one load burst
#endif

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
// C = A^+ B
__device__ __forceinline__ void mm3_dn(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma_conj(t, A[k * 3 + a], B[k * 3 + b]);
            C[a * 3 + b] = t;
        }
}
// rows 0, 1 of C = A B / A B^+ / A^+ B (row 2 of C is left alone: a product of SU(3) matrices gets it from finish_u)
__device__ __forceinline__ void mm2(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma(t, A[a * 3 + k], B[k * 3 + b]);
            C[a * 3 + b] = t;
        }
}
__device__ __forceinline__ void mm2_nd(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma_conj(t, B[b * 3 + k], A[a * 3 + k]);
            C[a * 3 + b] = t;
        }
}
__device__ __forceinline__ void mm2_dn(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) cfma_conj(t, A[k * 3 + a], B[k * 3 + b]);
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

// links of a field that is known to be on the group (lqcd_gauge_s::unitary_version): rows 0 and 1 from memory, row 2 = conj(row 0 x row 1) --
// two thirds of the bytes through the L1 / L2 path, which is what bounds the staple sweep (60 neighbour-link loads per site)
template <bool R2>
__device__ __forceinline__ void load_u(cd (&u)[9], const double2* __restrict__ base, int stride) {
    if constexpr (!R2) { load_m3(u, base, stride); return; }
#pragma unroll
    for (int e = 0; e < 6; e++) u[e] = ld(base + (size_t)e * stride);
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const cd x = cmul(u[b1], u[3 + b2]) - cmul(u[b2], u[3 + b1]);
        u[6 + b] = mk(x.re, -x.im);
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

// arguments of the staple kernels.  Partitioned lattice: ghost[lam] = the x_lam = 0 slice of all links of the +lam neighbour
// ([parity][nu][9][Fh], fields.hip gauge_face_pack), wrecv[nu] = the lower staples W_{mu nu} of the -nu neighbour's upper face
// ([parity of that site][mu][9][Fh]); both null on a single GPU.
struct GFArgs {
    Geom g;
    const double2* U;
    double2* out;
    double coef, factor;
    const double2* ghost[4];
    const double2* wrecv[4];
    double2* wsend[4];
    BlockMap bm;             // workgroup -> chunk map of the sweep (tunable md_remap: 1 = the Dslash kernels' XCD tile sweep, 0 = plain order)
    int mu_only, mu_out;     // MODE 2 (calc_dSdUmu!): only direction mu_only, staple sum written to direction slot mu_out of `out`
    // EXPU instances (P_update! and the U_update! that follows it in ONE sweep): uout <- exp(dt P_new) U, a second link buffer (the sweep still reads the old links)
    double2* uout;
    double dt;
    unsigned* notproj;
    int reunit;
};

// U_nu at the site c + dir_hat: local, or from the forward ghost slice when the step leaves the rank
template <bool PART = true, bool R2 = false>
__device__ __forceinline__ void link_fwd(cd (&u)[9], const GFArgs& k, const int (&c)[4], int dir, int nu) {
    const Geom& g = k.g;
    int d[4] = {c[0], c[1], c[2], c[3]};
    d[dir] += 1;
    if (d[dir] == g.L[dir]) {
        d[dir] = 0;
        if (PART && g.part[dir]) {
            const int p = (d[0] + d[1] + d[2] + d[3]) & 1, Fh = face_half_sites(g, dir), f = coords_to_face(g, dir, d);
            const double2* b = k.ghost[dir] + ((size_t)(p * 4 + nu) * 9) * Fh + f;
#pragma unroll
            for (int e = 0; e < 9; e++) u[e] = ld(b + (size_t)e * Fh);
            return;
        }
    }
    load_u<R2>(u, link_at(g, k.U, d, nu), glink_stride(g));
}

// lower staple seen from the site m = n - nu_hat:  W_{mu nu}(m) = U_nu(m+mu)^+ U_mu(m)^+ U_nu(m)
template <bool PART = true, bool R2 = false>
__device__ __forceinline__ void lower_staple_at(cd (&w)[9], const GFArgs& k, const int (&m)[4], int mu, int nu) {
    cd u1[9], u2[9], u3[9], t1[9];
    const int Gs = glink_stride(k.g);
    link_fwd<PART, R2>(u1, k, m, mu, nu);
    load_u<R2>(u2, link_at(k.g, k.U, m, mu), Gs);
    load_u<R2>(u3, link_at(k.g, k.U, m, nu), Gs);
    mm3_dd(t1, u1, u2);
    mm3(w, t1, u3);
}

// The partitioned-lattice instance: direction and plane stay run-time values here -- with the ghost-link and received-staple branches the
// fully templated body below needs > 256 registers (1000+ spilled); this form holds them in 256 without scratch.
template <int MODE>
__global__ __launch_bounds__(256) void gauge_force_kernel_part(GFArgs k) {
    constexpr bool FUSE_TA = MODE == 1 || MODE == 3;
    const Geom& g = k.g;
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = MODE >= 2 ? k.mu_only : (int)(threadIdx.x >> 6);
    if (i >= g.Vh) return;
    const int Gs = glink_stride(g);
    int c[4];
    cb_to_coords(g, p, i, c);
    cd A[9];
#pragma unroll
    for (int e = 0; e < 9; e++) A[e] = mk(0.0, 0.0);
    for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        cd u1[9], u2[9], u3[9], t1[9], t2[9];
        link_fwd(u1, k, c, mu, nu);                         // U_nu(n+mu)
        link_fwd(u2, k, c, nu, mu);                         // U_mu(n+nu)
        load_m3(u3, link_at(g, k.U, c, nu), Gs);
        mm3_nd(t1, u1, u2);
        mm3_nd(t2, t1, u3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        if (c[nu] == 0 && g.part[nu]) {                     // n - nu lives on the -nu neighbour: its W arrived with the exchange
            const int Fh = face_half_sites(g, nu), f = coords_to_face(g, nu, c);
            const double2* b = k.wrecv[nu] + ((size_t)((1 - p) * 4 + mu) * 9) * Fh + f;
#pragma unroll
            for (int e = 0; e < 9; e++) t2[e] = ld(b + (size_t)e * Fh);
        } else {
            int m[4] = {c[0], c[1], c[2], c[3]};
            shift(m, g, nu, -1);
            lower_staple_at(t2, k, m, mu, nu);
        }
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
    }
    const double coef = k.coef;
    if constexpr (MODE == 2) {
        double2* o2 = k.out + glink_off(g, p, k.mu_out, i);
#pragma unroll
        for (int e = 0; e < 9; e++) st(o2 + (size_t)e * Gs, mk(coef * A[e].re, coef * A[e].im));
        return;
    }
    cd um[9], r[9];
    load_m3(um, k.U + glink_off(g, p, mu, i), Gs);
    mm3(r, um, A);
    double2* o = k.out + glink_off(g, p, MODE == 3 ? k.mu_out : mu, i);
    if constexpr (!FUSE_TA) {
#pragma unroll
        for (int e = 0; e < 9; e++) st(o + (size_t)e * Gs, mk(coef * r[e].re, coef * r[e].im));
    } else {
        cd a[9];
        const double f = 0.5 * coef * k.factor;
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

__device__ __forceinline__ void exp_m3(cd (&e)[9], cd (&x)[9], double dt);
__device__ __forceinline__ void project_if_on_group(cd (&t)[9], unsigned* notproj);

// one plane (MU, NU) of the staple sum of link (n, MU): upper staple U_nu(n+mu) U_mu(n+nu)^+ U_nu(n)^+ and lower staple W_{mu nu}(n - nu).
// MU and NU are compile-time: every index into the by-value argument struct and the coordinate arrays is static.
// rows 0, 1 of a link as they come from memory (row 2 is rebuilt when the link is used: finish_u)
__device__ __forceinline__ void load_u_raw(cd (&u)[9], const double2* __restrict__ base, int stride) {
#pragma unroll
    for (int e = 0; e < 6; e++) u[e] = ld(base + (size_t)e * stride);
}
__device__ __forceinline__ void finish_u(cd (&u)[9]) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const cd x = cmul(u[b1], u[3 + b2]) - cmul(u[b2], u[3 + b1]);
        u[6 + b] = mk(x.re, -x.im);
    }
}
__device__ __forceinline__ const double2* link_at_shifted(const Geom& g, const double2* __restrict__ U, const int (&c)[4], int dir, int step, int mu) {
    int d[4] = {c[0], c[1], c[2], c[3]};
    shift(d, g, dir, step);
    return link_at(g, U, d, mu);
}

template <int MODE, int MU, int NU, bool PART, bool R2>
__device__ __forceinline__ void staple_plane(cd (&A)[9], const GFArgs& k, int (&c)[4], int p, int lane, const double2 (*own)[9][64]) {
    if constexpr (MU != NU && R2 && !PART && MODE < 2 && LQCD_STAPLE_BURST) {
        // single GPU, links on the group: the five neighbour links of the plane are issued as ONE burst of two-row loads (30 x 1 KiB per wave: one
        // memory round trip per plane instead of two), row 2 is rebuilt as each link is consumed
        const Geom& g = k.g;
        const int Gs = glink_stride(g);
        cd a1[9], a2[9], l1[9], l2[9], l3[9], u3[9], t1[9], t2[9];
        int m[4] = {c[0], c[1], c[2], c[3]};
        shift(m, g, NU, -1);
#if LQCD_STAPLE_ROWS3      // experiment (round 6, measured SLOWER: 1.416 -> 1.494 ms per block, profiles/r06_staple_ab.log): the links that enter a product as its right operand
                           // come with all three rows -- three finish_u less per plane (-13 % VALU), 39 instead of 30 loads: the bytes cost more than the instructions save
        load_u_raw(a1, link_at_shifted(g, k.U, c, MU, 1, NU), Gs);      // U_nu(n+mu)
        load_m3(a2, link_at_shifted(g, k.U, c, NU, 1, MU), Gs);         // U_mu(n+nu)
        load_m3(l1, link_at_shifted(g, k.U, m, MU, 1, NU), Gs);         // U_nu(m+mu)
        load_u_raw(l2, link_at(g, k.U, m, MU), Gs);                     // U_mu(m)
        load_m3(l3, link_at(g, k.U, m, NU), Gs);                        // U_nu(m)
#else
        load_u_raw(a1, link_at_shifted(g, k.U, c, MU, 1, NU), Gs);      // U_nu(n+mu)
        load_u_raw(a2, link_at_shifted(g, k.U, c, NU, 1, MU), Gs);      // U_mu(n+nu)
        load_u_raw(l1, link_at_shifted(g, k.U, m, MU, 1, NU), Gs);      // U_nu(m+mu)
        load_u_raw(l2, link_at(g, k.U, m, MU), Gs);                     // U_mu(m)
        load_u_raw(l3, link_at(g, k.U, m, NU), Gs);                     // U_nu(m)
#endif
#pragma unroll
        for (int e = 0; e < 9; e++) { const double2 t = own[NU][e][lane]; u3[e] = mk(t.x, t.y); }
#if LQCD_STAPLE_TWOROW
        // every staple is a product of SU(3) matrices: rows 0, 1 of each product (two thirds of the multiplications), row 2 rebuilt like a link's
        if (!LQCD_STAPLE_ROWS3) finish_u(a2);
        mm2_nd(t1, a1, a2);         // rows 0, 1 of a1 a2^+ need rows 0, 1 of a1 only
        mm2_nd(t2, t1, u3);
        finish_u(t2);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        if (!LQCD_STAPLE_ROWS3) finish_u(l1);
        mm2(t1, l2, l1);            // Q = l2 l1, rows 0, 1
        finish_u(t1);
        if (!LQCD_STAPLE_ROWS3) finish_u(l3);
        mm2_dn(t2, t1, l3);         // rows 0, 1 of Q^+ l3 = l1^+ l2^+ l3
        finish_u(t2);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
#else
        finish_u(a1); finish_u(a2);
        mm3_nd(t1, a1, a2);
        mm3_nd(t2, t1, u3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        finish_u(l1); finish_u(l2); finish_u(l3);
        mm3_dd(t1, l1, l2);
        mm3(t2, t1, l3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
#endif
        asm volatile("" : "+v"(c[0]), "+v"(A[0].re), "+v"(A[0].im), "+v"(A[4].re), "+v"(A[4].im), "+v"(A[8].re), "+v"(A[8].im));
    } else
    if constexpr (MU != NU) {
        const Geom& g = k.g;
        const int Gs = glink_stride(g);
        cd u1[9], u2[9], u3[9], t1[9], t2[9];
        link_fwd<PART, R2>(u1, k, c, MU, NU);               // U_nu(n+mu)
        link_fwd<PART, R2>(u2, k, c, NU, MU);               // U_mu(n+nu)
        if constexpr (MODE < 2) {
#pragma unroll
            for (int e = 0; e < 9; e++) { const double2 t = own[NU][e][lane]; u3[e] = mk(t.x, t.y); }
        } else {
            load_u<R2>(u3, link_at(g, k.U, c, NU), Gs);
        }
        mm3_nd(t1, u1, u2);
        mm3_nd(t2, t1, u3);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        if (PART && c[NU] == 0 && g.part[NU]) {             // n - nu lives on the -nu neighbour: its W arrived with the exchange
            const int Fh = face_half_sites(g, NU), f = coords_to_face(g, NU, c);
            const double2* b = k.wrecv[NU] + ((size_t)((1 - p) * 4 + MU) * 9) * Fh + f;
#pragma unroll
            for (int e = 0; e < 9; e++) t2[e] = ld(b + (size_t)e * Fh);
        } else {
            int m[4] = {c[0], c[1], c[2], c[3]};
            shift(m, g, NU, -1);
            lower_staple_at<PART, R2>(t2, k, m, MU, NU);
        }
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        // the next plane's five link loads wait for this plane's sum: without the tie the scheduler hoists all fifteen of a direction (and
        // spills hundreds of registers); one plane in flight per wave, the other waves of the CU cover its latency
        asm volatile("" : "+v"(c[0]), "+v"(A[0].re), "+v"(A[0].im), "+v"(A[4].re), "+v"(A[4].im), "+v"(A[8].re), "+v"(A[8].im));
    }
}

// ---- TILE form of the staple plane (round 6): the workgroup keeps rows 0, 1 of the links of BOTH parities of its chunk in LDS -- a chunk of 64 checkerboard sites is
// 64 / XH whole x-rows, so with the other parity it is a closed (x, y) tile of 128 sites -- and the neighbour links at n + x / n - x (12 of the 60 loads per site) are
// read from there; the loads of a plane come in two groups (upper staple, then lower staple), which holds the kernel at 146..151 VGPRs: THREE workgroups per CU
// instead of two.  One Sexton-Weingarten block at 32^3x64: 1.339 -> 1.272 ms (profiles/r06_staple_ab.log).  The y rows of the tile (another 10.5 loads) would need a
// per-lane choice between LDS and global memory: generic pointers / flat loads, 256 registers and spills -- measured 12 % SLOWER (LQCD_STAPLE_TILE_Y).
// lane of n + x / n - x inside the chunk of the other parity (x = 2 xh + q wraps inside its row), of n +- y (same xh, next / previous row: the caller checks the row)
__device__ __forceinline__ int tile_lane_px(int lane, int xh, int q, int XH) { return q ? (xh + 1 == XH ? lane - (XH - 1) : lane + 1) : lane; }
__device__ __forceinline__ int tile_lane_mx(int lane, int xh, int q, int XH) { return q ? lane : (xh == 0 ? lane + XH - 1 : lane - 1); }
template <int MODE, int MU, int NU>
__device__ __forceinline__ void staple_plane_tile(cd (&A)[9], const GFArgs& k, int (&c)[4], int lane, const double2 (*own2)[4][LQCD_STAPLE_TILE_ROWS][64]) {
    if constexpr (MU != NU) {
        const Geom& g = k.g;
        const int Gs = glink_stride(g), XH = g.XH, rows = 64 / XH;      // (the LDS copy has the component stride of the field: 64 elements)
        const int yr = lane / XH, xh = lane - yr * XH, q = c[0] & 1;
        // where the five neighbour links of the plane live: in the tile (LDS, lane index there) or outside (global memory).  x: always inside (compile time);
        // y: the row decides per lane -- ONE load sequence through a generic pointer that is an LDS address in some lanes and a global one in others
        const bool up_mu = MU == 0 || (MU == 1 && yr + 1 < rows), up_nu = NU == 0 || (NU == 1 && yr + 1 < rows), dn_nu = NU == 0 || (NU == 1 && yr >= 1);
        const int l_pmu = MU == 0 ? tile_lane_px(lane, xh, q, XH) : lane + XH;        // n + mu (other parity)
        const int l_pnu = NU == 0 ? tile_lane_px(lane, xh, q, XH) : lane + XH;        // n + nu
        const int l_m = NU == 0 ? tile_lane_mx(lane, xh, q, XH) : lane - XH;          // m = n - nu
        const bool in_l1 = (MU == 0 && NU == 1) ? yr >= 1 : (MU == 1 && NU == 0) ? yr + 1 < rows : false;      // m + mu (this parity): only the (x, y) planes stay inside
        int l_l1 = 0;
        if constexpr (MU == 0 && NU == 1) l_l1 = tile_lane_px(lane - XH, xh, q, XH);                 // a y hop keeps x: same xh, same q
        if constexpr (MU == 1 && NU == 0) l_l1 = tile_lane_mx(lane, xh, q, XH) + XH;
        int m[4] = {c[0], c[1], c[2], c[3]};
        shift(m, g, NU, -1);
#if LQCD_STAPLE_TILE_Y       // the y rows too: one generic pointer per operand, an LDS address in some lanes and a global one in others (flat loads)
        constexpr int YMAX = 1;
#elif LQCD_STAPLE_TILE_NBR   // x hops only: every select is made at compile time (ds_read or global_load, never a flat load)
        constexpr int YMAX = 0;
#else
        constexpr int YMAX = -1;
#endif
        const double2* pa1 = (MU <= YMAX && up_mu) ? &own2[LQCD_STAPLE_TILE_NBR][NU][0][l_pmu & 63] : link_at_shifted(g, k.U, c, MU, 1, NU);      // U_nu(n+mu)
        const double2* pa2 = (NU <= YMAX && up_nu) ? &own2[LQCD_STAPLE_TILE_NBR][MU][0][l_pnu & 63] : link_at_shifted(g, k.U, c, NU, 1, MU);      // U_mu(n+nu)
        const double2* pl1 = (YMAX && in_l1) ? &own2[0][NU][0][l_l1 & 63] : link_at_shifted(g, k.U, m, MU, 1, NU);             // U_nu(m+mu)
        const double2* pl2 = (NU <= YMAX && dn_nu) ? &own2[LQCD_STAPLE_TILE_NBR][MU][0][l_m & 63] : link_at(g, k.U, m, MU);                       // U_mu(m)
        const double2* pl3 = (NU <= YMAX && dn_nu) ? &own2[LQCD_STAPLE_TILE_NBR][NU][0][l_m & 63] : link_at(g, k.U, m, NU);                       // U_nu(m)
        cd a1[9], a2[9], l1[9], l2[9], l3[9], u3[9], t1[9], t2[9];
        // one burst for everything that may come from global memory; what is in LDS for every lane (the x cases) is read where it is used (short latency, no
        // registers held across the burst)
        if constexpr (MU != 0) load_u_raw(a1, pa1, Gs);
        if constexpr (NU != 0) load_u_raw(a2, pa2, Gs);
#if LQCD_STAPLE_TILE_BURST
        load_u_raw(l1, pl1, Gs);
        if constexpr (NU != 0) { load_u_raw(l2, pl2, Gs); load_u_raw(l3, pl3, Gs); }
#endif
        if constexpr (MU == 0) load_u_raw(a1, pa1, Gs);
        if constexpr (NU == 0) { if constexpr (LQCD_STAPLE_TILE_ROWS == 9) load_m3(a2, pa2, Gs); else { load_u_raw(a2, pa2, Gs); finish_u(a2); } }      // x: from the LDS copy
        else finish_u(a2);
        mm2_nd(t1, a1, a2);         // rows 0, 1 of a1 a2^+ need rows 0, 1 of a1 only
#pragma unroll
        for (int e = 0; e < LQCD_STAPLE_TILE_ROWS; e++) { const double2 t = own2[0][NU][e][lane]; u3[e] = mk(t.x, t.y); }
        if constexpr (LQCD_STAPLE_TILE_ROWS == 6) finish_u(u3);
        mm2_nd(t2, t1, u3);
        finish_u(t2);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
#if !LQCD_STAPLE_TILE_BURST
        asm volatile("" : "+v"(A[0].re), "+v"(A[4].im), "+v"(A[8].re));      // the lower staple's loads behind the upper staple's sum: 256 registers hold one staple at a time
        load_u_raw(l1, pl1, Gs);
        if constexpr (NU != 0) { load_u_raw(l2, pl2, Gs); load_u_raw(l3, pl3, Gs); }
#endif
        finish_u(l1);
        if constexpr (NU == 0) load_u_raw(l2, pl2, Gs);
        mm2(t1, l2, l1);            // Q = l2 l1, rows 0, 1
        finish_u(t1);
        if constexpr (NU == 0) { if constexpr (LQCD_STAPLE_TILE_ROWS == 9) load_m3(l3, pl3, Gs); else { load_u_raw(l3, pl3, Gs); finish_u(l3); } }
        else finish_u(l3);
        mm2_dn(t2, t1, l3);         // rows 0, 1 of Q^+ l3 = l1^+ l2^+ l3
        finish_u(t2);
#pragma unroll
        for (int e = 0; e < 9; e++) A[e] = A[e] + t2[e];
        asm volatile("" : "+v"(c[0]), "+v"(A[0].re), "+v"(A[0].im), "+v"(A[4].re), "+v"(A[4].im), "+v"(A[8].re), "+v"(A[8].im));
    }
}

template <int MODE, int MU, bool PART, bool R2, bool EXPU = false, bool TILE = false>
__device__ __forceinline__ void staple_links(const GFArgs& k, int p, int i, int lane, const double2 (*own)[9][64], const double2 (*own2)[4][LQCD_STAPLE_TILE_ROWS][64] = nullptr) {
    constexpr bool FUSE_TA = MODE == 1 || MODE == 3;
    const Geom& g = k.g;
    const int Gs = glink_stride(g);
    int c[4];
    cb_to_coords(g, p, i, c);
    cd A[9];
#pragma unroll
    for (int e = 0; e < 9; e++) A[e] = mk(0.0, 0.0);
    if constexpr (TILE) {
        staple_plane_tile<MODE, MU, 0>(A, k, c, lane, own2);
        staple_plane_tile<MODE, MU, 1>(A, k, c, lane, own2);
        staple_plane_tile<MODE, MU, 2>(A, k, c, lane, own2);
        staple_plane_tile<MODE, MU, 3>(A, k, c, lane, own2);
    } else {
        staple_plane<MODE, MU, 0, PART, R2>(A, k, c, p, lane, own);
        staple_plane<MODE, MU, 1, PART, R2>(A, k, c, p, lane, own);
        staple_plane<MODE, MU, 2, PART, R2>(A, k, c, p, lane, own);
        staple_plane<MODE, MU, 3, PART, R2>(A, k, c, p, lane, own);
    }
    const double coef = k.coef;
    if constexpr (MODE == 2) {
        double2* o2 = k.out + glink_off(g, p, k.mu_out, i);
#pragma unroll
        for (int e = 0; e < 9; e++) st(o2 + (size_t)e * Gs, mk(coef * A[e].re, coef * A[e].im));
        return;
    }
    cd um[9], r[9];
    if constexpr (MODE == 3) load_m3(um, k.U + glink_off(g, p, MU, i), Gs);      // one direction per launch: the link comes from memory
    else if constexpr (TILE) {
#pragma unroll
        for (int e = 0; e < LQCD_STAPLE_TILE_ROWS; e++) { const double2 t = own2[0][MU][e][lane]; um[e] = mk(t.x, t.y); }
        if constexpr (LQCD_STAPLE_TILE_ROWS == 6) finish_u(um);
    } else {
#pragma unroll
        for (int e = 0; e < 9; e++) { const double2 t = own[MU][e][lane]; um[e] = mk(t.x, t.y); }      // U_mu(n) from LDS
    }
    mm3(r, um, A);
    double2* o = k.out + glink_off(g, p, MODE == 3 ? k.mu_out : MU, i);
    if constexpr (!FUSE_TA) {
#pragma unroll
        for (int e = 0; e < 9; e++) st(o + (size_t)e * Gs, mk(coef * r[e].re, coef * r[e].im));
    } else {
        cd a[9];
        const double f = 0.5 * coef * k.factor;
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int y = 0; y < 3; y++) a[x * 3 + y] = mk(f * (r[x * 3 + y].re - r[y * 3 + x].re), f * (r[x * 3 + y].im + r[y * 3 + x].im));
        const double tr = (a[0].im + a[4].im + a[8].im) / 3.0;
        a[0].im -= tr; a[4].im -= tr; a[8].im -= tr;
        // the momenta are anti-Hermitian (every writer of a momentum field stores TA matrices: the reference's p[mu] is a TA field by type), and so is the increment: the upper
        // triangle is read, the lower one follows -- bit for bit what the nine sums gave -- and 48 of the 144 bytes per link stay unread (profiles/r06_pmc_staple.log)
        auto addp = [&](auto E) {
            constexpr int e = decltype(E)::value;
            const cd pv = EXPU ? ld_stream(o + (size_t)e * Gs) : ld(o + (size_t)e * Gs);
            a[e] = mk(pv.re + a[e].re, pv.im + a[e].im);
        };
        addp(std::integral_constant<int, 0>()); addp(std::integral_constant<int, 1>()); addp(std::integral_constant<int, 2>());
        addp(std::integral_constant<int, 4>()); addp(std::integral_constant<int, 5>()); addp(std::integral_constant<int, 8>());
        a[3] = mk(-a[1].re, a[1].im); a[6] = mk(-a[2].re, a[2].im); a[7] = mk(-a[5].re, a[5].im);
#pragma unroll
        for (int e = 0; e < 9; e++) {
            if constexpr (EXPU) st_stream(o + (size_t)e * Gs, a[e]); else st(o + (size_t)e * Gs, a[e]);
        }
        if constexpr (EXPU) {      // the link update that follows this momentum update: exp(dt P_new) U_mu(n) into the second link buffer
            cd ex[9], t[9];
            exp_m3(ex, a, k.dt);
            mm3(t, ex, um);
            if (k.reunit) project_if_on_group(t, k.notproj);
            double2* uo = k.uout + glink_off(g, p, MU, i);
#pragma unroll
            for (int e = 0; e < 9; e++) st_stream(uo + (size_t)e * Gs, t[e]);
        }
    }
}

// out_mu(n) = coef * U_mu(n) * sum_{nu != mu} [ U_nu(n+mu) U_mu(n+nu)^+ U_nu(n)^+  +  W_{mu nu}(n - nu) ]
// workgroup = 64 sites of one parity x 4 waves (wave = mu, dispatched to a compile-time direction); the four links of the site go through
// LDS once (each wave loads its own direction: 36 KiB), the 6 x 2 neighbour links of a plane are re-used across waves/sites through L2.
// MODE 0: out = G.   MODE 1: out (the momenta) += factor * TA(G) -- P_update! in one pass, G never stored.
// MODE 2 (64-thread blocks, one direction): out[mu_out] = coef * (sum of the six staples of direction mu_only) -- the reference's
// calc_dSdUmu!(dSdUmu, gauge_action, mu, U) (AbstractMD.jl:108); the caller multiplies by U[mu] itself (mul!, :109).
// MODE 3 (64-thread blocks, one direction): out[mu_out] += factor * TA(coef U_mu * staples) -- the three calls of the reference's P_update! for one
// direction (calc_dSdUmu!, mul!, Traceless_antihermitian_add!: AbstractMD.jl:108-110) in one pass (lqcd_link_add_ta_staple).
#ifndef LQCD_STAPLE_OCC
#define LQCD_STAPLE_OCC 2
#endif
template <int MODE, bool PART, bool R2 = false, bool EXPU = false>
__global__ __launch_bounds__(256, LQCD_STAPLE_OCC) void gauge_force_kernel(GFArgs k) {
    const Geom& g = k.g;
    int chunk, p;
    block_map(k.bm, blockIdx.x, chunk, p);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const int mu = MODE >= 2 ? k.mu_only : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool valid = i < g.Vh;
    __shared__ double2 own[MODE >= 2 ? 1 : 4][9][64];
    if constexpr (MODE < 2) {
        if (valid) {
            cd um[9];
            load_u<R2>(um, k.U + glink_off(g, p, mu, i), glink_stride(g));
#pragma unroll
            for (int e = 0; e < 9; e++) own[mu][e][lane] = mk2(um[e].re, um[e].im);
        }
        __syncthreads();
    }
    if (!valid) return;
    switch (mu) {
    case 0: staple_links<MODE, 0, PART, R2, EXPU>(k, p, i, lane, own); break;
    case 1: staple_links<MODE, 1, PART, R2, EXPU>(k, p, i, lane, own); break;
    case 2: staple_links<MODE, 2, PART, R2, EXPU>(k, p, i, lane, own); break;
    default: staple_links<MODE, 3, PART, R2, EXPU>(k, p, i, lane, own); break;
    }
}

// the TILE form (single GPU, links on the group, chunks of whole x-rows): MODE 0 / 1, optionally with the link update behind it.  LDS: the links of both parities of
// the chunk, all three rows: 72 KiB per workgroup, two workgroups per CU.
template <int MODE, bool EXPU>
__global__ __launch_bounds__(256, LQCD_STAPLE_TILE_OCC) void gauge_force_kernel_tile(GFArgs k) {
    const Geom& g = k.g;
    int chunk, p;
    block_map(k.bm, blockIdx.x, chunk, p);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const int mu = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    __shared__ double2 own2[1 + LQCD_STAPLE_TILE_NBR][4][LQCD_STAPLE_TILE_ROWS][64];      // [0]: this workgroup's parity, [1]: the other one
    {
        cd um[9], uo[9];
        load_u<true>(um, k.U + glink_off(g, p, mu, i), glink_stride(g));
        if (LQCD_STAPLE_TILE_NBR) load_u<true>(uo, k.U + glink_off(g, 1 - p, mu, i), glink_stride(g));
#pragma unroll
        for (int e = 0; e < LQCD_STAPLE_TILE_ROWS; e++) { own2[0][mu][e][lane] = mk2(um[e].re, um[e].im); if (LQCD_STAPLE_TILE_NBR) own2[LQCD_STAPLE_TILE_NBR][mu][e][lane] = mk2(uo[e].re, uo[e].im); }
    }
    __syncthreads();
    switch (mu) {
    case 0: staple_links<MODE, 0, false, true, EXPU, true>(k, p, i, lane, nullptr, own2); break;
    case 1: staple_links<MODE, 1, false, true, EXPU, true>(k, p, i, lane, nullptr, own2); break;
    case 2: staple_links<MODE, 2, false, true, EXPU, true>(k, p, i, lane, nullptr, own2); break;
    default: staple_links<MODE, 3, false, true, EXPU, true>(k, p, i, lane, nullptr, own2); break;
    }
}
// the tile form applies: a chunk is 64 / XH whole x-rows of one (z, t) plane (XH a divisor of 64, the rows of a plane divide into chunks, every chunk full)
static bool staple_tile_ok(lqcd_ctx_s* c) {
    const Geom& g = c->geom;
    return c->tun.staple_tile && g.XH >= 1 && g.XH <= 64 && 64 % g.XH == 0 && g.L[1] % (64 / g.XH) == 0 && g.Vh % 64 == 0 && !any_partitioned(c);
}

// upper nu-face of a partitioned direction nu = blockIdx.y: W_{mu nu}(m) for the three mu != nu, packed for the +nu neighbour
__global__ __launch_bounds__(128) void staple_face_kernel(GFArgs k) {
    const Geom& g = k.g;
    const int nu = blockIdx.y;
    if (!g.part[nu]) return;
    const int Fh = face_half_sites(g, nu);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * Fh) return;
    const int p = t / Fh, f = t - p * Fh;
    int m[4];
    face_to_coords(g, nu, g.L[nu] - 1, p, f, m);
    const int mc[4] = {m[0], m[1], m[2], m[3]};
    for (int mu = 0; mu < 4; mu++) {
        if (mu == nu) continue;
        cd w[9];
        lower_staple_at(w, k, mc, mu, nu);
        double2* b = k.wsend[nu] + ((size_t)(p * 4 + mu) * 9) * Fh + f;
#pragma unroll
        for (int e = 0; e < 9; e++) st(b + (size_t)e * Fh, w[e]);
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
    // (the momenta are anti-Hermitian and so is the increment: the upper triangle is read, the lower one follows -- the same bits, see staple_links)
    cd o[9];
    auto addp = [&](auto E) {
        constexpr int e = decltype(E)::value;
        const cd pv = ld(P + off + (size_t)e * Gs);
        o[e] = mk(fma(cf, a[e].re, pv.re), fma(cf, a[e].im, pv.im));
    };
    addp(std::integral_constant<int, 0>()); addp(std::integral_constant<int, 1>()); addp(std::integral_constant<int, 2>());
    addp(std::integral_constant<int, 4>()); addp(std::integral_constant<int, 5>()); addp(std::integral_constant<int, 8>());
    o[3] = mk(-o[1].re, o[1].im); o[6] = mk(-o[2].re, o[2].im); o[7] = mk(-o[5].re, o[5].im);
#pragma unroll
    for (int e = 0; e < 9; e++) st(P + off + (size_t)e * Gs, o[e]);
}

// exp(dt P).  Below max-abs-row-sum norm 2 of X = dt P (an MD step has a few 1e-2) the Taylor series is summed through the Cayley-Hamilton identity
// X^3 = t X^2 - s X + d I (t = tr X, s = (t^2 - tr X^2)/2, d = det X; true for every 3x3 matrix, nothing assumed about P): X^n = al_n I + be_n X + ga_n X^2
// with the scalar recurrence al' = d ga, be' = al - s ga, ga' = be + t ga, so exp X = a0 I + a1 X + a2 X^2 costs ONE matrix product and a dozen scalar
// steps instead of a matrix product per term (r04: the link update inside the staple sweep is ALU time, 1.78 -> 1.63 ms already from a shorter series).
// Terms: until nrm^(n+1)/(n+1)! < 1e-18, two more for the n^2 growth of the coefficients (<= 6e-16 from scipy's expm up to norm 2, near-degenerate spectra included).
// Norm >= 2: 24 terms in Horner form.
#ifndef LQCD_EXP_CH
#define LQCD_EXP_CH 1
#endif
__device__ __forceinline__ void exp_m3(cd (&e)[9], cd (&x)[9], double dt) {     // e = exp(dt x); x is scaled in place
    cd t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) x[k] = mk(dt * x[k].re, dt * x[k].im);
    double nrm = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
        nrm = fmax(nrm, (fabs(x[a * 3].re) + fabs(x[a * 3].im)) + (fabs(x[a * 3 + 1].re) + fabs(x[a * 3 + 1].im)) + (fabs(x[a * 3 + 2].re) + fabs(x[a * 3 + 2].im)));
#if LQCD_EXP_CH
    if (nrm < 2.0) {
        const int nt = (nrm < 0.009 ? 6 : nrm < 0.04 ? 8 : nrm < 0.11 ? 10 : nrm < 0.2 ? 12 : nrm < 0.5 ? 16 : nrm < 1.0 ? 20 : 28) + 2;
        mm3(t, x, x);
        const cd tr = x[0] + x[4] + x[8], tr2 = t[0] + t[4] + t[8], trtr = cmul(tr, tr);
        const cd s = mk(0.5 * (trtr.re - tr2.re), 0.5 * (trtr.im - tr2.im));
        const cd d = cmul(x[0], cmul(x[4], x[8]) - cmul(x[5], x[7])) - cmul(x[1], cmul(x[3], x[8]) - cmul(x[5], x[6])) +
                     cmul(x[2], cmul(x[3], x[7]) - cmul(x[4], x[6]));
        cd al = mk(1.0, 0.0), be = mk(0.0, 0.0), ga = mk(0.0, 0.0), a0 = al, a1 = be, a2 = ga;
        double f = 1.0;
        for (int n = 1; n <= nt; n++) {
            const cd al2 = cmul(d, ga), be2 = al - cmul(s, ga), ga2 = be + cmul(tr, ga);
            al = al2; be = be2; ga = ga2;
            f /= (double)n;
            a0 = mk(fma(f, al.re, a0.re), fma(f, al.im, a0.im));
            a1 = mk(fma(f, be.re, a1.re), fma(f, be.im, a1.im));
            a2 = mk(fma(f, ga.re, a2.re), fma(f, ga.im, a2.im));
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            e[k] = cmul(a1, x[k]) + cmul(a2, t[k]);
            if (k % 4 == 0) e[k] = e[k] + a0;
        }
        return;
    }
#endif
#pragma unroll
    for (int k = 0; k < 9; k++) e[k] = mk((k % 4 == 0) ? 1.0 : 0.0, 0.0);
    for (int n = nrm < 4.0 ? 36 : 60; n >= 1; n--) {
        mm3(t, x, e);
        const double inv = 1.0 / (double)n;
#pragma unroll
        for (int k = 0; k < 9; k++) e[k] = mk(((k % 4 == 0) ? 1.0 : 0.0) + inv * t[k].re, inv * t[k].im);
    }
}
// Back onto SU(3): rows 0 and 1 by Gram-Schmidt, row 2 = conj(row 0 x row 1) with the arithmetic of the 12-real gate (fields.hip
// gauge_compress12), so a projected link passes that gate with deviation 0.  For a link that is unitary up to accumulated rounding the
// change is of the order of that rounding.
__device__ __forceinline__ void reunitarize_m3(cd (&u)[9]) {
    double n0 = 0.0;
#pragma unroll
    for (int b = 0; b < 3; b++) n0 += u[b].re * u[b].re + u[b].im * u[b].im;
    const double i0 = 1.0 / sqrt(n0);
#pragma unroll
    for (int b = 0; b < 3; b++) u[b] = mk(i0 * u[b].re, i0 * u[b].im);
    cd d = mk(0.0, 0.0);      // <row0, row1>
#pragma unroll
    for (int b = 0; b < 3; b++) cfma_conj(d, u[b], u[3 + b]);
    double n1 = 0.0;
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const cd pr = cmul(d, u[b]);
        u[3 + b] = mk(u[3 + b].re - pr.re, u[3 + b].im - pr.im);
        n1 += u[3 + b].re * u[3 + b].re + u[3 + b].im * u[3 + b].im;
    }
    const double i1 = 1.0 / sqrt(n1);
#pragma unroll
    for (int b = 0; b < 3; b++) u[3 + b] = mk(i1 * u[3 + b].re, i1 * u[3 + b].im);
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const cd x = cmul(u[b1], u[3 + b2]) - cmul(u[b2], u[3 + b1]);
        u[6 + b] = mk(x.re, -x.im);
    }
}
// only a link that IS on the group up to accumulated rounding (deviation <= 1e-13) is put back onto it: the projection then moves it by
// about that rounding.  A configuration read from a text file (the reference's fixtures are unitary to 9e-11) is left exactly as the
// reference's literal update leaves it.
__device__ __forceinline__ void project_if_on_group(cd (&t)[9], unsigned* notproj) {
    cd v[9];
#pragma unroll
    for (int k = 0; k < 9; k++) v[k] = t[k];
    reunitarize_m3(v);
    double dev = 0.0;
#pragma unroll
    for (int k = 0; k < 9; k++) dev = fmax(dev, fmax(fabs(v[k].re - t[k].re), fabs(v[k].im - t[k].im)));
    if (dev <= 1e-13) {
#pragma unroll
        for (int k = 0; k < 9; k++) t[k] = v[k];
    } else {
        *notproj = 1u;      // some link of this field is not on the group (benign race: every writer stores the same value)
    }
}
template <bool REUNIT>
__global__ __launch_bounds__(256) void link_exp_update_kernel(Geom g, double2* __restrict__ U, double dt, const double2* __restrict__ P, unsigned* notproj) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd x[9], e[9], t[9], u[9];
    load_m3(x, P + off, Gs);
    exp_m3(e, x, dt);
    load_m3(u, U + off, Gs);
    mm3(t, e, u);
    if constexpr (REUNIT) project_if_on_group(t, notproj);
#pragma unroll
    for (int k = 0; k < 9; k++) st(U + off + (size_t)k * Gs, t[k]);
}
__global__ __launch_bounds__(256) void link_reunitarize_kernel(Geom g, double2* __restrict__ U) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd u[9];
    load_m3(u, U + off, Gs);
    reunitarize_m3(u);
#pragma unroll
    for (int k = 0; k < 9; k++) st(U + off + (size_t)k * Gs, u[k]);
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

// ---- single-direction forms: one 64-thread block per 64 sites of one parity, the direction slots are arguments.  They serve the
// reference's callers literally (U[mu], p[mu], one temporary link field at a time: AbstractMD.jl:78-135); the fused four-direction
// kernels above remain the fast path.
__device__ __forceinline__ bool site_of_thread(const Geom& g, int& p, int& i) {
    p = blockIdx.x & 1; i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    return i < g.Vh;
}
// op 0: C = t A (substitute_U! with t = 1)   op 1: C = exp(t A) (exptU!)   op 2: C = A B (mul!)   op 3: C += t * TA(A) (Traceless_antihermitian_add!)
template <int OP>
// C, A and B may be slots of one allocation, and C may be A or B itself (substitute_U!(U, U), mul!(temp1, U[mu], dSdUmu) on slots of one
// storage): no __restrict__ -- every thread loads all of its inputs before it stores, which is what makes the in-place forms well defined
__global__ __launch_bounds__(64) void link_op_kernel(Geom g, double2* C, int mc, const double2* A, int ma, const double2* B, int mb, double t, unsigned* notproj) {
    int p, i;
    if (!site_of_thread(g, p, i)) return;
    const int Gs = glink_stride(g);
    cd a[9], r[9];
    load_m3(a, A + glink_off(g, p, ma, i), Gs);
    double2* o = C + glink_off(g, p, mc, i);
    if constexpr (OP == 0) {
#pragma unroll
        for (int e = 0; e < 9; e++) r[e] = mk(t * a[e].re, t * a[e].im);
    } else if constexpr (OP == 1) {
        exp_m3(r, a, t);
    } else if constexpr (OP == 2) {
        cd b[9];
        load_m3(b, B + glink_off(g, p, mb, i), Gs);
        mm3(r, a, b);
    } else if constexpr (OP == 6) {                 // A^+ B: mul!(dSdU[mu], Uout[mu]', UdSfdU[mu]) (standardMD.jl:211)
        cd b[9];
        load_m3(b, B + glink_off(g, p, mb, i), Gs);
        mm3_dn(r, a, b);
    } else if constexpr (OP == 4 || OP == 5) {      // exp(t A) B: exptU! + mul! of the reference's U_update! in one pass (C may be B: the in-place link update)
        cd e[9], b[9];
        exp_m3(e, a, t);
        load_m3(b, B + glink_off(g, p, mb, i), Gs);
        mm3(r, e, b);
        if constexpr (OP == 5) {                   // the projection rule of link_exp_update_kernel<true>
            cd v[9];
#pragma unroll
            for (int k = 0; k < 9; k++) v[k] = r[k];
            reunitarize_m3(v);
            double dev = 0.0;
#pragma unroll
            for (int k = 0; k < 9; k++) dev = fmax(dev, fmax(fabs(v[k].re - r[k].re), fabs(v[k].im - r[k].im)));
            if (dev <= 1e-13) {
#pragma unroll
                for (int k = 0; k < 9; k++) r[k] = v[k];
            } else {
                *notproj = 1u;
            }
        }
    } else {
        cd h[9];
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int y = 0; y < 3; y++) h[x * 3 + y] = mk(0.5 * (a[x * 3 + y].re - a[y * 3 + x].re), 0.5 * (a[x * 3 + y].im + a[y * 3 + x].im));
        const double tr = (h[0].im + h[4].im + h[8].im) / 3.0;
        h[0].im -= tr; h[4].im -= tr; h[8].im -= tr;
#pragma unroll
        for (int e = 0; e < 9; e++) {
            const cd pv = ld(o + (size_t)e * Gs);
            r[e] = mk(fma(t, h[e].re, pv.re), fma(t, h[e].im, pv.im));
        }
    }
#pragma unroll
    for (int e = 0; e < 9; e++) st(o + (size_t)e * Gs, r[e]);
}

}  // namespace lqcd

using namespace lqcd;

static int same_ctx(lqcd_gauge_t a, lqcd_gauge_t b, const char* who) {
    if (!(a && b && a->ctx == b->ctx && a != b)) { set_error(std::string(who) + ": need two distinct gauge-shaped fields of one context"); return LQCD_ERR_ARG; }
    return LQCD_OK;
}

extern "C" int lqcd_gauge_copy(lqcd_gauge_t dst, lqcd_gauge_t src) {      // substitute_U!(Uold, U) (standardHMC.jl:45)
    LQCHK(lqcd::links_flush_of(dst));      // recorded single-direction link operations run first (md.hip)
    LQCHK(same_ctx(dst, src, "lqcd_gauge_copy"));
    lqcd_ctx_s* c = dst->ctx;
    HIPCHK(hipSetDevice(c->device));
    dst->version++;
    dst->unitary_version = src->unitary_version == src->version ? dst->version : 0;
    HIPCHK(hipMemcpyAsync(dst->data, src->data, src->elems * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// S_g = -(beta/3) sum_plaq Re tr U_p = -beta * 6 V_global * plaquette
extern "C" int lqcd_gauge_action(lqcd_gauge_t U, double beta, double* Sg) {
    LQCHK(lqcd::links_flush_of(U));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(U && Sg, "lqcd_gauge_action: null argument");
    double plaq = 0;
    LQCHK(lqcd_gauge_plaquette(U, &plaq));
    const lqcd_ctx_s* c = U->ctx;
    *Sg = -beta * 6.0 * (double)c->gL[0] * c->gL[1] * c->gL[2] * c->gL[3] * plaq;
    return LQCD_OK;
}

// ---- staple force on one rank / on a partitioned lattice
// Partitioned: (1) the x_lam = 0 link slices travel to the -lam neighbours (forward ghosts); (2) every rank computes the lower
// staples W_{mu nu} of its upper nu-faces (they need forward ghosts only) and sends them to the +nu neighbours; (3) the sweep
// reads ghosts for n+mu / n+nu and the received W for n-nu.  No corner exchange, two grouped send/recv steps.
static size_t gf_face_elems(lqcd_ctx_s* c, int mu) { return (size_t)2 * 4 * 9 * face_half_sites(c->geom, mu); }

int gf_buffers(lqcd_ctx_s* c) {
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu] || c->gf_ghost[mu]) continue;
        const size_t bytes = gf_face_elems(c, mu) * sizeof(double2);
        HIPCHK(hipMalloc((void**)&c->gf_ghost[mu], bytes));
        HIPCHK(hipMalloc((void**)&c->gf_gsend[mu], bytes));
        HIPCHK(hipMalloc((void**)&c->gf_wsend[mu], bytes));
        HIPCHK(hipMalloc((void**)&c->gf_wrecv[mu], bytes));
    }
    return LQCD_OK;
}

static GFArgs make_gfargs(lqcd_ctx_s* c, lqcd_gauge_s* U, lqcd_gauge_s* out, double beta, double factor) {
    GFArgs k;
    k.g = c->geom;
    k.U = U->data;
    k.out = out->data;
    k.coef = -beta / 6.0;
    k.factor = factor;
    k.mu_only = -1; k.mu_out = 0;
    k.uout = nullptr; k.dt = 0.0; k.notproj = nullptr; k.reunit = 0;
    k.bm = make_block_map(c->geom, c->tun.md_remap ? c->tun.xcd_remap : 0, c->tun.xcd_nsub, c->tun.xcd_ysplit);
    for (int mu = 0; mu < 4; mu++) { k.ghost[mu] = c->gf_ghost[mu]; k.wrecv[mu] = c->gf_wrecv[mu]; k.wsend[mu] = c->gf_wsend[mu]; }
    return k;
}

static int launch_staple_faces(lqcd_ctx_s* c, const GFArgs& k) {
    int maxf = 0;
    for (int mu = 0; mu < 4; mu++)
        if (c->geom.part[mu]) maxf = std::max(maxf, face_half_sites(c->geom, mu));
    if (!maxf) return LQCD_OK;
    hipLaunchKernelGGL(staple_face_kernel, dim3((2 * maxf + 127) / 128, 4), dim3(128), 0, c->stream, k);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
static int launch_staple_sweep(lqcd_ctx_s* c, const GFArgs& k, bool fuse, bool two_rows) {
    const dim3 grid(2 * c->geom.nch);
    if (any_partitioned(c)) {       // the instance with the ghost-link / received-staple branches
        if (k.mu_only >= 0 && fuse) hipLaunchKernelGGL(gauge_force_kernel_part<3>, grid, dim3(64), 0, c->stream, k);
        else if (k.mu_only >= 0) hipLaunchKernelGGL(gauge_force_kernel_part<2>, grid, dim3(64), 0, c->stream, k);
        else if (fuse) hipLaunchKernelGGL(gauge_force_kernel_part<1>, grid, dim3(256), 0, c->stream, k);
        else hipLaunchKernelGGL(gauge_force_kernel_part<0>, grid, dim3(256), 0, c->stream, k);
    } else {
        if (k.mu_only >= 0 && fuse) { if (two_rows) hipLaunchKernelGGL((gauge_force_kernel<3, false, true>), grid, dim3(64), 0, c->stream, k);
                                      else hipLaunchKernelGGL((gauge_force_kernel<3, false>), grid, dim3(64), 0, c->stream, k); }
        else if (k.mu_only >= 0) hipLaunchKernelGGL((gauge_force_kernel<2, false>), grid, dim3(64), 0, c->stream, k);
        else if (fuse) { if (two_rows && staple_tile_ok(c)) hipLaunchKernelGGL((gauge_force_kernel_tile<1, false>), grid, dim3(256), 0, c->stream, k);
                         else if (two_rows) hipLaunchKernelGGL((gauge_force_kernel<1, false, true>), grid, dim3(256), 0, c->stream, k);
                         else hipLaunchKernelGGL((gauge_force_kernel<1, false>), grid, dim3(256), 0, c->stream, k); }
        else { if (two_rows && staple_tile_ok(c)) hipLaunchKernelGGL((gauge_force_kernel_tile<0, false>), grid, dim3(256), 0, c->stream, k);
               else if (two_rows) hipLaunchKernelGGL((gauge_force_kernel<0, false, true>), grid, dim3(256), 0, c->stream, k);
               else hipLaunchKernelGGL((gauge_force_kernel<0, false>), grid, dim3(256), 0, c->stream, k); }
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// send `sendb[mu]` to one neighbour and receive into `recvb[mu]` from the opposite one, all partitioned directions in one group
int gf_exchange_rccl(lqcd_ctx_s* c, double2* const sendb[4], double2* const recvb[4], bool to_backward) {
    ARGCHK(c->has_comm, "staple force: communicator not initialised (call lqcd_ctx_comm_init or lqcd_ctx_peer_init)");
    CommXfer x[4];
    int n = 0;
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        x[n++] = CommXfer{sendb[mu], recvb[mu], gf_face_elems(c, mu) * sizeof(double2), mu, to_backward ? 1 : 0};
    }
    return comm_sendrecv(c, x, n, c->stream, false);
}

static int staple_force(lqcd_gauge_s* out, lqcd_gauge_s* U, double beta, double factor, bool fuse, int mu_only = -1, int mu_out = 0,
                        double coef_override = 0.0) {
    lqcd_ctx_s* c = U->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (any_partitioned(c)) {
        ARGCHK(c->local_peers.empty(), "staple force: this context belongs to an in-process PE grid, use lqcd_mdom_gauge_force (fuse = 1 adds it to the momenta)");
        LQCHK(gf_buffers(c));
        for (int mu = 0; mu < 4; mu++)
            if (c->geom.part[mu]) LQCHK(gauge_pack_face(U, mu, c->gf_gsend[mu]));
        LQCHK(gf_exchange_rccl(c, c->gf_gsend, c->gf_ghost, true));
    }
    out->version++;     // arguments are valid: the field is about to be written
    GFArgs k = make_gfargs(c, U, out, beta, factor);
    if (mu_only >= 0) { k.mu_only = mu_only; k.mu_out = mu_out; k.coef = coef_override; }
    if (any_partitioned(c)) {
        LQCHK(launch_staple_faces(c, k));
        LQCHK(gf_exchange_rccl(c, c->gf_wsend, c->gf_wrecv, false));
    }
    // links known to be on the group (tracked per version: generated there, measured, or projected by the link update): two rows are loaded
    LQCHK(launch_staple_sweep(c, k, fuse, c->tun.staple_recon && U->unitary_version == U->version));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// P_update! followed by U_update! (what every Sexton-Weingarten block of runMD_QPQ_sw! asks for, standardMD.jl:150-152) in ONE sweep over the links, single
// GPU: P += factor TA(-(beta/6) U staples), then U' = exp(dt P) U with the momentum still in registers -- U' goes to the context's spare link buffer (the sweep
// reads the old links of the neighbours until its last workgroup) and the two buffers change places in the handle.  Moves 576 (U) + 1152 (P r/w) + 576 (U')
// B/site where the two separate passes move 3456.
static int staple_force_expu(lqcd_gauge_s* P, lqcd_gauge_s* U, double beta, double factor, double dt) {
    lqcd_ctx_s* c = U->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!c->gauge_spare) {
        HIPCHK(hipMalloc((void**)&c->gauge_spare, U->elems * sizeof(double2)));
        HIPCHK(hipMemsetAsync(c->gauge_spare, 0, U->elems * sizeof(double2), c->stream));      // stride padding stays zero
    }
    GFArgs k = make_gfargs(c, U, P, beta, factor);
    unsigned* flag = c->pipe_ctr + PIPE_CTR_NOTPROJ_WORD;
    unsigned notproj = 1;
    k.uout = c->gauge_spare; k.dt = dt; k.notproj = flag; k.reunit = c->tun.md_reunitarize;
    const bool two_rows = c->tun.staple_recon && U->unitary_version == U->version;
    P->version++;
    if (k.reunit) HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
    const dim3 grid(2 * c->geom.nch);
    if (two_rows && staple_tile_ok(c)) hipLaunchKernelGGL((gauge_force_kernel_tile<1, true>), grid, dim3(256), 0, c->stream, k);
    else if (two_rows) hipLaunchKernelGGL((gauge_force_kernel<1, false, true, true>), grid, dim3(256), 0, c->stream, k);
    else hipLaunchKernelGGL((gauge_force_kernel<1, false, false, true>), grid, dim3(256), 0, c->stream, k);
    HIPCHK(hipGetLastError());
    if (k.reunit) HIPCHK(hipMemcpyAsync(&notproj, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    std::swap(U->data, c->gauge_spare);
    U->version++;
    if (k.reunit && !notproj) U->unitary_version = U->version;      // every link was projected: the field is on the group to rounding
    return LQCD_OK;
}

// ---- single-direction entry points (the interface the reference's unchanged callers use, AbstractMD.jl:78-135)
static int link_args(lqcd_gauge_t a, int ma, lqcd_gauge_t b, int mb, const char* who) {
    if (!(a && b && a->ctx == b->ctx && ma >= 0 && ma < 4 && mb >= 0 && mb < 4)) {
        set_error(std::string(who) + ": need gauge-shaped fields of one context and direction slots in 0..3");
        return LQCD_ERR_ARG;
    }
    return LQCD_OK;
}
template <int OP>
static int link_op(lqcd_gauge_t C, int mc, lqcd_gauge_t A, int ma, lqcd_gauge_t B, int mb, double t) {
    lqcd_ctx_s* c = C->ctx;
    HIPCHK(hipSetDevice(c->device));
    C->version++;
    hipLaunchKernelGGL(link_op_kernel<OP>, dim3(link_grid(c->geom)), dim3(64), 0, c->stream, c->geom, C->data, mc, A->data, ma,
                       B ? B->data : (const double2*)nullptr, mb, t, c->pipe_ctr + PIPE_CTR_NOTPROJ_WORD);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}
// ---- lazy link triples.  The reference's unchanged U_update! / P_update! (AbstractMD.jl:89-93, 107-111) update links and momenta one direction at a
// time through three generics each:
//     exptU!(expU, t, p[mu]);  mul!(W, expU, U[mu]);  substitute_U!(U[mu], W)                       -> lqcd_link_exp, lqcd_link_mul, lqcd_link_copy
//     calc_dSdUmu!(dSdUmu, ga, mu, U);  mul!(temp1, U[mu], dSdUmu);  Traceless_antihermitian_add!(p[mu], factor, temp1)
//                                                                                                  -> lqcd_link_staple, lqcd_link_mul, lqcd_link_add_ta
// The context RECORDS the first two calls of such a triple and launches ONE fused kernel at the third (link_exp_mul_now / link_add_ta_staple_now);
// completed triples are deferred once more, and when the same update has been asked for all four directions (what U_update! / P_update! do) the four
// become ONE launch of the fused four-direction kernel -- 1 launch per update instead of 12, callers and bindings unchanged (one ccall per generic).
// Every other entry point that reads, writes or destroys a gauge-shaped field (or applies an operator built on one) calls links_flush first, which
// runs what is recorded with the plain single-direction kernels in the order it was asked for: a temporary that IS read holds what the eager call
// would have put there.  The temporaries of a COMPLETED fused triple (expU, W / dSdUmu, temp1) are never written -- the reference's callers hand them
// back to their pool unread (unused!, AbstractMD.jl:95-97,113-117); a caller that does read them sets the tunable lazy_links = 0.
// In-process PE grids (lqcd_ctx_link_local, tests) run eagerly: their collectives are issued through lqcd_mdom_*.
static int link_exp_mul_now(lqcd_gauge_t W, int mu_w, double t, lqcd_gauge_t P, int mu_p, lqcd_gauge_t U, int mu_u);
static int gauge_exp_update_now(lqcd_gauge_t U, double dt, lqcd_gauge_t P);
static bool lazy_on(lqcd_ctx_s* c) { return c->tun.lazy_links && c->local_peers.empty(); }
static LinkRef lref(lqcd_gauge_s* g, int mu) { LinkRef r; r.g = g; r.mu = mu; return r; }

static int lazy_run_done(lqcd_ctx_s* c) {
    // waiting complete updates are older than every deferred triple; a momentum update is older than the link update behind it
    if (c->lazy.has_pp) {
        const LazyLinks::Done q = c->lazy.pp, r = c->lazy.pend;
        const bool both = c->lazy.has_pend && r.F == q.G && r.G == q.F && c->tun.lazy_merge > 1;
        c->lazy.has_pp = false;
        if (both) {
            c->lazy.has_pend = false;
            LQCHK(staple_force_expu(q.F, q.G, q.b, q.a, r.a));
        } else LQCHK(staple_force(q.F, q.G, q.b, q.a, true));
    }
    if (c->lazy.has_pend) {
        const LazyLinks::Done r = c->lazy.pend;
        c->lazy.has_pend = false;
        LQCHK(gauge_exp_update_now(r.F, r.a, r.G));
    }
    std::vector<LazyLinks::Done> d;
    d.swap(c->lazy.done);
    for (const LazyLinks::Done& r : d) {
        if (r.kind == 1) LQCHK(link_exp_mul_now(r.F, r.slot, r.a, r.G, r.slot, r.F, r.slot));
        else LQCHK(staple_force(r.F, r.G, r.b, r.a, true, r.slot, r.slot, 0.5 * r.b));
    }
    return LQCD_OK;
}
// a complete link update U <- exp(dt P) U: waits for a second one to merge with (tunable lazy_merge), or runs now
static int lazy_full_update(lqcd_ctx_s* c, lqcd_gauge_t U, double dt, lqcd_gauge_t P) {
    LazyLinks& z = c->lazy;
    if (!c->tun.lazy_merge) {
        if (z.has_pend || z.has_pp) LQCHK(lazy_run_done(c));
        return gauge_exp_update_now(U, dt, P);
    }
    if (z.has_pend && z.pend.F == U && z.pend.G == P) { z.pend.a += dt; return LQCD_OK; }
    if (z.has_pend || (z.has_pp && !(z.pp.F == P && z.pp.G == U))) LQCHK(lazy_run_done(c));
    const LazyLinks::Done d = {1, U, 0, dt, P, 0.0};
    z.pend = d;
    z.has_pend = true;
    return LQCD_OK;
}
// a complete momentum update P += factor TA(-(beta/6) U staples): everything that waits runs first (it reads the links); on one GPU it then waits itself
// for the link update that follows it (lazy_merge = 2: staple_force_expu)
static int lazy_full_pupdate(lqcd_ctx_s* c, lqcd_gauge_t P, double factor, lqcd_gauge_t U, double beta) {
    LazyLinks& z = c->lazy;
    if (z.has_pend || z.has_pp) LQCHK(lazy_run_done(c));
    if (c->tun.lazy_merge < 2 || any_partitioned(c) || P == U) return staple_force(P, U, beta, factor, true);
    const LazyLinks::Done d = {2, P, 0, factor, U, beta};
    z.pp = d;
    z.has_pp = true;
    return LQCD_OK;
}
// a completed triple: one of (up to) four of the same update, or run on its own
static int lazy_defer(lqcd_ctx_s* c, const LazyLinks::Done& r) {
    std::vector<LazyLinks::Done>& done = c->lazy.done;
    if (!done.empty()) {
        const LazyLinks::Done& d = done[0];
        bool clash = d.kind != r.kind || d.F != r.F || d.G != r.G || d.a != r.a || d.b != r.b;
        for (const LazyLinks::Done& e : done) clash = clash || e.slot == r.slot;
        if (clash) LQCHK(lazy_run_done(c));
    }
    {
        const LazyLinks& z = c->lazy;
        const bool wait_ok = r.kind == 1 && (!z.has_pend || (r.F == z.pend.F && r.G == z.pend.G)) && (!z.has_pp || (r.F == z.pp.G && r.G == z.pp.F));
        if ((z.has_pend || z.has_pp) && !wait_ok) LQCHK(lazy_run_done(c));
    }
    done.push_back(r);
    if (done.size() == 4) {
        done.clear();
        if (r.kind == 1) return lazy_full_update(c, r.F, r.a, r.G);
        return lazy_full_pupdate(c, r.F, -3.0 * r.a, r.G, r.b);      // factor TA(U (beta/2) staples) = (-3 factor) TA(-(beta/6) U staples)
    }
    return LQCD_OK;
}
// a new triple starts: an interrupted one runs first; deferred triples of the same kind stay deferred unless the new one writes one of their fields
static int lazy_open_triple(lqcd_ctx_s* c, int kind, const lqcd_gauge_s* tmp) {
    if (c->lazy.kind) return links_flush(c);
    bool run = (c->lazy.has_pend && (kind != 1 || c->lazy.pend.F == tmp || c->lazy.pend.G == tmp)) ||
               (c->lazy.has_pp && (kind != 1 || c->lazy.pp.F == tmp || c->lazy.pp.G == tmp));
    if (!c->lazy.done.empty()) {
        run = run || c->lazy.done[0].kind != kind;
        for (const LazyLinks::Done& e : c->lazy.done) run = run || e.F == tmp || e.G == tmp;
    }
    return run ? lazy_run_done(c) : LQCD_OK;
}
namespace lqcd {
int links_flush(lqcd_ctx_s* c) {
    if (c->lazy.has_pend || c->lazy.has_pp || !c->lazy.done.empty()) LQCHK(lazy_run_done(c));
    LazyLinks z = c->lazy;
    c->lazy.kind = 0;
    if (z.kind == 1 || z.kind == 2) {
        LQCHK(link_op<1>(z.E.g, z.E.mu, z.P.g, z.P.mu, nullptr, 0, z.t));
        if (z.kind == 2) LQCHK(link_op<2>(z.W.g, z.W.mu, z.E.g, z.E.mu, z.U.g, z.U.mu, 0.0));
    } else if (z.kind == 3 || z.kind == 4) {
        LQCHK(staple_force(z.S.g, z.Ug, z.beta, 0.0, false, z.mu, z.S.mu, 0.5 * z.beta));
        if (z.kind == 4) LQCHK(link_op<2>(z.T.g, z.T.mu, z.Ug, z.mu, z.S.g, z.S.mu, 0.0));
    }
    return LQCD_OK;
}
}  // namespace lqcd

// substitute_U!(U[mu], W) (AbstractMD.jl:93): one direction of dst <- one direction of src (the same field is allowed).  Third call of the
// U_update! triple: U[mu] <- exp(t p[mu]) U[mu] in one pass
extern "C" int lqcd_link_copy(lqcd_gauge_t dst, int mu_dst, lqcd_gauge_t src, int mu_src) {
    LQCHK(link_args(dst, mu_dst, src, mu_src, "lqcd_link_copy"));
    lqcd_ctx_s* c = dst->ctx;
    LazyLinks& z = c->lazy;
    if (z.kind == 2 && z.W.is(src, mu_src) && z.U.is(dst, mu_dst) && z.P.g != dst) {
        const LazyLinks r = z;
        z.kind = 0;
        if (r.P.mu == mu_dst) {      // p[mu] with U[mu]: maybe one of four
            LazyLinks::Done d = {1, dst, mu_dst, r.t, r.P.g, 0.0};
            return lazy_defer(c, d);
        }
        if (c->lazy.has_pend || c->lazy.has_pp) LQCHK(lazy_run_done(c));
        return link_exp_mul_now(dst, mu_dst, r.t, r.P.g, r.P.mu, dst, mu_dst);
    }
    LQCHK(links_flush_of(c));
    if (dst == src && mu_dst == mu_src) return LQCD_OK;
    return link_op<0>(dst, mu_dst, src, mu_src, nullptr, 0, 1.0);
}
// dst[mu_dst] = s * src[mu_src]: hands one direction of a force field to the reference's caller in ITS sign convention
// (calc_UdSfdU! fills "U dS_f/dU" = -G, P_update_fermion! adds factor = -eps dtau times its TA part: AbstractMD.jl:127-132)
extern "C" int lqcd_link_scaled_copy(lqcd_gauge_t dst, int mu_dst, double s, lqcd_gauge_t src, int mu_src) {
    LQCHK(link_args(dst, mu_dst, src, mu_src, "lqcd_link_scaled_copy"));
    LQCHK(links_flush_of(dst));
    return link_op<0>(dst, mu_dst, src, mu_src, nullptr, 0, s);
}
// exptU!(expU, t, p[mu], temps) (AbstractMD.jl:91): E[mu_e] = exp(t P[mu_p]), the Taylor-Horner series of lqcd_gauge_exp_update.  First call of the
// U_update! triple: recorded
extern "C" int lqcd_link_exp(lqcd_gauge_t E, int mu_e, double t, lqcd_gauge_t P, int mu_p) {
    LQCHK(link_args(E, mu_e, P, mu_p, "lqcd_link_exp"));
    lqcd_ctx_s* c = E->ctx;
    if (lazy_on(c) && E != P) {
        LQCHK(lazy_open_triple(c, 1, E));      // p[mu] is only read, by this triple and by the deferred ones
        LazyLinks& z = c->lazy;
        z.kind = 1; z.E = lref(E, mu_e); z.P = lref(P, mu_p); z.t = t;
        return LQCD_OK;
    }
    LQCHK(links_flush_of(c));
    return link_op<1>(E, mu_e, P, mu_p, nullptr, 0, t);
}
// mul!(W, expU, U[mu]) / mul!(temp1, U[mu], dSdUmu) (AbstractMD.jl:92,109): C[mu_c](n) = A[mu_a](n) B[mu_b](n), site by site.  Second call of
// either triple: recorded
extern "C" int lqcd_link_mul(lqcd_gauge_t C, int mu_c, lqcd_gauge_t A, int mu_a, lqcd_gauge_t B, int mu_b) {
    LQCHK(link_args(C, mu_c, A, mu_a, "lqcd_link_mul"));
    LQCHK(link_args(C, mu_c, B, mu_b, "lqcd_link_mul"));
    lqcd_ctx_s* c = C->ctx;
    LazyLinks& z = c->lazy;
    if (z.kind == 1 && z.E.is(A, mu_a) && !z.E.is(C, mu_c) && !z.P.is(C, mu_c)) {
        z.kind = 2; z.W = lref(C, mu_c); z.U = lref(B, mu_b);
        return LQCD_OK;
    }
    if (z.kind == 3 && z.S.is(B, mu_b) && A == z.Ug && mu_a == z.mu && !z.S.is(C, mu_c) && C != z.Ug) {
        z.kind = 4; z.T = lref(C, mu_c);
        return LQCD_OK;
    }
    LQCHK(links_flush_of(c));
    return link_op<2>(C, mu_c, A, mu_a, B, mu_b, 0.0);
}
// mul!(C, A', B) on link fields (standardMD.jl:211: mul!(md.dSdU[mu], Uout[mu]', UdSfdUmu[mu])): C[mu_c](n) = A[mu_a](n)^+ B[mu_b](n)
extern "C" int lqcd_link_mul_adj(lqcd_gauge_t C, int mu_c, lqcd_gauge_t A, int mu_a, lqcd_gauge_t B, int mu_b) {
    LQCHK(link_args(C, mu_c, A, mu_a, "lqcd_link_mul_adj"));
    LQCHK(link_args(C, mu_c, B, mu_b, "lqcd_link_mul_adj"));
    LQCHK(links_flush_of(C));
    return link_op<6>(C, mu_c, A, mu_a, B, mu_b, 0.0);
}
// Traceless_antihermitian_add!(p[mu], factor, temp1) (AbstractMD.jl:110,131): P[mu_p] += factor * TA(G[mu_g]).  Third call of the P_update! triple:
// p[mu] += factor TA(U[mu] (beta/2) staples) in one pass
extern "C" int lqcd_link_add_ta(lqcd_gauge_t P, int mu_p, double factor, lqcd_gauge_t G, int mu_g) {
    LQCHK(link_args(P, mu_p, G, mu_g, "lqcd_link_add_ta"));
    ARGCHK(!(P == G && mu_p == mu_g), "lqcd_link_add_ta: P and G are the same link field");
    lqcd_ctx_s* c = P->ctx;
    LazyLinks& z = c->lazy;
    if (z.kind == 4 && z.T.is(G, mu_g) && P != z.Ug && P != z.T.g && P != z.S.g) {
        const LazyLinks r = z;
        z.kind = 0;
        if (mu_p == r.mu) {
            LazyLinks::Done d = {2, P, mu_p, factor, r.Ug, r.beta};
            return lazy_defer(c, d);
        }
        if (c->lazy.has_pend || c->lazy.has_pp) LQCHK(lazy_run_done(c));
        return staple_force(P, r.Ug, r.beta, factor, true, r.mu, mu_p, 0.5 * r.beta);
    }
    LQCHK(links_flush_of(c));
    return link_op<3>(P, mu_p, G, mu_g, nullptr, 0, factor);
}
// calc_dSdUmu!(dSdUmu, gauge_action, mu, U) (AbstractMD.jl:108) for the plaquette action pushed with coefficient beta/2
// (universe.jl:92-95): out[mu_out](n) = (beta/2) * sum of the six staples of U_mu(n), so that U_mu(n) out(n) is the plaquette
// sum whose -1/NC-weighted traceless anti-Hermitian part P_update! adds to p[mu].  Collective on a partitioned lattice.  First call of the
// P_update! triple: recorded
extern "C" int lqcd_link_staple(lqcd_gauge_t out, int mu_out, lqcd_gauge_t U, int mu, double beta) {
    LQCHK(link_args(out, mu_out, U, mu, "lqcd_link_staple"));
    ARGCHK(out != U, "lqcd_link_staple: out must not be the link field itself");
    lqcd_ctx_s* c = out->ctx;
    if (lazy_on(c)) {
        LQCHK(lazy_open_triple(c, 2, out));
        LazyLinks& z = c->lazy;
        z.kind = 3; z.S = lref(out, mu_out); z.Ug = U; z.mu = mu; z.beta = beta;
        return LQCD_OK;
    }
    LQCHK(links_flush_of(c));
    return staple_force(out, U, beta, 0.0, false, mu, mu_out, 0.5 * beta);
}

// The three per-direction calls of the reference's U_update! (AbstractMD.jl:91-93) -- exptU!(expU, t, p[mu]); mul!(W, expU, U[mu]);
// substitute_U!(U[mu], W) -- as ONE pass: W[mu_w] = exp(t P[mu_p]) U[mu_u], W = U allowed (the in-place update of one direction).  Reached by
// the lazy triples above, or directly.  In place and with the tunable
// md_reunitarize the updated links are projected back onto SU(3) under the rule of lqcd_gauge_exp_update (the field stays "on the group" if it was).
static int link_exp_mul_now(lqcd_gauge_t W, int mu_w, double t, lqcd_gauge_t P, int mu_p, lqcd_gauge_t U, int mu_u) {
    lqcd_ctx_s* c = W->ctx;
    const bool inplace = W == U && mu_w == mu_u;
    if (!(inplace && c->tun.md_reunitarize)) return link_op<4>(W, mu_w, P, mu_p, U, mu_u, t);
    HIPCHK(hipSetDevice(c->device));
    const bool was_on_group = U->unitary_version == U->version;
    unsigned* flag = c->pipe_ctr + PIPE_CTR_NOTPROJ_WORD;
    unsigned notproj = 1;
    HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
    U->version++;
    hipLaunchKernelGGL(link_op_kernel<5>, dim3(link_grid(c->geom)), dim3(64), 0, c->stream, c->geom, U->data, mu_w, P->data, mu_p, U->data, mu_u, t, flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&notproj, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (was_on_group && !notproj) U->unitary_version = U->version;      // the other three directions were on the group, this one was projected
    return LQCD_OK;
}
extern "C" int lqcd_link_exp_mul(lqcd_gauge_t W, int mu_w, double t, lqcd_gauge_t P, int mu_p, lqcd_gauge_t U, int mu_u) {
    LQCHK(link_args(W, mu_w, P, mu_p, "lqcd_link_exp_mul"));
    LQCHK(link_args(W, mu_w, U, mu_u, "lqcd_link_exp_mul"));
    ARGCHK(W != P && U != P, "lqcd_link_exp_mul: the momentum field must be a field of its own");
    LQCHK(links_flush_of(W));
    return link_exp_mul_now(W, mu_w, t, P, mu_p, U, mu_u);
}

// The three per-direction calls of the reference's P_update! (AbstractMD.jl:108-110) -- calc_dSdUmu!(dSdUmu, gauge_action, mu, U);
// mul!(temp1, U[mu], dSdUmu); Traceless_antihermitian_add!(p[mu], factor, temp1) -- as ONE pass: P[mu_p] += factor * TA(U[mu] * (beta/2) * staples);
// reached by the lazy triples above, or directly
extern "C" int lqcd_link_add_ta_staple(lqcd_gauge_t P, int mu_p, double factor, lqcd_gauge_t U, int mu, double beta) {
    LQCHK(link_args(P, mu_p, U, mu, "lqcd_link_add_ta_staple"));
    ARGCHK(P != U, "lqcd_link_add_ta_staple: the momentum field must not be the link field itself");
    LQCHK(links_flush_of(P));
    return staple_force(P, U, beta, factor, true, mu, mu_p, 0.5 * beta);
}

// G_mu(n) = -(beta/6) U_mu(n) * (sum of the six staples)      (calc_dSdUmu! + mul!(temp, U, dSdUmu), AbstractMD.jl:108-110)
extern "C" int lqcd_gauge_force(lqcd_gauge_t out, lqcd_gauge_t U, double beta) {
    LQCHK(lqcd::links_flush_of(out));      // recorded single-direction link operations run first (md.hip)
    LQCHK(same_ctx(out, U, "lqcd_gauge_force"));
    return staple_force(out, U, beta, 0.0, false);
}

// P_update!(U, p, eps, md) (AbstractMD.jl:99-118) in one pass:  P += factor * TA(-(beta/6) U * staples); the force field is never stored
extern "C" int lqcd_momentum_add_gauge_force(lqcd_gauge_t P, double factor, lqcd_gauge_t U, double beta) {
    LQCHK(same_ctx(P, U, "lqcd_momentum_add_gauge_force"));
    lqcd_ctx_s* c = P->ctx;
    if (lazy_on(c) && c->tun.lazy_merge > 1) {      // waits for the link update that follows it (lazy_full_pupdate)
        LQCHK(links_flush(c));
        return lazy_full_pupdate(c, P, factor, U, beta);
    }
    LQCHK(lqcd::links_flush_of(P));      // recorded single-direction link operations run first
    return staple_force(P, U, beta, factor, true);
}

// the same on an in-process PE grid (tests): arrays ordered by rank; fuse = 0 writes the force field, 1 accumulates into momenta
extern "C" int lqcd_mdom_gauge_force(int n, lqcd_gauge_t* outs, lqcd_gauge_t* Us, double beta, double factor, int fuse) {
    ARGCHK(outs && Us && n >= 1, "lqcd_mdom_gauge_force: null");
    lqcd_ctx_s* c0 = Us[0]->ctx;
    ARGCHK((int)c0->local_peers.size() == n, "lqcd_mdom_gauge_force: contexts are not linked with lqcd_ctx_link_local (or wrong n)");
    std::vector<GFArgs> ks(n);
    for (int r = 0; r < n; r++) {
        LQCHK(same_ctx(outs[r], Us[r], "lqcd_mdom_gauge_force"));
        lqcd_ctx_s* c = Us[r]->ctx;
        ARGCHK(c->rank == r, "lqcd_mdom_gauge_force: fields must be ordered by rank");
        HIPCHK(hipSetDevice(c->device));
        LQCHK(gf_buffers(c));
        outs[r]->version++;
    }
    for (int r = 0; r < n; r++) {          // forward ghosts: the x_mu = 0 slice of the +mu neighbour
        lqcd_ctx_s* c = Us[r]->ctx;
        for (int mu = 0; mu < 4; mu++)
            if (c->geom.part[mu]) LQCHK(gauge_pack_face(Us[c->nbr_fwd[mu]], mu, c->gf_ghost[mu]));
    }
    HIPCHK(hipDeviceSynchronize());
    for (int r = 0; r < n; r++) {
        ks[r] = make_gfargs(Us[r]->ctx, Us[r], outs[r], beta, factor);
        LQCHK(launch_staple_faces(Us[r]->ctx, ks[r]));
    }
    HIPCHK(hipDeviceSynchronize());
    for (int r = 0; r < n; r++) {          // lower staples of the upper nu-face -> the +nu neighbour
        lqcd_ctx_s* c = Us[r]->ctx;
        for (int mu = 0; mu < 4; mu++)
            if (c->geom.part[mu])
                HIPCHK(hipMemcpy(Us[c->nbr_fwd[mu]]->ctx->gf_wrecv[mu], c->gf_wsend[mu], gf_face_elems(c, mu) * sizeof(double2), hipMemcpyDeviceToDevice));
    }
    HIPCHK(hipDeviceSynchronize());
    for (int r = 0; r < n; r++) LQCHK(launch_staple_sweep(Us[r]->ctx, ks[r], fuse != 0, false));
    HIPCHK(hipDeviceSynchronize());
    return LQCD_OK;
}

// Traceless_antihermitian_add!(p, factor, G) (AbstractMD.jl:110,131):  P += factor * TA(G)
extern "C" int lqcd_momentum_add_ta(lqcd_gauge_t P, double factor, lqcd_gauge_t G) {
    LQCHK(lqcd::links_flush_of(P));      // recorded single-direction link operations run first (md.hip)
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
    if (lazy_on(c) && c->tun.lazy_merge && U != P) {      // waits for a second update of the same fields to merge with (lazy_full_update)
        const LazyLinks& z = c->lazy;
        if (z.kind || !z.done.empty()) LQCHK(links_flush(c));
        return lazy_full_update(c, U, dt, P);
    }
    LQCHK(links_flush_of(U));
    return gauge_exp_update_now(U, dt, P);
}
static int gauge_exp_update_now(lqcd_gauge_t U, double dt, lqcd_gauge_t P) {
    lqcd_ctx_s* c = U->ctx;
    HIPCHK(hipSetDevice(c->device));
    U->version++;
    // md_reunitarize (default): an updated link that is unitary up to accumulated rounding (1e-13) is projected back onto SU(3) in the same
    // pass (links of a configuration that was never on the group to that precision are left alone).  exp(dt P) U leaves the group only by
    // rounding, but that rounding accumulates: max |row2 - conj(row0 x row1)| passes 1e-14 after ~280 updates (profiles/r03_unitarity_drift.log),
    // i.e. inside the FIRST trajectory, and the 12-real Dslash would be lost for the rest of the run.  0 = the reference's literal U_update!.
    unsigned* flag = c->pipe_ctr + PIPE_CTR_NOTPROJ_WORD;      // a spare word of the counter block
    unsigned notproj = 1;
    if (c->tun.md_reunitarize) {
        HIPCHK(hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
        hipLaunchKernelGGL(link_exp_update_kernel<true>, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, U->data, dt, P->data, flag);
        HIPCHK(hipMemcpyAsync(&notproj, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    } else hipLaunchKernelGGL(link_exp_update_kernel<false>, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, U->data, dt, P->data, flag);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!notproj) U->unitary_version = U->version;      // every link was projected: the field is on the group to rounding
    return LQCD_OK;
}

// every link back onto SU(3) (Gram-Schmidt of rows 0, 1; row 2 = conj(row 0 x row 1)): for callers that update links through the
// single-direction entry points (the reference's own U_update!), once per trajectory keeps the 12-real Dslash path alive
extern "C" int lqcd_gauge_reunitarize(lqcd_gauge_t U) {
    LQCHK(lqcd::links_flush_of(U));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(U, "lqcd_gauge_reunitarize: null argument");
    lqcd_ctx_s* c = U->ctx;
    HIPCHK(hipSetDevice(c->device));
    U->version++;
    U->unitary_version = U->version;
    hipLaunchKernelGGL(link_reunitarize_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, U->data);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// gauss_distribution!(p) (standardMD.jl:86)
extern "C" int lqcd_momentum_gaussian(lqcd_gauge_t P, uint64_t seed) {
    LQCHK(lqcd::links_flush_of(P));      // recorded single-direction link operations run first (md.hip)
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
    LQCHK(lqcd::links_flush_of(P));      // recorded single-direction link operations run first (md.hip)
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

// ---------------------------------------------------------------------------------- stout smearing and its back-propagation
// The CovNeuralnet of the reference's fermion force (src/system/universe.jl:147-171: STOUT_Layer(p.stout_loops, p.stout_ρ, U); src/md/standardMD.jl:192-227:
// calc_smearedU, calc_UdSfdU! on the smeared links, back_prop; src/updates/standardHMC.jl:67-68).  The arithmetic is Gaugefields.jl's, which is not under the
// reference tree: this is Morningstar-Peardon's definition for the plaquette loop [EXT-RECALL, parity unpinned; the CPU restatement in the test
// infrastructure is itself checked by finite differences, tests/test_cpu_stout_restatement.py],
//     U'_mu(n) = exp(Z) U_mu(n),   Z = i Q = -rho TA(W),   W = U_mu(n) A_mu(n) (A: the six staples -- the staple sweep above with beta = -6),
// and the chain rule with the Frechet derivative L(Z, .) of exp in place of the closed-form B matrices (the same linear map, no special cases):
//     G_i = e^{-Z_i} G'_i e^{Z_i} + force of S~ = -2 rho sum_j Re tr(W_j N_j),   N_j = TA(L(Z_j, e^{-Z_j} G'_j)) held fixed,
// where G' = "U' dS/dU'" at the smeared links and G = "U dS/dU" at the thin ones, both in the convention of lqcd_fermion_force.  S~ puts N_j at the start of each
// of the 24 plaquette loops through a link: 6 plaquettes x the 4 links whose staple sums contain them (stout_gather_kernel; on a partitioned lattice
// stout_gather_ext_kernel of clover.hip, which reads links and N matrices from the halo-extended block).
namespace lqcd {

__device__ __forceinline__ void dag3(cd (&o)[9], const cd (&a)[9]) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) o[r * 3 + q] = mk(a[q * 3 + r].re, -a[q * 3 + r].im);
}
__device__ __forceinline__ void add3(cd (&o)[9], const cd (&a)[9]) {
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = o[k] + a[k];
}
__device__ __forceinline__ void sub3(cd (&o)[9], const cd (&a)[9]) {
#pragma unroll
    for (int k = 0; k < 9; k++) o[k] = o[k] - a[k];
}
// x = s TA(w),  TA(w) = (w - w^+)/2 - tr(w - w^+)/6
__device__ __forceinline__ void ta3(cd (&x)[9], const cd (&w)[9], double s) {
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int q = 0; q < 3; q++) x[r * 3 + q] = mk(0.5 * s * (w[r * 3 + q].re - w[q * 3 + r].re), 0.5 * s * (w[r * 3 + q].im + w[q * 3 + r].im));
    const double tr = (x[0].im + x[4].im + x[8].im) / 3.0;
    x[0].im -= tr; x[4].im -= tr; x[8].im -= tr;
}
__device__ __forceinline__ double rowsum_norm(const cd (&x)[9]) {
    double nrm = 0.0;
#pragma unroll
    for (int a = 0; a < 3; a++)
        nrm = fmax(nrm, (fabs(x[a * 3].re) + fabs(x[a * 3].im)) + (fabs(x[a * 3 + 1].re) + fabs(x[a * 3 + 1].im)) + (fabs(x[a * 3 + 2].re) + fabs(x[a * 3 + 2].im)));
    return nrm;
}
// series length for exp and its Frechet derivative at norm nrm: n nrm^n / n! < 1e-18 (rho |TA(W)| of a stout layer is a few tenths; 3 is far outside)
__device__ __forceinline__ int stout_terms(double nrm) { return nrm < 0.2 ? 14 : nrm < 0.5 ? 18 : nrm < 1.0 ? 23 : nrm < 2.0 ? 31 : 42; }
// e = exp(x), Taylor-Horner
__device__ __forceinline__ void exp_any3(cd (&e)[9], const cd (&x)[9]) {
    cd t[9];
#pragma unroll
    for (int k = 0; k < 9; k++) e[k] = mk((k % 4 == 0) ? 1.0 : 0.0, 0.0);
    for (int n = stout_terms(rowsum_norm(x)); n >= 1; n--) {
        mm3(t, x, e);
        const double inv = 1.0 / (double)n;
#pragma unroll
        for (int k = 0; k < 9; k++) e[k] = mk(((k % 4 == 0) ? 1.0 : 0.0) + inv * t[k].re, inv * t[k].im);
    }
}
// l = L(z, k) = sum_n D_n,  P_0 = 1, D_0 = 0,  D_n = (z D_{n-1} + k P_{n-1}) / n,  P_n = z P_{n-1} / n
__device__ __forceinline__ void frechet3(cd (&l)[9], const cd (&z)[9], const cd (&k)[9]) {
    cd P[9], D[9], t1[9], t2[9];
#pragma unroll
    for (int e = 0; e < 9; e++) { P[e] = mk((e % 4 == 0) ? 1.0 : 0.0, 0.0); D[e] = mk(0.0, 0.0); l[e] = mk(0.0, 0.0); }
    const int nt = stout_terms(rowsum_norm(z));
    for (int n = 1; n <= nt; n++) {
        const double inv = 1.0 / (double)n;
        mm3(t1, z, D);
        mm3(t2, k, P);
#pragma unroll
        for (int e = 0; e < 9; e++) D[e] = mk(inv * (t1[e].re + t2[e].re), inv * (t1[e].im + t2[e].im));
        mm3(t1, z, P);
#pragma unroll
        for (int e = 0; e < 9; e++) { P[e] = mk(inv * t1[e].re, inv * t1[e].im); l[e] = l[e] + D[e]; }
    }
}
__device__ __forceinline__ void store_m3(double2* base, int stride, const cd (&a)[9]) {
#pragma unroll
    for (int e = 0; e < 9; e++) st(base + (size_t)e * stride, a[e]);
}

// out = exp(-rho TA(W)) U
__global__ __launch_bounds__(256) void stout_smear_kernel(Geom g, double2* __restrict__ out, const double2* __restrict__ U, const double2* __restrict__ W, double rho) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd w[9], z[9], e[9], u[9], r[9];
    load_m3(w, W + off, Gs);
    ta3(z, w, -rho);
    exp_any3(e, z);
    load_m3(u, U + off, Gs);
    mm3(r, e, u);
    store_m3(out + off, Gs, r);
}
// per link: Z = -rho TA(W); K = e^{-Z} G'; N = TA(L(Z, K)); G0 = K e^{Z}  (G0 may be written over G')
__global__ __launch_bounds__(256) void stout_prep_kernel(Geom g, double2* __restrict__ N, double2* G0, const double2* Gp, const double2* __restrict__ W, double rho, int lam_layout) {
    size_t off;
    if (!link_of_thread(g, off)) return;
    const int Gs = glink_stride(g);
    cd w[9], z[9], zm[9], em[9], gp[9], K[9], l[9], n[9], ep[9], g0[9];
    load_m3(w, W + off, Gs);
    ta3(z, w, -rho);
#pragma unroll
    for (int e = 0; e < 9; e++) zm[e] = mk(-z[e].re, -z[e].im);
    exp_any3(em, zm);
    load_m3(gp, Gp + off, Gs);
    mm3(K, em, gp);
    frechet3(l, z, K);
    ta3(n, l, 1.0);
    if (lam_layout) {      // partitioned lattice: plane mu of the Lambda-shaped buffer the halo-extended block is filled from ([parity][chunk][6][9][64])
        const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
        store_m3(N + ((((size_t)p * g.nch + (size_t)(i >> 6)) * 6 + mu) * 9) * 64 + (i & 63), 64, n);
    } else store_m3(N + off, Gs, n);
    dag3(ep, em);                  // Z anti-Hermitian: e^{Z} = (e^{-Z})^+
    mm3(g0, K, ep);
    store_m3(G0 + off, Gs, g0);
}
// G_mu(n) += -rho * (the 24 loop terms through the link): thread = (site, mu)
__global__ __launch_bounds__(256) void stout_gather_kernel(Geom g, double2* __restrict__ G, const double2* __restrict__ U, const double2* __restrict__ N, double rho) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    const int Gs = glink_stride(g);
    int c[4];
    cb_to_coords(g, p, i, c);
    cd a[9], Na[9], acc[9];
    load_m3(a, link_at(g, U, c, mu), Gs);
    load_m3(Na, link_at(g, N, c, mu), Gs);
#pragma unroll
    for (int e = 0; e < 9; e++) acc[e] = mk(0.0, 0.0);
    for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        cd b[9], cc[9], d[9], Nb[9], Nc[9], Nd[9], t1[9], t2[9], t3[9], X[9];
        int cm[4] = {c[0], c[1], c[2], c[3]}, cn[4] = {c[0], c[1], c[2], c[3]};
        shift(cm, g, mu, 1);
        shift(cn, g, nu, 1);
        // the plaquette (n; mu, nu), this link as a: b = U_nu(n+mu), c = U_mu(n+nu), d = U_nu(n)
        load_m3(b, link_at(g, U, cm, nu), Gs); load_m3(cc, link_at(g, U, cn, mu), Gs); load_m3(d, link_at(g, U, c, nu), Gs);
        load_m3(Nb, link_at(g, N, cm, nu), Gs); load_m3(Nc, link_at(g, N, cn, mu), Gs); load_m3(Nd, link_at(g, N, c, nu), Gs);
        mm3_nd(t1, b, cc);          // b c^+
        mm3_nd(X, t1, d);           // X = b c^+ d^+
        mm3(t2, X, Na);
        mm3(t3, Nb, X);
        add3(t2, t3);
        mm3(t3, a, t2);             // a (X Na + Nb X)
        add3(acc, t3);
        mm3(t2, a, X);
        dag3(t3, t2);               // (a X)^+ = d c b^+ a^+
        mm3(t2, Nd, t3);
        sub3(acc, t2);              // - Nd d c b^+ a^+
        mm3(t2, a, t1);
        dag3(t3, t2);               // (a b c^+)^+ = c b^+ a^+
        mm3(t2, Nc, t3);
        mm3(t3, d, t2);
        sub3(acc, t3);              // - d Nc c b^+ a^+
        // the plaquette (n - nu; mu, nu), this link as c: a2 = U_mu(m), b2 = U_nu(m + mu), d2 = U_nu(m), m = n - nu
        int m[4] = {c[0], c[1], c[2], c[3]};
        shift(m, g, nu, -1);
        int mm[4] = {m[0], m[1], m[2], m[3]};
        shift(mm, g, mu, 1);
        load_m3(b, link_at(g, U, mm, nu), Gs); load_m3(cc, link_at(g, U, m, mu), Gs); load_m3(d, link_at(g, U, m, nu), Gs);       // b2, a2, d2
        load_m3(Nb, link_at(g, N, mm, nu), Gs); load_m3(Nc, link_at(g, N, m, mu), Gs); load_m3(Nd, link_at(g, N, m, nu), Gs);     // Nb2, Na2, Nd2
        mm3(t1, cc, b);             // a2 b2
        dag3(X, t1);                // R = b2^+ a2^+
        mm3(t1, Nd, d);
        mm3(t2, d, Na);             // Nc of that plaquette is this link's own N
        add3(t1, t2);               // Nd2 d2 + d2 Nc
        mm3(t2, X, t1);
        mm3(t3, a, t2);             // c R (Nd2 d2 + d2 Nc)
        add3(acc, t3);
        mm3(t1, Nc, cc);            // Na2 a2
        mm3(t2, cc, Nb);            // a2 Nb2
        add3(t1, t2);
        mm3(t2, t1, b);             // (Na2 a2 + a2 Nb2) b2
        mm3_nd(t1, t2, a);          // ... c^+
        dag3(t3, d);
        mm3(t2, t3, t1);            // d2^+ (...)
        sub3(acc, t2);
    }
    double2* o = G + glink_off(g, p, mu, i);
#pragma unroll
    for (int e = 0; e < 9; e++) {
        const cd v = ld(o + (size_t)e * Gs);
        st(o + (size_t)e * Gs, mk(v.re - rho * acc[e].re, v.im - rho * acc[e].im));      // 0.5 * c0 = -rho
    }
}

static int stout_tmp(lqcd_ctx_s* c, int i, lqcd_gauge_s** out) {
    if (!c->stout_tmp[i]) LQCHK(lqcd_gauge_create(c, &c->stout_tmp[i]));
    *out = c->stout_tmp[i];
    return LQCD_OK;
}

}  // namespace lqcd

// calc_smearedU(U, nn) for one STOUT_Layer(["plaquette"], [rho], U) (standardMD.jl:207, universe.jl:150-154): out = the smeared links; out must not be U
extern "C" int lqcd_stout_smear(lqcd_gauge_t out, lqcd_gauge_t U, double rho) {
    LQCHK(same_ctx(out, U, "lqcd_stout_smear"));
    ARGCHK(out != U, "lqcd_stout_smear: the smeared links need a field of their own");
    LQCHK(lqcd::links_flush_of(out));
    lqcd_ctx_s* c = U->ctx;
    ARGCHK(c->local_peers.empty(), "lqcd_stout_smear: not available on an in-process PE grid (RCCL ranks only)");
    HIPCHK(hipSetDevice(c->device));
    lqcd_gauge_s* W;
    LQCHK(stout_tmp(c, 0, &W));
    LQCHK(staple_force(W, U, -6.0, 0.0, false));      // W = U A
    out->version++;
    hipLaunchKernelGGL(stout_smear_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, out->data, U->data, W->data, rho);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// back_prop(dSdU, nn, Uout_multi, U) for that layer (standardMD.jl:216): G (thin links) from Gs = "U' dS/dU'" (smeared links), the convention of
// lqcd_fermion_force on both sides; U = the thin links the layer smeared.  G = Gs is allowed (in place)
extern "C" int lqcd_stout_backprop(lqcd_gauge_t G, lqcd_gauge_t Gs, lqcd_gauge_t U, double rho) {
    LQCHK(same_ctx(G, U, "lqcd_stout_backprop"));
    LQCHK(same_ctx(Gs, U, "lqcd_stout_backprop"));
    ARGCHK(G != U && Gs != U, "lqcd_stout_backprop: the force fields must not be the link field");
    LQCHK(lqcd::links_flush_of(G));
    lqcd_ctx_s* c = U->ctx;
    ARGCHK(c->local_peers.empty(), "lqcd_stout_backprop: not available on an in-process PE grid (RCCL ranks only)");
    HIPCHK(hipSetDevice(c->device));
    lqcd_gauge_s *W, *N;
    LQCHK(stout_tmp(c, 0, &W)); LQCHK(stout_tmp(c, 1, &N));
    LQCHK(staple_force(W, U, -6.0, 0.0, false));
    G->version++;
    if (any_partitioned(c)) {      // the gather reaches n + mu - nu: links and N matrices from the halo-extended block (clover.hip), collective
        double2* lamN = stout_lambda_buffer(c);
        ARGCHK(lamN, "lqcd_stout_backprop: out of device memory");
        hipLaunchKernelGGL(stout_prep_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, lamN, G->data, Gs->data, W->data, rho, 1);
        HIPCHK(hipGetLastError());
        LQCHK(stout_gather_ext(c, U, lamN, G, rho));
        HIPCHK(hipStreamSynchronize(c->stream));
        return LQCD_OK;
    }
    hipLaunchKernelGGL(stout_prep_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, N->data, G->data, Gs->data, W->data, rho, 0);
    HIPCHK(hipGetLastError());
    hipLaunchKernelGGL(stout_gather_kernel, dim3(link_grid(c->geom)), dim3(256), 0, c->stream, c->geom, G->data, U->data, N->data, rho);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- Polyakov loop
// The second observable every trajectory of the reference's driver measures (measurement_methods Plaquette + Polyakov_loop in every test/*.toml;
// src/system/lqcd.jl:141 -> QCDMeasurements' Polyakov_measurement -> Gaugefields' calculate_Polyakov_loop(U, temp1, temp2)):
//     P = 1/(NC NX NY NZ) sum_x tr prod_{t = 0}^{NT-1} U_4(x, t)        [normalisation EXT-RECALL: the package's, as this file's author knows it]
// One thread per spatial site walks the time direction (links carry no boundary sign).  The time direction must not be partitioned.
namespace lqcd {
__global__ __launch_bounds__(256) void polyakov_kernel(Geom g, const double2* __restrict__ U, double* __restrict__ partial) {
    const int V3 = g.L[0] * g.L[1] * g.L[2];
    const int s3 = blockIdx.x * 256 + threadIdx.x;
    double re = 0.0, im = 0.0;
    if (s3 < V3) {
        int c[4] = {s3 % g.L[0], (s3 / g.L[0]) % g.L[1], s3 / (g.L[0] * g.L[1]), 0};
        const int Gs = glink_stride(g);
        cd acc[9], u[9], t[9];
        load_m3(acc, link_at(g, U, c, 3), Gs);
        for (c[3] = 1; c[3] < g.L[3]; c[3]++) {
            load_m3(u, link_at(g, U, c, 3), Gs);
            mm3(t, acc, u);
#pragma unroll
            for (int e = 0; e < 9; e++) acc[e] = t[e];
        }
        re = acc[0].re + acc[4].re + acc[8].re;
        im = acc[0].im + acc[4].im + acc[8].im;
    }
    __shared__ double sh[2][4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { re += __shfl_down(re, off, 64); im += __shfl_down(im, off, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = re; sh[1][threadIdx.x >> 6] = im; }
    __syncthreads();
    if (threadIdx.x < 2) partial[blockIdx.x * 2 + threadIdx.x] = (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]);
}
}  // namespace lqcd

extern "C" int lqcd_gauge_polyakov(lqcd_gauge_t U, double* re, double* im) {
    LQCHK(lqcd::links_flush_of(U));
    ARGCHK(U && re && im, "lqcd_gauge_polyakov: null argument");
    lqcd_ctx_s* c = U->ctx;
    ARGCHK(!c->geom.part[3], "lqcd_gauge_polyakov: the time direction is partitioned (the loop would cross ranks)");
    ARGCHK(c->local_peers.empty(), "lqcd_gauge_polyakov: not available on an in-process PE grid");
    HIPCHK(hipSetDevice(c->device));
    const int V3 = c->geom.L[0] * c->geom.L[1] * c->geom.L[2], nb = (V3 + 255) / 256;
    ARGCHK(nb <= MAX_PARTIAL_BLOCKS, "lqcd_gauge_polyakov: partial buffer too small");
    hipLaunchKernelGGL(polyakov_kernel, dim3(nb), dim3(256), 0, c->stream, c->geom, U->data, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 2, S_RED0, true, 0));
    HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const double norm = 1.0 / (3.0 * (double)c->gL[0] * (double)c->gL[1] * (double)c->gL[2]);
    *re = norm * c->h_scal[0];
    *im = norm * c->h_scal[1];
    return LQCD_OK;
}
