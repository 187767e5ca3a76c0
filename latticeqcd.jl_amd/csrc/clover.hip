// clover.hip -- the clover (Sheikholeslami-Wohlert) term of the Wilson operator (BASELINE.json
// configs[3] "Wilson-clover"; parameter names of the reference: Dirac_operator = "WilsonClover", Clover_coefficient,
// /root/reference/src/system/parameter_structs.jl:125, test/test_wilsonclover.toml:9).  The reference itself rejects this
// operator (src/system/universe.jl:129-131), so the definition is the textbook one (Luscher et al., hep-lat/9605038):
//     D_sw = 1 - kappa H + A - 1,   A(x) = 1 + i kappa c_sw sum_{mu<nu} sigma_{mu nu} F_{mu nu}(x),
//     sigma_{mu nu} = (i/2)[g_mu, g_nu],  F = (Q - Q^+)/8,  Q = the four plaquette loops around x in the mu-nu plane.
// sigma commutes with gamma5, so in the chiral basis chi_(+-) = (psi_upper -+ psi_lower)/sqrt2 (gamma5 = -offdiag(1,1) here) A is
// two Hermitian 6x6 blocks: 72 reals = 576 B/site, stored packed as 36 double2 per site,
//     [parity][chunk][36][64]:  block b at 18 b:  3 double2 = the 6 real diagonals, then the 15 upper-triangle entries (i < j).
// The term is applied as its own streaming pass tmp = A x (read 192 + 576, write 192 B/site) feeding the stencil's diagonal
// input (out = tmp - kappa H x): 960 B/site on top of the Dslash's 960; folding it into the stencil epilogue (compulsory
// 1536 B/site in total) is the obvious next step.
#include "lqcd_internal.h"

#include <complex>

// exchange hooks of the staple force (md.hip): buffers of 4 matrices per face site and grouped send / recv
int gf_buffers(lqcd_ctx_s* c);
int gf_exchange_rccl(lqcd_ctx_s* c, double2* const sendb[4], double2* const recvb[4], bool to_backward);

namespace lqcd {

struct CloverTables {
    double sr[6][2][2][2], si[6][2][2][2];   // sigma^b_plane[s][s'] (real, imaginary), b = 0: chi_+ block, 1: chi_- block
};

__host__ __device__ inline size_t clover_off(const Geom& g, int p, int i) { return (((size_t)p * g.nch + (size_t)(i >> 6)) * 36) * 64 + (i & 63); }
size_t clover_elems(const Geom& g) { return (size_t)2 * g.nch * 36 * 64; }
// six 3x3 matrices per site, one per plane (0,1) (0,2) (0,3) (1,2) (1,3) (2,3): [parity][chunk][plane][9][64]
__host__ __device__ inline size_t lambda_off(const Geom& g, int p, int i, int plane) {
    return ((((size_t)p * g.nch + (size_t)(i >> 6)) * 6 + plane) * 9) * 64 + (i & 63);
}
size_t clover_lambda_elems(const Geom& g) { return (size_t)2 * g.nch * 54 * 64; }

__device__ __forceinline__ void ldm(cd (&u)[9], const double2* __restrict__ U, const Geom& g, const int (&c)[4], int mu) {
    const int p = (c[0] + c[1] + c[2] + c[3]) & 1;
    const double2* b = U + glink_off(g, p, mu, coords_to_cb(g, c));
    const int Gs = glink_stride(g);
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = ld(b + (size_t)e * Gs);
}
// C = op(A) op(B), op = identity or dagger
template <bool DA, bool DB>
__device__ __forceinline__ void mmx(cd (&C)[9], const cd (&A)[9], const cd (&B)[9]) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                cd x = DA ? mk(A[k * 3 + a].re, -A[k * 3 + a].im) : A[a * 3 + k];
                cd y = DB ? mk(B[b * 3 + k].re, -B[b * 3 + k].im) : B[k * 3 + b];
                cfma(t, x, y);
            }
            C[a * 3 + b] = t;
        }
}
__device__ __forceinline__ void step(int (&d)[4], const Geom& g, int mu, int dir) {
    d[mu] += dir;
    if (d[mu] == g.L[mu]) d[mu] = 0;
    if (d[mu] < 0) d[mu] = g.L[mu] - 1;
}

// one thread per site: six field-strength matrices, then the two packed chiral blocks
// FROMQ: the clover sums Q_{mu nu}(x) come from qbuf ([parity][chunk][plane][9][64], built by the transport passes below -- the
// partitioned lattice); otherwise the four leaves are multiplied out here from local links.
template <bool FROMQ>
__global__ __launch_bounds__(64) void clover_build_kernel(Geom g, const double2* __restrict__ U, double2* __restrict__ clov, double coef, CloverTables tb,
                                                           const double2* __restrict__ qbuf) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    // blk[b][r][q] for r <= q: diagonal real, upper triangle complex
    cd blk[2][6][6];
#pragma unroll
    for (int b = 0; b < 2; b++)
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = 0; q < 6; q++) blk[b][r][q] = mk(r == q ? 1.0 : 0.0, 0.0);
    int plane = 0;
    for (int mu = 0; mu < 4; mu++)
        for (int nu = mu + 1; nu < 4; nu++, plane++) {
            cd Q[9];
            if constexpr (FROMQ) {
                const double2* qb = qbuf + ((((size_t)p * g.nch + (size_t)(i >> 6)) * 6 + plane) * 9) * 64 + (i & 63);
#pragma unroll
                for (int e = 0; e < 9; e++) Q[e] = ld(qb + (size_t)e * 64);
            } else {
            cd A[9], B[9], C[9], D[9], t1[9], t2[9], t3[9];
            int xm[4] = {c[0], c[1], c[2], c[3]}, xn[4] = {c[0], c[1], c[2], c[3]}, xpm[4] = {c[0], c[1], c[2], c[3]}, xpn[4] = {c[0], c[1], c[2], c[3]};
            step(xm, g, mu, -1); step(xn, g, nu, -1); step(xpm, g, mu, 1); step(xpn, g, nu, 1);
            int xmpn[4] = {xm[0], xm[1], xm[2], xm[3]}, xmn[4] = {xm[0], xm[1], xm[2], xm[3]}, xnpm[4] = {xn[0], xn[1], xn[2], xn[3]};
            step(xmpn, g, nu, 1); step(xmn, g, nu, -1); step(xnpm, g, mu, 1);
            // 1: U_mu(x) U_nu(x+mu) U_mu^+(x+nu) U_nu^+(x)
            ldm(A, U, g, c, mu); ldm(B, U, g, xpm, nu); ldm(C, U, g, xpn, mu); ldm(D, U, g, c, nu);
            mmx<false, false>(t1, A, B); mmx<false, true>(t2, t1, C); mmx<false, true>(Q, t2, D);
            // 2: U_nu(x) U_mu^+(x-mu+nu) U_nu^+(x-mu) U_mu(x-mu)
            ldm(A, U, g, c, nu); ldm(B, U, g, xmpn, mu); ldm(C, U, g, xm, nu); ldm(D, U, g, xm, mu);
            mmx<false, true>(t1, A, B); mmx<false, true>(t2, t1, C); mmx<false, false>(t3, t2, D);
#pragma unroll
            for (int e = 0; e < 9; e++) Q[e] = Q[e] + t3[e];
            // 3: U_mu^+(x-mu) U_nu^+(x-mu-nu) U_mu(x-mu-nu) U_nu(x-nu)
            ldm(A, U, g, xm, mu); ldm(B, U, g, xmn, nu); ldm(C, U, g, xmn, mu); ldm(D, U, g, xn, nu);
            mmx<true, true>(t1, A, B); mmx<false, false>(t2, t1, C); mmx<false, false>(t3, t2, D);
#pragma unroll
            for (int e = 0; e < 9; e++) Q[e] = Q[e] + t3[e];
            // 4: U_nu^+(x-nu) U_mu(x-nu) U_nu(x+mu-nu) U_mu^+(x)
            ldm(A, U, g, xn, nu); ldm(B, U, g, xn, mu); ldm(C, U, g, xnpm, nu); ldm(D, U, g, c, mu);
            mmx<true, false>(t1, A, B); mmx<false, false>(t2, t1, C); mmx<false, true>(t3, t2, D);
#pragma unroll
            for (int e = 0; e < 9; e++) Q[e] = Q[e] + t3[e];
            }
            // F = (Q - Q^+)/8 ;  blk_b += i coef sigma^b (x) F
#pragma unroll
            for (int ca = 0; ca < 3; ca++)
#pragma unroll
                for (int cb = 0; cb < 3; cb++) {
                    const cd F = mk(0.125 * (Q[ca * 3 + cb].re - Q[cb * 3 + ca].re), 0.125 * (Q[ca * 3 + cb].im + Q[cb * 3 + ca].im));
                    const cd iF = mk(-coef * F.im, coef * F.re);      // i coef F
#pragma unroll
                    for (int b = 0; b < 2; b++)
#pragma unroll
                        for (int s = 0; s < 2; s++)
#pragma unroll
                            for (int s2 = 0; s2 < 2; s2++) {
                                const int r = s * 3 + ca, q = s2 * 3 + cb;
                                if (r <= q) cfma(blk[b][r][q], mk(tb.sr[plane][b][s][s2], tb.si[plane][b][s][s2]), iF);
                            }
                }
        }
    double2* o = clov + clover_off(g, p, i);
#pragma unroll
    for (int b = 0; b < 2; b++) {
#pragma unroll
        for (int k = 0; k < 3; k++) st(o + (size_t)(18 * b + k) * 64, mk(blk[b][2 * k][2 * k].re, blk[b][2 * k + 1][2 * k + 1].re));
        int e = 3;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = r + 1; q < 6; q++, e++) st(o + (size_t)(18 * b + e) * 64, blk[b][r][q]);
    }
}

// out = sa (A in) + sz z on Wilson spinors, one thread per site.  pfix < 0: both parity blocks of FULL fields (blockIdx & 1 = parity);
// pfix = 0 | 1: that parity only (in / out / z are then the single blocks of EVEN / ODD fields, passed in slot pfix).  z may be null.
struct CloverApplyArgs {
    const double2* in[2];
    double2* out[2];
    const double2* z[2];
    double sa, sz;
    int pfix;
};
__global__ __launch_bounds__(64) void clover_apply_kernel(Geom g, const double2* __restrict__ clov, CloverApplyArgs k) {
    const int p = k.pfix < 0 ? (blockIdx.x & 1) : k.pfix, i = (k.pfix < 0 ? (blockIdx.x >> 1) : blockIdx.x) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    const int Vs = sp_stride(g);
    const double2* __restrict__ x = k.in[p] + sp_off(12, i);
    double2* __restrict__ y = k.out[p] + sp_off(12, i);
    const double2* __restrict__ a = clov + clover_off(g, p, i);
    cd psi[12], res[12];
#pragma unroll
    for (int j = 0; j < 12; j++) psi[j] = ld(x + (size_t)j * Vs);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const double sg = b == 0 ? -1.0 : 1.0;          // chi_+ = upper - lower, chi_- = upper + lower (normalised by the 1/2 below)
        cd chi[6], ych[6];
#pragma unroll
        for (int k = 0; k < 6; k++) chi[k] = mk(psi[k].re + sg * psi[6 + k].re, psi[k].im + sg * psi[6 + k].im);
        double dg[6];
#pragma unroll
        for (int k = 0; k < 3; k++) { const cd d = ld(a + (size_t)(18 * b + k) * 64); dg[2 * k] = d.re; dg[2 * k + 1] = d.im; }
#pragma unroll
        for (int r = 0; r < 6; r++) ych[r] = mk(dg[r] * chi[r].re, dg[r] * chi[r].im);
        int e = 3;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = r + 1; q < 6; q++, e++) {
                const cd m = ld(a + (size_t)(18 * b + e) * 64);
                cfma(ych[r], m, chi[q]);          // upper triangle
                cfma_conj(ych[q], m, chi[r]);     // lower triangle = conjugate
            }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            if (b == 0) { res[k] = mk(0.5 * ych[k].re, 0.5 * ych[k].im); res[6 + k] = mk(-0.5 * ych[k].re, -0.5 * ych[k].im); }
            else { res[k] = mk(res[k].re + 0.5 * ych[k].re, res[k].im + 0.5 * ych[k].im); res[6 + k] = mk(res[6 + k].re + 0.5 * ych[k].re, res[6 + k].im + 0.5 * ych[k].im); }
        }
    }
    if (k.z[p]) {
        const double2* __restrict__ z = k.z[p] + sp_off(12, i);
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const cd zv = ld(z + (size_t)j * Vs);
            res[j] = mk(fma(k.sa, res[j].re, k.sz * zv.re), fma(k.sa, res[j].im, k.sz * zv.im));
        }
    } else if (k.sa != 1.0) {
#pragma unroll
        for (int j = 0; j < 12; j++) res[j] = mk(k.sa * res[j].re, k.sa * res[j].im);
    }
#pragma unroll
    for (int j = 0; j < 12; j++) st(y + (size_t)j * Vs, res[j]);
}

// inv = A^-1 block by block (the inverse of a Hermitian block is Hermitian: same packed format, same apply kernel).  In-place
// Gauss-Jordan without pivoting, fully unrolled in registers; A = 1 + O(kappa c_sw F) is positive definite for every sensible c_sw.
// One thread per (site, chiral block).
__global__ __launch_bounds__(64) void clover_invert_kernel(Geom g, const double2* __restrict__ clov, double2* __restrict__ inv) {
    const int p = blockIdx.x & 1, b = blockIdx.y, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    const double2* __restrict__ a = clov + clover_off(g, p, i) + (size_t)(18 * b) * 64;
    double2* __restrict__ o = inv + clover_off(g, p, i) + (size_t)(18 * b) * 64;
    cd m[6][6];
#pragma unroll
    for (int k = 0; k < 3; k++) { const cd d = ld(a + (size_t)k * 64); m[2 * k][2 * k] = mk(d.re, 0.0); m[2 * k + 1][2 * k + 1] = mk(d.im, 0.0); }
    {
        int e = 3;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = r + 1; q < 6; q++, e++) { const cd v = ld(a + (size_t)e * 64); m[r][q] = v; m[q][r] = mk(v.re, -v.im); }
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        // 1 / m[k][k] (complex in the intermediate steps)
        const double dn = 1.0 / (m[k][k].re * m[k][k].re + m[k][k].im * m[k][k].im);
        const cd piv = mk(m[k][k].re * dn, -m[k][k].im * dn);
        m[k][k] = mk(1.0, 0.0);
#pragma unroll
        for (int j = 0; j < 6; j++) { const cd t = m[k][j]; m[k][j] = mk(t.re * piv.re - t.im * piv.im, t.re * piv.im + t.im * piv.re); }
#pragma unroll
        for (int r = 0; r < 6; r++) {
            if (r == k) continue;
            const cd f = m[r][k];
            m[r][k] = mk(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < 6; j++) cfma(m[r][j], mk(-f.re, -f.im), m[k][j]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) st(o + (size_t)k * 64, mk(m[2 * k][2 * k].re, m[2 * k + 1][2 * k + 1].re));
    {
        int e = 3;
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = r + 1; q < 6; q++, e++)      // symmetrise the round-off: (m[r][q] + conj(m[q][r])) / 2
                st(o + (size_t)e * 64, mk(0.5 * (m[r][q].re + m[q][r].re), 0.5 * (m[r][q].im - m[q][r].im)));
    }
}

// sigma_{mu nu} = (i/2)[g_mu, g_nu] in the chiral basis e^+_s = (e_s - e_{s+2})/sqrt2, e^-_s = (e_s + e_{s+2})/sqrt2
static int clover_tables(CloverTables& tb) {
    typedef std::complex<double> cx;
    cx G[4][4][4] = {};
    const cx ipow[4] = {cx(1, 0), cx(0, 1), cx(-1, 0), cx(0, -1)};
    for (int mu = 0; mu < 3; mu++)
        for (int a = 0; a < 4; a++) G[mu][a][PERM[mu][a]] = ipow[GK[mu][a] & 3];
    for (int a = 0; a < 4; a++) G[3][a][a] = a < 2 ? 1.0 : -1.0;
    int plane = 0;
    for (int mu = 0; mu < 4; mu++)
        for (int nu = mu + 1; nu < 4; nu++, plane++) {
            cx S[4][4];
            for (int a = 0; a < 4; a++)
                for (int b = 0; b < 4; b++) {
                    cx t = 0;
                    for (int k = 0; k < 4; k++) t += G[mu][a][k] * G[nu][k][b] - G[nu][a][k] * G[mu][k][b];
                    S[a][b] = cx(0, 0.5) * t;
                }
            for (int s = 0; s < 2; s++)
                for (int t = 0; t < 2; t++) {
                    const cx pp = 0.5 * (S[s][t] - S[s][t + 2] - S[s + 2][t] + S[s + 2][t + 2]);
                    const cx mm = 0.5 * (S[s][t] + S[s][t + 2] + S[s + 2][t] + S[s + 2][t + 2]);
                    const cx pm = 0.5 * (S[s][t] + S[s][t + 2] - S[s + 2][t] - S[s + 2][t + 2]);   // <e^+_s| sigma |e^-_t> must vanish
                    if (std::abs(pm) > 1e-14) { set_error("clover: sigma does not commute with gamma5 in this basis"); return LQCD_ERR_ARG; }
                    tb.sr[plane][0][s][t] = pp.real(); tb.si[plane][0][s][t] = pp.imag();
                    tb.sr[plane][1][s][t] = mm.real(); tb.si[plane][1][s][t] = mm.imag();
                }
        }
    return LQCD_OK;
}

// ---- the clover sums on a partitioned lattice (or with the tunable clover_transport = 1): Q = (1 + T_nu)(1 + T_mu) P with the
// elementary plaquette P_{mu nu}(y) = U_mu(y) U_nu(y+mu) U_mu^+(y+nu) U_nu^+(y) and the backward transport
// T_d[M](x) = U_d^+(x-d) M(x-d) U_d(x-d): P needs forward link ghosts only (the ones the staple force exchanges), every transport is
// one face exchange of 3x3 matrices to the +d neighbour (the sender conjugates with its own links) -- no corner exchange.
struct CloverQArgs {
    Geom g;
    const double2* U;
    const double2* ghost[4];   // x_d = 0 link slices of the +d neighbours ([parity][nu][9][Fh]); null when unpartitioned
    const double2* in;         // [parity][chunk][plane][9][64]
    double2* out;
    double2* wsend[4];         // transported upper-face matrices for the +d neighbour: [parity of the sender site][slot][9][Fh]
    const double2* wrecv[4];
    int stage;                 // 0: transport along the first index of each plane, 1: along the second
};
__device__ __forceinline__ size_t qoff(const Geom& g, int p, int i, int plane) { return lambda_off(g, p, i, plane); }
__device__ __forceinline__ void ld_link_fwd(cd (&u)[9], const CloverQArgs& k, const int (&c)[4], int dir, int nu) {
    const Geom& g = k.g;
    int d[4] = {c[0], c[1], c[2], c[3]};
    d[dir] += 1;
    if (d[dir] == g.L[dir]) {
        d[dir] = 0;
        if (g.part[dir]) {
            const int p = (d[0] + d[1] + d[2] + d[3]) & 1, Fh = face_half_sites(g, dir), f = coords_to_face(g, dir, d);
            const double2* b = k.ghost[dir] + ((size_t)(p * 4 + nu) * 9) * Fh + f;
#pragma unroll
            for (int e = 0; e < 9; e++) u[e] = ld(b + (size_t)e * Fh);
            return;
        }
    }
    ldm(u, k.U, g, d, nu);
}
__device__ __forceinline__ void plane_dirs(int plane, int& mu, int& nu) {
    mu = plane < 3 ? 0 : (plane < 5 ? 1 : 2);
    nu = plane < 3 ? plane + 1 : (plane < 5 ? plane - 1 : 3);
}
// slot of `plane` in the face buffer of direction d for this stage (-1: d does not transport this plane in this stage)
__device__ __forceinline__ int plane_slot(int plane, int d, int stage) {
    int mu, nu;
    plane_dirs(plane, mu, nu);
    if (stage == 0) return mu == d ? nu - d - 1 : -1;     // planes (d, nu), nu > d
    return nu == d ? mu : -1;                             // planes (mu, d), mu < d
}
__global__ __launch_bounds__(64) void clover_plaq_kernel(CloverQArgs k) {
    const Geom& g = k.g;
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    for (int plane = 0; plane < 6; plane++) {
        int mu, nu;
        plane_dirs(plane, mu, nu);
        cd A[9], B[9], t1[9], t2[9];
        ldm(A, k.U, g, c, mu);
        ld_link_fwd(B, k, c, mu, nu);
        mmx<false, false>(t1, A, B);
        ld_link_fwd(A, k, c, nu, mu);
        mmx<false, true>(t2, t1, A);
        ldm(B, k.U, g, c, nu);
        mmx<false, true>(t1, t2, B);
        double2* o = k.out + qoff(g, p, i, plane);
#pragma unroll
        for (int e = 0; e < 9; e++) st(o + (size_t)e * 64, t1[e]);
    }
}
// W = U_d^+(y) M(y) U_d(y)
__device__ __forceinline__ void conj_transport(cd (&W)[9], const CloverQArgs& k, const int (&y)[4], int d, int plane) {
    const Geom& g = k.g;
    const int py = (y[0] + y[1] + y[2] + y[3]) & 1;
    cd Ud[9], M[9], t[9];
    ldm(Ud, k.U, g, y, d);
    const double2* m = k.in + qoff(g, py, coords_to_cb(g, y), plane);
#pragma unroll
    for (int e = 0; e < 9; e++) M[e] = ld(m + (size_t)e * 64);
    mmx<true, false>(t, Ud, M);
    mmx<false, false>(W, t, Ud);
}
// upper faces (y_d = L_d - 1) of the partitioned directions d = blockIdx.y
__global__ __launch_bounds__(128) void clover_transport_face_kernel(CloverQArgs k) {
    const Geom& g = k.g;
    const int d = blockIdx.y;
    if (!g.part[d]) return;
    const int Fh = face_half_sites(g, d), t = blockIdx.x * 128 + threadIdx.x;
    if (t >= 2 * Fh) return;
    const int py = t / Fh, f = t - py * Fh;
    int y[4];
    face_to_coords(g, d, g.L[d] - 1, py, f, y);
    for (int plane = 0; plane < 6; plane++) {
        const int slot = plane_slot(plane, d, k.stage);
        if (slot < 0) continue;
        cd W[9];
        conj_transport(W, k, y, d, plane);
        double2* o = k.wsend[d] + ((size_t)(py * 4 + slot) * 9) * Fh + f;
#pragma unroll
        for (int e = 0; e < 9; e++) st(o + (size_t)e * Fh, W[e]);
    }
}
__global__ __launch_bounds__(64) void clover_transport_kernel(CloverQArgs k) {
    const Geom& g = k.g;
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    for (int plane = 0; plane < 6; plane++) {
        int mu, nu;
        plane_dirs(plane, mu, nu);
        const int d = k.stage == 0 ? mu : nu;
        cd W[9];
        if (c[d] == 0 && g.part[d]) {
            const int Fh = face_half_sites(g, d), f = coords_to_face(g, d, c), slot = plane_slot(plane, d, k.stage);
            const double2* b = k.wrecv[d] + ((size_t)((1 - p) * 4 + slot) * 9) * Fh + f;      // the sender site y = x - d has the other parity
#pragma unroll
            for (int e = 0; e < 9; e++) W[e] = ld(b + (size_t)e * Fh);
        } else {
            int y[4] = {c[0], c[1], c[2], c[3]};
            step(y, g, d, -1);
            conj_transport(W, k, y, d, plane);
        }
        const double2* m = k.in + qoff(g, p, i, plane);
        double2* o = k.out + qoff(g, p, i, plane);
#pragma unroll
        for (int e = 0; e < 9; e++) { const cd v = ld(m + (size_t)e * 64); st(o + (size_t)e * 64, mk(v.re + W[e].re, v.im + W[e].im)); }
    }
}

static int clover_q_transport(lqcd_ctx_s* c, const lqcd_gauge_s* U, double2* q0, double2* q1) {
    const bool part = any_partitioned(c);
    if (part) {
        ARGCHK(c->local_peers.empty(), "clover term: not available on an in-process PE grid");
        LQCHK(gf_buffers(c));
        for (int mu = 0; mu < 4; mu++)
            if (c->geom.part[mu]) LQCHK(gauge_pack_face(const_cast<lqcd_gauge_s*>(U), mu, c->gf_gsend[mu]));
        LQCHK(gf_exchange_rccl(c, c->gf_gsend, c->gf_ghost, true));
    }
    CloverQArgs k;
    k.g = c->geom;
    k.U = U->data;
    int maxf = 0;
    for (int mu = 0; mu < 4; mu++) {
        k.ghost[mu] = c->gf_ghost[mu]; k.wsend[mu] = c->gf_wsend[mu]; k.wrecv[mu] = c->gf_wrecv[mu];
        if (c->geom.part[mu]) maxf = std::max(maxf, face_half_sites(c->geom, mu));
    }
    k.in = nullptr; k.out = q0; k.stage = 0;
    hipLaunchKernelGGL(clover_plaq_kernel, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, k);
    HIPCHK(hipGetLastError());
    for (int stage = 0; stage < 2; stage++) {
        k.stage = stage;
        k.in = stage == 0 ? q0 : q1;
        k.out = stage == 0 ? q1 : q0;
        if (part) {
            hipLaunchKernelGGL(clover_transport_face_kernel, dim3((2 * maxf + 127) / 128, 4), dim3(128), 0, c->stream, k);
            HIPCHK(hipGetLastError());
            LQCHK(gf_exchange_rccl(c, c->gf_wsend, c->gf_wrecv, false));
        }
        hipLaunchKernelGGL(clover_transport_kernel, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, k);
        HIPCHK(hipGetLastError());
    }
    return LQCD_OK;       // Q in q0
}

int clover_build(lqcd_ctx_s* c, const lqcd_gauge_s* U, double2* clov, double kappa, double csw) {
    CloverTables tb;
    LQCHK(clover_tables(tb));
    if (any_partitioned(c) || c->tun.clover_transport) {
        for (int j = 0; j < 2; j++)
            if (!c->clover_q[j]) HIPCHK(hipMalloc((void**)&c->clover_q[j], clover_lambda_elems(c->geom) * sizeof(double2)));
        LQCHK(clover_q_transport(c, U, c->clover_q[0], c->clover_q[1]));
        hipLaunchKernelGGL(clover_build_kernel<true>, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, c->geom, U->data, clov, kappa * csw, tb,
                           c->clover_q[0]);
    } else {
        hipLaunchKernelGGL(clover_build_kernel<false>, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, c->geom, U->data, clov, kappa * csw, tb,
                           (const double2*)nullptr);
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int clover_apply(lqcd_ctx_s* c, const double2* clov, lqcd_spinor_s* out, lqcd_spinor_s* in) {
    CloverApplyArgs k;
    for (int p = 0; p < 2; p++) { k.in[p] = spinor_block(in, p); k.out[p] = spinor_block(out, p); k.z[p] = nullptr; }
    k.sa = 1.0; k.sz = 0.0; k.pfix = -1;
    hipLaunchKernelGGL(clover_apply_kernel, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, c->geom, clov, k);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// one parity block: out = sa (C in) + sz z with raw block pointers (the even-odd solver works on EVEN / ODD fields)
int clover_apply_parity(lqcd_ctx_s* c, const double2* clov, int parity, double2* out, const double2* in, double sa, const double2* z, double sz) {
    CloverApplyArgs k;
    for (int p = 0; p < 2; p++) { k.in[p] = in; k.out[p] = out; k.z[p] = z; }
    k.sa = sa; k.sz = sz; k.pfix = parity;
    hipLaunchKernelGGL(clover_apply_kernel, dim3(c->geom.nch), dim3(64), 0, c->stream, c->geom, clov, k);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int clover_invert(lqcd_ctx_s* c, const double2* clov, double2* inv) {
    hipLaunchKernelGGL(clover_invert_kernel, dim3(2 * c->geom.nch, 2), dim3(64), 0, c->stream, c->geom, clov, inv);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// ------------------------------------------------------------------------------------------------ force of the clover term
// S_f = phi^+ (D_sw^+ D_sw)^-1 phi, X = (D_sw^+ D_sw)^-1 phi, Y = D_sw X (2-flavour Wilson-clover HMC, BASELINE.json configs[3]).  The
// hopping part of dS_f is the Wilson force sweep (force.hip); the clover part is
//     dS = (kappa c_sw / 4) sum_x sum_{mu<nu} Im tr( dQ_{mu nu}(x) Lambda^{mu nu}(x) ),
//     Lambda^{mu nu}(x) = M + M^+,  M = sum_{s} sigma^{mu nu}_{s s'} X_{s'}(x) Y_s(x)^+   (colour outer products; one s' per row s).
// Pass 1 builds the six Hermitian Lambda matrices per site ([parity][chunk][plane][9][64]).  Pass 2 is a gather per link (z, rho):
// for every nu != rho and both sides s = +-nu the plaquette through the link is a leaf of Q at each of its four corners c, so with
// the loop  z -> z+rho -> z+rho+s nu -> z+s nu -> z  (links U A B C)
//     W = U (Lambda(c1) A B C + A Lambda(c2) B C + A B Lambda(c3) C + A B C Lambda(c0)),
// and, o = s * sign(rho < nu) being the orientation of that loop relative to the counter-clockwise leaves,
//     G_rho(z) += -i (kappa c_sw / 8) W   (o = +1)        G_rho(z) += +i (kappa c_sw / 8) W^+   (o = -1)
// in the convention of the other force fields (dS/d eps[U -> exp(i eps T) U] = -2 Im tr(T G)).  The CPU check in tests/ computes the
// same field as a scatter over (site, plane, leaf, step).
struct SigmaTab { int col[6][4]; double re[6][4], im[6][4]; };


__global__ __launch_bounds__(64) void clover_lambda_kernel(Geom g, const double2* __restrict__ X0, const double2* __restrict__ X1,
                                                            const double2* __restrict__ Y0, const double2* __restrict__ Y1,
                                                            double2* __restrict__ lam, SigmaTab tb) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    const int Vs = sp_stride(g);
    const double2* __restrict__ xp = (p ? X1 : X0) + sp_off(12, i);
    const double2* __restrict__ yp = (p ? Y1 : Y0) + sp_off(12, i);
    cd x[12], y[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { x[j] = ld(xp + (size_t)j * Vs); y[j] = ld(yp + (size_t)j * Vs); }
#pragma unroll
    for (int plane = 0; plane < 6; plane++) {
        cd M[9];
#pragma unroll
        for (int e = 0; e < 9; e++) M[e] = mk(0.0, 0.0);
#pragma unroll
        for (int sp = 0; sp < 4; sp++) {
            const cd sg = mk(tb.re[plane][sp], tb.im[plane][sp]);
            const int s2 = tb.col[plane][sp];
#pragma unroll
            for (int b = 0; b < 3; b++) {
                // sigma_{s s2} X_{s2 b}: s2 is a table entry, select without dynamic register indexing
                cd xv = x[b];
                if (s2 == 1) xv = x[3 + b];
                if (s2 == 2) xv = x[6 + b];
                if (s2 == 3) xv = x[9 + b];
                cd sx = mk(0.0, 0.0);
                cfma(sx, sg, xv);
#pragma unroll
                for (int a = 0; a < 3; a++) cfma_conj(M[b * 3 + a], y[sp * 3 + a], sx);      // += conj(Y_{s a}) * (sigma X)_b
            }
        }
        double2* o = lam + lambda_off(g, p, i, plane);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) st(o + (size_t)(a * 3 + b) * 64, mk(M[a * 3 + b].re + M[b * 3 + a].re, M[a * 3 + b].im - M[b * 3 + a].im));
    }
}

__device__ __forceinline__ void ldlam(cd (&l)[9], const double2* __restrict__ lam, const Geom& g, const int (&c)[4], int plane) {
    const int p = (c[0] + c[1] + c[2] + c[3]) & 1;
    const double2* b = lam + lambda_off(g, p, coords_to_cb(g, c), plane);
#pragma unroll
    for (int e = 0; e < 9; e++) l[e] = ld(b + (size_t)e * 64);
}
__device__ __forceinline__ void dag9(cd (&u)[9]) {
    cd t[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) t[a * 3 + b] = mk(u[b * 3 + a].re, -u[b * 3 + a].im);
#pragma unroll
    for (int e = 0; e < 9; e++) u[e] = t[e];
}
// T = T R + P L  (all 3x3)
__device__ __forceinline__ void horner(cd (&T)[9], const cd (&R)[9], const cd (&P)[9], const cd (&Lm)[9]) {
    cd o[9];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd t = mk(0.0, 0.0);
#pragma unroll
            for (int k = 0; k < 3; k++) { cfma(t, T[a * 3 + k], R[k * 3 + b]); cfma(t, P[a * 3 + k], Lm[k * 3 + b]); }
            o[a * 3 + b] = t;
        }
#pragma unroll
    for (int e = 0; e < 9; e++) T[e] = o[e];
}

// ---- partitioned lattice: the gather above reaches z+rho+-nu, i.e. corners of the neighbouring ranks.  The links and the Lambda
// matrices are copied into one halo-extended block ext[m][9][E0*E1*E2*E3] (m = 0..3 links, 4..9 planes; extents E = L + 2,
// local site x at x + 1) whose depth-1 halos are filled direction by direction: the boundary layers exchanged for direction d span
// the full extended cross-section, so they carry the halos of the directions done before -- corners arrive without a corner
// message.  Unpartitioned directions take the same pack / unpack with a device copy instead of a message.  Once per MD step.
struct ExtGeom {
    int L[4], E[4];
    size_t n;          // E0 E1 E2 E3
};
__host__ __device__ inline size_t ext_site(const ExtGeom& eg, const int (&c)[4]) {      // c = local coordinates in [-1, L]
    return (size_t)(c[0] + 1) + (size_t)eg.E[0] * ((size_t)(c[1] + 1) + (size_t)eg.E[1] * ((size_t)(c[2] + 1) + (size_t)eg.E[2] * (size_t)(c[3] + 1)));
}
__global__ __launch_bounds__(64) void clover_ext_fill_kernel(Geom g, ExtGeom eg, const double2* __restrict__ U, const double2* __restrict__ lam,
                                                              double2* __restrict__ ext) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + threadIdx.x;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    const size_t e = ext_site(eg, c);
    const int Gs = glink_stride(g);
    for (int m = 0; m < 10; m++) {
        const double2* src = m < 4 ? U + glink_off(g, p, m, i) : lam + lambda_off(g, p, i, m - 4);
        const int st_ = m < 4 ? Gs : 64;
#pragma unroll
        for (int q = 0; q < 9; q++) st(ext + ((size_t)m * 9 + q) * eg.n + e, ld(src + (size_t)q * st_));
    }
}
// cross-section index f of direction d <-> extended coordinates (the other three run over their full extended range)
__device__ __forceinline__ size_t ext_face_site(const ExtGeom& eg, int d, size_t f, int layer) {
    int e[4];
    for (int k = 0; k < 4; k++) {
        if (k == d) { e[k] = layer; continue; }
        e[k] = (int)(f % (size_t)eg.E[k]);
        f /= (size_t)eg.E[k];
    }
    return (size_t)e[0] + (size_t)eg.E[0] * ((size_t)e[1] + (size_t)eg.E[1] * ((size_t)e[2] + (size_t)eg.E[2] * (size_t)e[3]));
}
// pack = true: buf[side][m*9+q][f] <- ext at layer (side 0: e_d = 1, the lower boundary layer; side 1: e_d = L_d, the upper one)
// pack = false: ext at halo layer (side 0: e_d = 0 ; side 1: e_d = L_d + 1) <- buf[side]
__global__ __launch_bounds__(256) void clover_ext_face_kernel(ExtGeom eg, int d, size_t F, double2* __restrict__ ext, double2* __restrict__ buf, int pack) {
    const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int side = blockIdx.y;
    if (f >= F) return;
    const int layer = pack ? (side ? eg.L[d] : 1) : (side ? eg.L[d] + 1 : 0);
    const size_t e = ext_face_site(eg, d, f, layer);
    double2* b = buf + (size_t)side * 90 * F + f;
    for (int j = 0; j < 90; j++) {
        if (pack) st(b + (size_t)j * F, ld(ext + (size_t)j * eg.n + e));
        else st(ext + (size_t)j * eg.n + e, ld(b + (size_t)j * F));
    }
}

// sources of the gather: the local fields (periodic wrap) or the extended block (no wrap)
struct ForceSrc {
    const double2* U;
    const double2* lam;
    const double2* ext;
    ExtGeom eg;
};
template <bool EXT>
__device__ __forceinline__ void fstep(int (&d)[4], const Geom& g, int mu, int dir) {
    if constexpr (EXT) d[mu] += dir;
    else step(d, g, mu, dir);
}
template <bool EXT>
__device__ __forceinline__ void get_link(cd (&u)[9], const ForceSrc& s, const Geom& g, const int (&c)[4], int mu) {
    if constexpr (EXT) {
        const double2* b = s.ext + (size_t)mu * 9 * s.eg.n + ext_site(s.eg, c);
#pragma unroll
        for (int e = 0; e < 9; e++) u[e] = ld(b + (size_t)e * s.eg.n);
    } else ldm(u, s.U, g, c, mu);
}
template <bool EXT>
__device__ __forceinline__ void get_lam(cd (&l)[9], const ForceSrc& s, const Geom& g, const int (&c)[4], int plane) {
    if constexpr (EXT) {
        const double2* b = s.ext + (size_t)(4 + plane) * 9 * s.eg.n + ext_site(s.eg, c);
#pragma unroll
        for (int e = 0; e < 9; e++) l[e] = ld(b + (size_t)e * s.eg.n);
    } else ldlam(l, s.lam, g, c, plane);
}

// one thread per link: blockDim = 256 = 64 sites x 4 directions
template <bool EXT>
__global__ __launch_bounds__(256) void clover_force_kernel(Geom g, ForceSrc src, double2* __restrict__ out, double cf, double scale, int acc) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), rho = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    int z[4];
    cb_to_coords(g, p, i, z);
    cd Wp[9], Wm[9];
#pragma unroll
    for (int e = 0; e < 9; e++) { Wp[e] = mk(0.0, 0.0); Wm[e] = mk(0.0, 0.0); }
    int zr[4] = {z[0], z[1], z[2], z[3]};
    fstep<EXT>(zr, g, rho, 1);                                   // c1 = z + rho
    for (int nu = 0; nu < 4; nu++) {
        if (nu == rho) continue;
        const int mu0 = rho < nu ? rho : nu, nu0 = rho < nu ? nu : rho;
        const int plane = mu0 == 0 ? nu0 - 1 : (mu0 == 1 ? nu0 + 1 : 5);      // (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
        for (int sd = 1; sd >= -1; sd -= 2) {
            int c2[4] = {zr[0], zr[1], zr[2], zr[3]}, c3[4] = {z[0], z[1], z[2], z[3]};
            fstep<EXT>(c2, g, nu, sd);                           // z + rho + s nu
            fstep<EXT>(c3, g, nu, sd);                           // z + s nu
            cd A[9], B[9], L[9], T[9], P[9], Q[9];
            if (sd > 0) get_link<EXT>(A, src, g, zr, nu);              // A: z+rho -> z+rho+s nu
            else { get_link<EXT>(A, src, g, c2, nu); dag9(A); }
            get_lam<EXT>(L, src, g, zr, plane);
            mmx<false, false>(T, L, A);                    // Lambda(c1) A
            get_lam<EXT>(L, src, g, c2, plane);
            mmx<false, false>(Q, A, L);                    // A Lambda(c2)
#pragma unroll
            for (int e = 0; e < 9; e++) T[e] = mk(T[e].re + Q[e].re, T[e].im + Q[e].im);
            get_link<EXT>(B, src, g, c3, rho);                         // B: z+rho+s nu -> z+s nu  = U_rho(z + s nu)^+
            dag9(B);
            mmx<false, false>(P, A, B);                    // A B
            get_lam<EXT>(L, src, g, c3, plane);
            horner(T, B, P, L);                            // (..) B + A B Lambda(c3)
            if (sd > 0) { get_link<EXT>(B, src, g, z, nu); dag9(B); }  // C: z+s nu -> z
            else get_link<EXT>(B, src, g, c3, nu);
            mmx<false, false>(Q, P, B);                    // A B C
            get_lam<EXT>(L, src, g, z, plane);
            horner(T, B, Q, L);                            // (..) C + A B C Lambda(c0)
            get_link<EXT>(A, src, g, z, rho);
            mmx<false, false>(Q, A, T);                    // W = U (..)
            const int o = sd * (rho < nu ? 1 : -1);
#pragma unroll
            for (int e = 0; e < 9; e++) {
                if (o > 0) Wp[e] = mk(Wp[e].re + Q[e].re, Wp[e].im + Q[e].im);
                else Wm[e] = mk(Wm[e].re + Q[e].re, Wm[e].im + Q[e].im);
            }
        }
    }
    // G = -i cf Wp + i cf Wm^+ ;  out = (acc ? out : 0) + scale G
    double2* o = out + glink_off(g, p, rho, i);
    const int Gs = glink_stride(g);
    const double f = cf * scale;
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const cd wp = Wp[a * 3 + b], wm = Wm[b * 3 + a];              // (Wm^+)_{ab} = conj(Wm_{ba})
            // -i wp = (wp.im, -wp.re);  +i conj(wm) = i (wm.re - i wm.im) = (wm.im, wm.re)
            cd v = mk(f * (wp.im + wm.im), f * (-wp.re + wm.re));
            if (acc) { const cd old = ld(o + (size_t)(a * 3 + b) * Gs); v = mk(v.re + old.re, v.im + old.im); }
            st(o + (size_t)(a * 3 + b) * Gs, v);
        }
}

static int sigma_table(SigmaTab& tb) {
    typedef std::complex<double> cx;
    cx G[4][4][4] = {};
    const cx ipow[4] = {cx(1, 0), cx(0, 1), cx(-1, 0), cx(0, -1)};
    for (int mu = 0; mu < 3; mu++)
        for (int a = 0; a < 4; a++) G[mu][a][PERM[mu][a]] = ipow[GK[mu][a] & 3];
    for (int a = 0; a < 4; a++) G[3][a][a] = a < 2 ? 1.0 : -1.0;
    int plane = 0;
    for (int mu = 0; mu < 4; mu++)
        for (int nu = mu + 1; nu < 4; nu++, plane++)
            for (int a = 0; a < 4; a++) {
                int found = 0;
                for (int b = 0; b < 4; b++) {
                    cx t = 0;
                    for (int k = 0; k < 4; k++) t += G[mu][a][k] * G[nu][k][b] - G[nu][a][k] * G[mu][k][b];
                    t *= cx(0, 0.5);
                    if (std::abs(t) > 1e-14) { tb.col[plane][a] = b; tb.re[plane][a] = t.real(); tb.im[plane][a] = t.imag(); found++; }
                }
                if (found != 1) { set_error("clover force: sigma is not a generalised permutation matrix in this basis"); return LQCD_ERR_ARG; }
            }
    return LQCD_OK;
}

// halo-extended copy of the links and the Lambda matrices (partitioned lattice)
static int clover_ext_build(lqcd_ctx_s* c, const lqcd_gauge_s* U, const double2* lam, ExtGeom& eg) {
    for (int k = 0; k < 4; k++) { eg.L[k] = c->geom.L[k]; eg.E[k] = c->geom.L[k] + 2; }
    eg.n = (size_t)eg.E[0] * eg.E[1] * eg.E[2] * eg.E[3];
    size_t Fmax = 0;
    for (int d = 0; d < 4; d++) Fmax = std::max(Fmax, eg.n / (size_t)eg.E[d]);
    const size_t ext_bytes = 90 * eg.n * sizeof(double2), buf_bytes = 2 * 90 * Fmax * sizeof(double2);
    if (c->clover_ext_bytes < ext_bytes) {
        (void)hipFree(c->clover_ext);
        c->clover_ext = nullptr; c->clover_ext_bytes = 0;
        HIPCHK(hipMalloc((void**)&c->clover_ext, ext_bytes));
        c->clover_ext_bytes = ext_bytes;
    }
    for (int j = 0; j < 2; j++)
        if (c->clover_ext_buf_bytes[j] < buf_bytes) {
            (void)hipFree(c->clover_ext_buf[j]);
            c->clover_ext_buf[j] = nullptr; c->clover_ext_buf_bytes[j] = 0;
            HIPCHK(hipMalloc((void**)&c->clover_ext_buf[j], buf_bytes));
            c->clover_ext_buf_bytes[j] = buf_bytes;
        }
    double2* ext = c->clover_ext;
    double2 *sendb = c->clover_ext_buf[0], *recvb = c->clover_ext_buf[1];
    hipLaunchKernelGGL(clover_ext_fill_kernel, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, c->geom, eg, U->data, lam, ext);
    HIPCHK(hipGetLastError());
    for (int d = 0; d < 4; d++) {
        const size_t F = eg.n / (size_t)eg.E[d];
        const dim3 grid((unsigned)((F + 255) / 256), 2);
        hipLaunchKernelGGL(clover_ext_face_kernel, grid, dim3(256), 0, c->stream, eg, d, F, ext, sendb, 1);
        HIPCHK(hipGetLastError());
        double2* src = sendb;
        if (c->geom.part[d]) {
            // my lower boundary layer (side 0) is the -d neighbour's upper halo (side 1), my upper layer its +d neighbour's lower halo
            ARGCHK(c->has_comm, "clover force: communicator not initialised (call lqcd_ctx_comm_init or lqcd_ctx_peer_init)");
            const size_t nb = 90 * F * sizeof(double2);     // bytes per side
            const CommXfer x[2] = {{sendb, recvb + 90 * F, nb, d, 1},            // my lower layer travels backward; their lower layer -> my upper halo
                                   {sendb + 90 * F, recvb, nb, d, 0}};           // my upper layer travels forward; their upper layer -> my lower halo
            LQCHK(comm_sendrecv(c, x, 2, c->stream, false));
            src = recvb;
        } else {
            // periodic wrap on this rank: lower halo <- own upper layer, upper halo <- own lower layer
            HIPCHK(hipMemcpyAsync(recvb, sendb + 90 * F, 90 * F * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(recvb + 90 * F, sendb, 90 * F * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
            src = recvb;
        }
        hipLaunchKernelGGL(clover_ext_face_kernel, grid, dim3(256), 0, c->stream, eg, d, F, ext, src, 0);
        HIPCHK(hipGetLastError());
    }
    return LQCD_OK;
}

// ---- stout back-propagation on a partitioned lattice (md.hip, "stout smearing"): the same 24-loop gather as stout_gather_kernel, reading the links and the N matrices
// (four per site, carried in planes 0..3 of a Lambda-shaped buffer) from the halo-extended block -- it reaches n + mu - nu, a corner of the neighbouring ranks.
__global__ __launch_bounds__(256) void stout_gather_ext_kernel(Geom g, ForceSrc src, double2* __restrict__ G, double rho) {
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    cd a[9], Na[9], acc[9];
    get_link<true>(a, src, g, c, mu);
    get_lam<true>(Na, src, g, c, mu);
#pragma unroll
    for (int e = 0; e < 9; e++) acc[e] = mk(0.0, 0.0);
    for (int nu = 0; nu < 4; nu++) {
        if (nu == mu) continue;
        cd b[9], cc[9], d[9], Nb[9], Nc[9], Nd[9], t1[9], t2[9], t3[9], X[9];
        int cm[4] = {c[0], c[1], c[2], c[3]}, cn[4] = {c[0], c[1], c[2], c[3]};
        cm[mu] += 1;
        cn[nu] += 1;
        // the plaquette (n; mu, nu), this link as a: b = U_nu(n+mu), c = U_mu(n+nu), d = U_nu(n)
        get_link<true>(b, src, g, cm, nu); get_link<true>(cc, src, g, cn, mu); get_link<true>(d, src, g, c, nu);
        get_lam<true>(Nb, src, g, cm, nu); get_lam<true>(Nc, src, g, cn, mu); get_lam<true>(Nd, src, g, c, nu);
        mmx<false, true>(t1, b, cc);           // b c^+
        mmx<false, true>(X, t1, d);            // X = b c^+ d^+
        mmx<false, false>(t2, X, Na);
        mmx<false, false>(t3, Nb, X);
#pragma unroll
        for (int e = 0; e < 9; e++) t2[e] = mk(t2[e].re + t3[e].re, t2[e].im + t3[e].im);
        mmx<false, false>(t3, a, t2);          // a (X Na + Nb X)
#pragma unroll
        for (int e = 0; e < 9; e++) acc[e] = mk(acc[e].re + t3[e].re, acc[e].im + t3[e].im);
        mmx<false, false>(t2, a, X);
        dag9(t2);                              // (a X)^+ = d c b^+ a^+
        mmx<false, false>(t3, Nd, t2);
#pragma unroll
        for (int e = 0; e < 9; e++) acc[e] = mk(acc[e].re - t3[e].re, acc[e].im - t3[e].im);
        mmx<false, false>(t2, a, t1);
        dag9(t2);                              // (a b c^+)^+ = c b^+ a^+
        mmx<false, false>(t3, Nc, t2);
        mmx<false, false>(t2, d, t3);
#pragma unroll
        for (int e = 0; e < 9; e++) acc[e] = mk(acc[e].re - t2[e].re, acc[e].im - t2[e].im);
        // the plaquette (n - nu; mu, nu), this link as c: a2 = U_mu(m), b2 = U_nu(m + mu), d2 = U_nu(m), m = n - nu
        int m[4] = {c[0], c[1], c[2], c[3]};
        m[nu] -= 1;
        int mm[4] = {m[0], m[1], m[2], m[3]};
        mm[mu] += 1;
        get_link<true>(b, src, g, mm, nu); get_link<true>(cc, src, g, m, mu); get_link<true>(d, src, g, m, nu);       // b2, a2, d2
        get_lam<true>(Nb, src, g, mm, nu); get_lam<true>(Nc, src, g, m, mu); get_lam<true>(Nd, src, g, m, nu);        // Nb2, Na2, Nd2
        mmx<false, false>(t1, cc, b);          // a2 b2
        dag9(t1);                              // R = b2^+ a2^+
        mmx<false, false>(t2, Nd, d);
        mmx<false, false>(t3, d, Na);          // Nc of that plaquette is this link's own N
#pragma unroll
        for (int e = 0; e < 9; e++) t2[e] = mk(t2[e].re + t3[e].re, t2[e].im + t3[e].im);
        mmx<false, false>(t3, t1, t2);
        mmx<false, false>(t2, a, t3);          // c R (Nd2 d2 + d2 Nc)
#pragma unroll
        for (int e = 0; e < 9; e++) acc[e] = mk(acc[e].re + t2[e].re, acc[e].im + t2[e].im);
        mmx<false, false>(t1, Nc, cc);         // Na2 a2
        mmx<false, false>(t2, cc, Nb);         // a2 Nb2
#pragma unroll
        for (int e = 0; e < 9; e++) t1[e] = mk(t1[e].re + t2[e].re, t1[e].im + t2[e].im);
        mmx<false, false>(t2, t1, b);          // (Na2 a2 + a2 Nb2) b2
        mmx<false, true>(t1, t2, a);           // ... c^+
        mmx<true, false>(t2, d, t1);           // d2^+ (...)
#pragma unroll
        for (int e = 0; e < 9; e++) acc[e] = mk(acc[e].re - t2[e].re, acc[e].im - t2[e].im);
    }
    const int Gs = glink_stride(g);
    double2* o = G + glink_off(g, p, mu, i);
#pragma unroll
    for (int e = 0; e < 9; e++) {
        const cd v = ld(o + (size_t)e * Gs);
        st(o + (size_t)e * Gs, mk(v.re - rho * acc[e].re, v.im - rho * acc[e].im));
    }
}

// G += -rho * (the 24 loop terms) with U and N (planes 0..3 of the Lambda-shaped buffer lamN) taken from the halo-extended block: collective on RCCL ranks
int stout_gather_ext(lqcd_ctx_s* c, const lqcd_gauge_s* U, const double2* lamN, lqcd_gauge_s* G, double rho) {
    ARGCHK(c->local_peers.empty(), "stout back-propagation: not available on an in-process PE grid");
    ForceSrc src;
    src.U = U->data; src.lam = lamN; src.ext = nullptr;
    src.eg = ExtGeom();
    LQCHK(clover_ext_build(c, U, lamN, src.eg));
    src.ext = c->clover_ext;
    hipLaunchKernelGGL(stout_gather_ext_kernel, dim3(2 * c->geom.nch), dim3(256), 0, c->stream, c->geom, src, G->data, rho);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
// where the N matrices of a partitioned back-propagation go: direction mu of site (p, i) in plane mu of a Lambda-shaped buffer of the context
double2* stout_lambda_buffer(lqcd_ctx_s* c) {
    if (!c->clover_q[0] && hipMalloc((void**)&c->clover_q[0], clover_lambda_elems(c->geom) * sizeof(double2)) != hipSuccess) return nullptr;
    (void)hipMemsetAsync(c->clover_q[0], 0, clover_lambda_elems(c->geom) * sizeof(double2), c->stream);      // planes 4, 5 travel with the faces: keep them finite
    return c->clover_q[0];
}

// out = (accumulate ? out : 0) + scale * (clover part of "U dS_f/dU"); lam = scratch of clover_lambda_elems() elements
int clover_force(lqcd_ctx_s* c, const lqcd_gauge_s* U, lqcd_gauge_s* out, lqcd_spinor_s* X, lqcd_spinor_s* Y, double2* lam, double kappa,
                 double csw, double scale, int accumulate) {
    SigmaTab tb;
    LQCHK(sigma_table(tb));
    hipLaunchKernelGGL(clover_lambda_kernel, dim3(2 * c->geom.nch), dim3(64), 0, c->stream, c->geom, spinor_block(X, 0), spinor_block(X, 1),
                       spinor_block(Y, 0), spinor_block(Y, 1), lam, tb);
    HIPCHK(hipGetLastError());
    out->version++;
    ForceSrc src;
    src.U = U->data; src.lam = lam; src.ext = nullptr;
    src.eg = ExtGeom();
    if (any_partitioned(c) || c->tun.clover_transport) {
        ARGCHK(c->local_peers.empty(), "clover force: not available on an in-process PE grid");
        LQCHK(clover_ext_build(c, U, lam, src.eg));
        src.ext = c->clover_ext;
        hipLaunchKernelGGL(clover_force_kernel<true>, dim3(2 * c->geom.nch), dim3(256), 0, c->stream, c->geom, src, out->data, kappa * csw / 8.0, scale,
                           accumulate);
    } else {
        hipLaunchKernelGGL(clover_force_kernel<false>, dim3(2 * c->geom.nch), dim3(256), 0, c->stream, c->geom, src, out->data, kappa * csw / 8.0, scale,
                           accumulate);
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

}  // namespace lqcd
