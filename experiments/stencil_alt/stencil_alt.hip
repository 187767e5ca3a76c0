// stencil_alt.hip -- the opt-in Wilson Dslash variants (tunable dslash_variant = 2..8): alternatives to the default direction-split
// kernel of stencil.hip that were built, verified (4-8: bit-identical to variant 1; 2, 3: to the parity tolerance) and MEASURED -- none beats
// variant 1 (LABNOTES.md section 2 has the table; profiles/r0*_variant*.log the numbers).  They stay reachable for A/B runs and as
// the record of what was tried; fp64 only.  Shared device helpers: stencil_common.h.
#include "stencil_common.h"

namespace lqcd {
inline namespace LQCD_PNS {

// ------------------------------------------------------------------------------------------ Wilson, lane-split
// Variant 4 ("lanesplit"): the four directions of a site live in the four 16-lane rows of ONE wavefront -- lane = row * 16 + site,
// row 0 = x, 1 = z, 2 = y, 3 = t -- so a wave owns 16 consecutive checkerboard sites (one x-row at XH = 16) and a workgroup of four
// waves the same 64-site chunk as the other variants.  Every lane does what a lane of the direction-split kernel does (forward +
// backward hop of its direction), but the four partial spinors of a site are combined INSIDE the wave by two
// v_permlane{32,16}_swap reduce-scatter steps (36 swaps + 18 adds per lane) instead of 48 KiB of LDS and a workgroup barrier:
// no LDS, no s_barrier, occupancy is limited by VGPRs only, and lane (row r, site s) ends up with spin row r of site s and
// stores it.  The direction of a lane is data (per-lane projector rows and unit phases), the instruction stream is uniform; the
// t rows simply mask the six spinor loads their projector does not need.  Association of the four-direction sum and of every hop
// is the direction-split kernel's, so the two variants agree bit for bit.  r = 1 only.
__device__ inline void lane_swap32(real& a, real& b) {   // a.upper32 <-> b.lower32 (v_permlane32_swap_b32)
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#ifdef LQCD_F32
    u2v r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
#else
    u2v lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
#endif
}
__device__ inline void lane_swap16(real& a, real& b) {   // odd 16-lane rows of a <-> even 16-lane rows of b (v_permlane16_swap_b32)
    typedef unsigned int u2v __attribute__((ext_vector_type(2)));
#ifdef LQCD_F32
    u2v r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x); b = __uint_as_float(r.y);
#else
    u2v lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    u2v hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    a = __hiloint2double((int)hi.x, (int)lo.x); b = __hiloint2double((int)hi.y, (int)lo.y);
#endif
}

// one hop of a lane whose direction is run-time data: chi_r = [U or U^+] sign (ca psi[a_r] + (pr_r + i pi_r) psi[b_r]),  r = 0, 1
template <bool ADJ, bool R12>
__device__ inline void lane_hop(cd (&chi0)[3], cd (&chi1)[3], const real2* __restrict__ psi, const real2* __restrict__ U, int Us,
                                bool spatial, int a0, int b0, int b1, real ca, real pr0, real pi0, real pr1, real pi1, real sign, bool nt) {
    cd h0[3], h1[3], u[9];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const cd A0 = ld(psi + (size_t)(a0 * 3 + c) * 64), A1 = ld(psi + (size_t)(a0 * 3 + 3 + c) * 64);
        cd B0 = mk(0, 0), B1 = mk(0, 0);
        if (spatial) { B0 = ld(psi + (size_t)(b0 * 3 + c) * 64); B1 = ld(psi + (size_t)(b1 * 3 + c) * 64); }
        h0[c] = mk(ca * A0.re + (pr0 * B0.re - pi0 * B0.im), ca * A0.im + (pr0 * B0.im + pi0 * B0.re));
        h1[c] = mk(ca * A1.re + (pr1 * B1.re - pi1 * B1.im), ca * A1.im + (pr1 * B1.im + pi1 * B1.re));
    }
    if constexpr (R12) load_link12(u, U, nt);
    else { if (nt) load_link_nt(u, U, Us); else load_link(u, U, Us); }
#pragma unroll
    for (int c = 0; c < 3; c++) { h0[c] = sign * h0[c]; h1[c] = sign * h1[c]; }
    su3_mv<ADJ>(chi0, u, h0);
    su3_mv<ADJ>(chi1, u, h1);
}

template <bool DAG, bool R12>
__global__ __launch_bounds__(256) void wilson_lanesplit(KArgs k) {
    __shared__ double red[4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const Geom& g = k.g;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int lr = lane >> 4;                      // lane row = output spin row this lane stores
    const int mu = ((lr & 1) << 1) | (lr >> 1);    // direction of this lane: rows 0,1,2,3 = x,z,y,t  ->  (x + y) + (z + t) like dirsplit
    const int i = chunk * 64 + w * 16 + (lane & 15);
    const bool valid = i < g.Vh;
    constexpr int SF = DAG ? -1 : 1;
    cd cf0[3], cf1[3], cb0[3], cb1[3], xv[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { cf0[c] = cf1[c] = cb0[c] = cb1[c] = xv[c] = mk(0, 0); }
    const bool spatial = mu < 3;
    if (valid) {
        if (k.a != 0.0) {
#pragma unroll
            for (int c = 0; c < 3; c++) xv[c] = ld(k.xin[p] + sp_off(12, i) + (size_t)(3 * lr + c) * 64);
        }
        // neighbours of this lane's direction
        int cc[4];
        cb_to_coords(g, p, i, cc);
        int nf, nb;
        real sf, sb;
        {
            // per-direction geometry selected by VALUE (readfirstlane makes the kernel arguments opaque scalars: a select between
            // loads of neighbouring struct fields would be rewritten into a per-lane indexed load of the by-value struct = scratch)
            const int L0 = __builtin_amdgcn_readfirstlane(g.L[0]), L1 = __builtin_amdgcn_readfirstlane(g.L[1]);
            const int L2 = __builtin_amdgcn_readfirstlane(g.L[2]), L3 = __builtin_amdgcn_readfirstlane(g.L[3]);
            const int XH = __builtin_amdgcn_readfirstlane(g.XH);
            const int q = cc[0] & 1;
            const int s1 = XH, s2 = XH * L1, s3 = s2 * L2;
            const int stride = mu == 1 ? s1 : (mu == 2 ? s2 : s3);
            const int Lm = mu == 0 ? L0 : (mu == 1 ? L1 : (mu == 2 ? L2 : L3));
            const int cm = mu == 0 ? cc[0] : (mu == 1 ? cc[1] : (mu == 2 ? cc[2] : cc[3]));
            const bool wf = cm == Lm - 1, wb = cm == 0;
            if (mu == 0) {
                nf = q ? (wf ? i - (XH - 1) : i + 1) : i;
                nb = q ? i : (wb ? i + (XH - 1) : i - 1);
            } else {
                nf = wf ? i - (Lm - 1) * stride : i + stride;
                nb = wb ? i + (Lm - 1) * stride : i - stride;
            }
            // sign of a wrapping hop: 0 = off-rank (partitioned direction), else the boundary condition (an integer sign, lqcd_op_create)
            const int f0 = __builtin_amdgcn_readfirstlane(g.part[0] ? 0 : (int)g.bc_fwd[0]), f1 = __builtin_amdgcn_readfirstlane(g.part[1] ? 0 : (int)g.bc_fwd[1]);
            const int f2 = __builtin_amdgcn_readfirstlane(g.part[2] ? 0 : (int)g.bc_fwd[2]), f3 = __builtin_amdgcn_readfirstlane(g.part[3] ? 0 : (int)g.bc_fwd[3]);
            const int r0 = __builtin_amdgcn_readfirstlane(g.part[0] ? 0 : (int)g.bc_bwd[0]), r1 = __builtin_amdgcn_readfirstlane(g.part[1] ? 0 : (int)g.bc_bwd[1]);
            const int r2 = __builtin_amdgcn_readfirstlane(g.part[2] ? 0 : (int)g.bc_bwd[2]), r3 = __builtin_amdgcn_readfirstlane(g.part[3] ? 0 : (int)g.bc_bwd[3]);
            const int bfi = mu == 0 ? f0 : (mu == 1 ? f1 : (mu == 2 ? f2 : f3));
            const int bbi = mu == 0 ? r0 : (mu == 1 ? r1 : (mu == 2 ? r2 : r3));
            sf = wf ? real(bfi) : real(1.0);
            sb = wb ? real(bbi) : real(1.0);
        }
        // projector data of the lane (tables PERM / GK, lqcd_internal.h): partner rows b0, b1 of rows 0, 1 and the powers of i
        const int b0 = mu == 2 ? 2 : 3, b1 = mu == 2 ? 3 : 2;
        const int k0 = mu == 1 ? 2 : 3, k1 = mu == 0 ? 3 : (mu == 1 ? 0 : 1);    // GK[mu][0], GK[mu][1]
        auto unit = [](int kk, real& pr, real& pi) {   // i^kk
            kk &= 3;
            pr = kk == 0 ? real(1) : (kk == 2 ? real(-1) : real(0));
            pi = kk == 1 ? real(1) : (kk == 3 ? real(-1) : real(0));
        };
        const real ca = spatial ? real(1) : real(2);
        const real2* __restrict__ psi = k.in[1 - p];
        const int Us = glink_stride(g);
        const real2* __restrict__ Uf = R12 ? k.gauge12 + glink12_off(g, p, mu, i) : k.gauge + glink_off(g, p, mu, i);
        const real2* __restrict__ Ub = R12 ? k.gauge12 + glink12_off(g, 1 - p, mu, nb) : k.gauge + glink_off(g, 1 - p, mu, nb);
        if (sf != 0.0) {   // forward hop: (1 - SF gamma_mu) U psi(n + mu)
            real pr0 = 0, pi0 = 0, pr1 = 0, pi1 = 0;
            if (spatial) { unit(k0 + (SF > 0 ? 2 : 0), pr0, pi0); unit(k1 + (SF > 0 ? 2 : 0), pr1, pi1); }
            lane_hop<false, R12>(cf0, cf1, psi + sp_off(12, nf), Uf, Us, spatial, spatial ? 0 : (SF > 0 ? 2 : 0), b0, b1, ca, pr0, pi0, pr1, pi1,
                                 sf, (k.nt & 2) != 0);
        }
        if (sb != 0.0) {   // backward hop: (1 + SF gamma_mu) U^+(n - mu) psi(n - mu)
            real pr0 = 0, pi0 = 0, pr1 = 0, pi1 = 0;
            if (spatial) { unit(k0 + (SF > 0 ? 0 : 2), pr0, pi0); unit(k1 + (SF > 0 ? 0 : 2), pr1, pi1); }
            lane_hop<true, R12>(cb0, cb1, psi + sp_off(12, nb), Ub, Us, spatial, spatial ? 0 : (SF > 0 ? 0 : 2), b0, b1, ca, pr0, pi0, pr1, pi1,
                                sb, (k.nt & 1) != 0);
        }
    }
    // the lane's partial spinor, spin rows 0..3 (3 colours each).  Spatial directions: rows 0,1 = chi_f + chi_b, rows b0,b1 = q (chi_f - chi_b)
    // with q = i^(-GK + (SF > 0 ? 2 : 0)); t: the hop whose projector keeps rows 0,1 goes there, the other one to rows 2,3.
    cd A[6], B[6];
    {
        const int mu_ = mu;
        const int k0 = mu_ == 1 ? 2 : 3, k1 = mu_ == 0 ? 3 : (mu_ == 1 ? 0 : 1);
        const int e0 = (-k0 + (SF > 0 ? 2 : 0)) & 3, e1 = (-k1 + (SF > 0 ? 2 : 0)) & 3;
        const real qr0 = e0 == 0 ? real(1) : (e0 == 2 ? real(-1) : real(0)), qi0 = e0 == 1 ? real(1) : (e0 == 3 ? real(-1) : real(0));
        const real qr1 = e1 == 0 ? real(1) : (e1 == 2 ? real(-1) : real(0)), qi1 = e1 == 1 ? real(1) : (e1 == 3 ? real(-1) : real(0));
        const bool swap = mu_ < 2;          // x, y: row 3 <- chi0 part, row 2 <- chi1 part;  z (and t): row 2 <- chi0, row 3 <- chi1
        constexpr bool F_LOW = SF < 0;      // t: forward hop keeps rows (SF > 0 ? 2 : 0)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const cd f0 = cf0[c], f1 = cf1[c], g0 = cb0[c], g1 = cb1[c];
            const cd sum0 = f0 + g0, sum1 = f1 + g1, dif0 = f0 - g0, dif1 = f1 - g1;
            A[c] = spatial ? sum0 : (F_LOW ? f0 : g0);
            A[3 + c] = spatial ? sum1 : (F_LOW ? f1 : g1);
            const cd E0 = spatial ? mk(qr0 * dif0.re - qi0 * dif0.im, qr0 * dif0.im + qi0 * dif0.re) : (F_LOW ? g0 : f0);
            const cd E1 = spatial ? mk(qr1 * dif1.re - qi1 * dif1.im, qr1 * dif1.im + qi1 * dif1.re) : (F_LOW ? g1 : f1);
            B[c] = swap ? E1 : E0;
            B[3 + c] = swap ? E0 : E1;
        }
    }
    // reduce-scatter over the four lane rows: rows {0,1} keep spin rows 0,1 and rows {2,3} spin rows 2,3, then each row its own
#pragma unroll
    for (int j = 0; j < 6; j++) {
        lane_swap32(A[j].re, B[j].re);
        lane_swap32(A[j].im, B[j].im);
        A[j] = A[j] + B[j];
    }
    cd F[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        lane_swap16(A[c].re, A[3 + c].re);
        lane_swap16(A[c].im, A[3 + c].im);
        F[c] = A[c] + A[3 + c];
    }
    real nrm = 0.0;
    if (valid) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            cd v = k.b * F[c];
            v = mk(fma(k.a, xv[c].re, v.re), fma(k.a, xv[c].im, v.im));
            emit(k, p, (size_t)(3 * lr + c) * 64 + sp_off(12, i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}


// ------------------------------------------------------------------------------------------ Wilson, direction-split, 4 workgroups per CU
// Variant 5: the direction-split kernel with the footprint of FOUR resident workgroups per CU instead of three: (i) a wave keeps
// the spin row it stores in registers and leaves only the nine components bound for the other three waves in LDS (36 KiB per
// workgroup, 4 x 36 <= 160 KiB); (ii) each hop is carried as the two colour-multiplied projector rows (6 complex) and expanded
// to spin rows only on the way to LDS, which removes the 12-component accumulator from the live set while the loads of the
// backward hop are in flight (<= 128 VGPRs, __launch_bounds__(256, 4)).  Same arithmetic and summation order as variant 1.
template <int MU, int S, bool ADJ, bool R12>
__device__ inline void wilson_hop_chi(cd (&chi0)[3], cd (&chi1)[3], const real2* __restrict__ psi, const real2* __restrict__ U, int Vh, int Us,
                                      real sign, bool nt) {
    cd h0[3], h1[3], u[9];
    project<MU, S>(h0, h1, psi, Vh);
    if constexpr (R12) load_link12(u, U, nt);
    else { if (nt) load_link_nt(u, U, Us); else load_link(u, U, Us); }
#pragma unroll
    for (int c = 0; c < 3; c++) { h0[c] = sign * h0[c]; h1[c] = sign * h1[c]; }
    su3_mv<ADJ>(chi0, u, h0);
    su3_mv<ADJ>(chi1, u, h1);
}

// spin row ROW (3 colours) of  reconstruct<MU, SF>(chi_f) + reconstruct<MU, -SF>(chi_b)  in the accumulation order of wilson_hop
template <int MU, int SF, int ROW>
__device__ inline void recon_row(cd (&out)[3], const cd (&f0)[3], const cd (&f1)[3], const cd (&b0)[3], const cd (&b1)[3]) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        cd a = mk(0.0, 0.0);
        if constexpr (MU < 3) {
            constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
            constexpr int kf0 = -GK[MU][0] + (SF > 0 ? 2 : 0) + 8, kf1 = -GK[MU][1] + (SF > 0 ? 2 : 0) + 8;
            constexpr int kb0 = -GK[MU][0] + (SF > 0 ? 0 : 2) + 8, kb1 = -GK[MU][1] + (SF > 0 ? 0 : 2) + 8;
            if constexpr (ROW == 0) a = (a + f0[c]) + b0[c];
            else if constexpr (ROW == 1) a = (a + f1[c]) + b1[c];
            else if constexpr (ROW == p0) a = (a + mul_ipow<kf0>(f0[c])) + mul_ipow<kb0>(b0[c]);
            else a = (a + mul_ipow<kf1>(f1[c])) + mul_ipow<kb1>(b1[c]);
        } else {
            constexpr int bf = SF > 0 ? 2 : 0, bb = SF > 0 ? 0 : 2;   // forward hop S = SF keeps rows bf, bf+1; backward hop rows bb, bb+1
            if constexpr (ROW == bf) a = a + f0[c];
            else if constexpr (ROW == bf + 1) a = a + f1[c];
            else if constexpr (ROW == bb) a = a + b0[c];
            else a = a + b1[c];
        }
        out[c] = a;
    }
}

template <int MU, bool DAG, bool R12>
__device__ inline void dirsplit4_body(const KArgs& k, int p, int i, int lane, bool valid, real2 (*part)[3][3][64], cd (&own)[3]) {
    constexpr int SF = DAG ? -1 : 1;
    cd f0[3], f1[3], b0[3], b1[3];
#pragma unroll
    for (int c = 0; c < 3; c++) { f0[c] = f1[c] = b0[c] = b1[c] = mk(0.0, 0.0); }
    if (valid) {
        Nbr n;
        int c[4];
        neighbours(k.g, p, i, n, c);
        const int Vh = sp_stride(k.g);
        const real2* __restrict__ psi = k.in[1 - p];
        const real2* __restrict__ Uf = R12 ? k.gauge12 + glink12_off(k.g, p, MU, i) : k.gauge + glink_off(k.g, p, MU, i);
        const real2* __restrict__ Ub = R12 ? k.gauge12 + glink12_off(k.g, 1 - p, MU, n.bwd[MU]) : k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]);
        const int Us = glink_stride(k.g);
        if (n.sf[MU] != 0.0) wilson_hop_chi<MU, SF, false, R12>(f0, f1, psi + sp_off(12, n.fwd[MU]), Uf, Vh, Us, n.sf[MU], (k.nt & 2) != 0);
        if (n.sb[MU] != 0.0) wilson_hop_chi<MU, -SF, true, R12>(b0, b1, psi + sp_off(12, n.bwd[MU]), Ub, Vh, Us, n.sb[MU], (k.nt & 1) != 0);
    }
    // spin row r goes to wave r: slot (MU < r ? MU : MU - 1) of its three source slots; the own row stays in registers
    cd row[3];
#define LQ_ROW(R)                                                                      \
    recon_row<MU, SF, R>(row, f0, f1, b0, b1);                                         \
    if constexpr (R == MU) { own[0] = row[0]; own[1] = row[1]; own[2] = row[2]; }      \
    else {                                                                             \
        _Pragma("unroll") for (int c = 0; c < 3; c++) part[R][MU < R ? MU : MU - 1][c][lane] = mk2(row[c].re, row[c].im); \
    }
    LQ_ROW(0) LQ_ROW(1) LQ_ROW(2) LQ_ROW(3)
#undef LQ_ROW
}

template <bool DAG, bool R12>
__global__ __launch_bounds__(256, 4) void wilson_dirsplit4(KArgs k) {
    __shared__ real2 part[4][3][3][64];  // [destination wave = spin row][source slot][colour][lane]: 36 KiB
    __shared__ double red[4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    if (valid && k.a != 0.0) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ld(k.xin[p] + sp_off(12, i) + (size_t)(3 * w + cc) * Vh);
    }
    cd own[3];
    switch (w) {
    case 0: dirsplit4_body<0, DAG, R12>(k, p, i, lane, valid, part, own); break;
    case 1: dirsplit4_body<1, DAG, R12>(k, p, i, lane, valid, part, own); break;
    case 2: dirsplit4_body<2, DAG, R12>(k, p, i, lane, valid, part, own); break;
    default: dirsplit4_body<3, DAG, R12>(k, p, i, lane, valid, part, own); break;
    }
    __syncthreads();
    real nrm = 0.0;
    if (valid) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            // (s0 + s1) + (s2 + s3) with s_w the partial of wave w: the own one from registers, the others from their slots
            cd sv[4];
#pragma unroll
            for (int src = 0; src < 4; src++) {
                if (src == w) sv[src] = own[cc];
                else { const real2 t = part[w][src < w ? src : src - 1][cc][lane]; sv[src] = mk(t.x, t.y); }
            }
            cd s = mk((sv[0].re + sv[1].re) + (sv[2].re + sv[3].re), (sv[0].im + sv[1].im) + (sv[2].im + sv[3].im));
            cd v = k.b * s;
            v = mk(fma(k.a, xv[cc].re, v.re), fma(k.a, xv[cc].im, v.im));
            emit(k, p, (size_t)(3 * w + cc) * Vh + sp_off(12, i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}


// ------------------------------------------------------------------------------------------ Wilson, direction-split, both parities of a chunk in one workgroup
// Variant 7: the even and the odd sites of a 64-site chunk in ONE 512-thread workgroup (waves 0-3: parity 0, waves 4-7: parity 1, each
// group exactly variant 5).  Every x link and three quarters of the y links are used forward by one group and backward by the other, and
// each group's x / y neighbour spinors are the other group's centre chunk: with both in the same workgroup those second uses coincide in
// time on one CU instead of depending on how two workgroups happen to be scheduled.  72 KiB of LDS, <= 128 VGPRs: two workgroups = 16 waves
// per CU.  Full-lattice applications with 12-real links only (the 18-real instance needs more than 128 registers); block partials are
// written at the indices variant 1 uses, so solver iterates do not change.
template <bool DAG, bool R12>
__global__ __launch_bounds__(512, 4) void wilson_pair4(KArgs k) {
    __shared__ real2 part[2][4][3][3][64];  // 72 KiB
    __shared__ double red[8];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = wave & 3;
    int p = wave >> 2, chunk, vb;
    if (k.remap == 2 && k.cps > 0) {        // the two virtual blocks of variant 1's map that hold (chunk, 0) and (chunk, 1)
        vb = (blockIdx.x & 7) + 8 * (p + 2 * (blockIdx.x >> 3));
        map_block_v(k, vb, chunk, p);
    } else {
        chunk = blockIdx.x;
        vb = 2 * blockIdx.x + p;
    }
    const int Vh = sp_stride(k.g);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    if (valid && k.a != 0.0) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ld(k.xin[p] + sp_off(12, i) + (size_t)(3 * w + cc) * Vh);
    }
    cd own[3];
    switch (w) {
    case 0: dirsplit4_body<0, DAG, R12>(k, p, i, lane, valid, part[p], own); break;
    case 1: dirsplit4_body<1, DAG, R12>(k, p, i, lane, valid, part[p], own); break;
    case 2: dirsplit4_body<2, DAG, R12>(k, p, i, lane, valid, part[p], own); break;
    default: dirsplit4_body<3, DAG, R12>(k, p, i, lane, valid, part[p], own); break;
    }
    __syncthreads();
    real nrm = 0.0;
    if (valid) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            cd sv[4];
#pragma unroll
            for (int src = 0; src < 4; src++) {
                if (src == w) sv[src] = own[cc];
                else { const real2 t = part[p][w][src < w ? src : src - 1][cc][lane]; sv[src] = mk(t.x, t.y); }
            }
            cd s = mk((sv[0].re + sv[1].re) + (sv[2].re + sv[3].re), (sv[0].im + sv[1].im) + (sv[2].im + sv[3].im));
            cd v = k.b * s;
            v = mk(fma(k.a, xv[cc].re, v.re), fma(k.a, xv[cc].im, v.im));
            emit(k, p, (size_t)(3 * w + cc) * Vh + sp_off(12, i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[wave] = nrm;
        __syncthreads();
        if ((threadIdx.x & 255) == 0) k.norm_partial[vb] = (red[4 * p] + red[4 * p + 1]) + (red[4 * p + 2] + red[4 * p + 3]);
    }
}

// ------------------------------------------------------------------------------------------ Wilson, hop-split
// Variant 2 ("hopsplit"): 8 waves per 64 sites, one per hop (direction x sign).  Every wave issues its 21 loads
// (12 spinor + 9 link; 15 for the t hops) as ONE burst -- a workgroup has a single memory round trip -- and leaves the
// colour-multiplied half spinor (6 complex) in LDS; six waves then reconstruct two output components each.
template <int MU, int S, int ROW>
struct Recon {  // contribution of hop (MU, S) to spin row ROW: src half-spinor row (or -1) and the power of i to apply
    static constexpr int src = (MU < 3) ? (ROW < 2 ? ROW : (PERM[MU < 3 ? MU : 0][0] == ROW ? 0 : 1))
                                        : ((ROW == (S > 0 ? 2 : 0)) ? 0 : (ROW == (S > 0 ? 3 : 1)) ? 1 : -1);
    static constexpr int kpow = (MU < 3 && ROW >= 2) ? ((-GK[MU < 3 ? MU : 0][src < 0 ? 0 : src] + (S > 0 ? 2 : 0) + 8) % 4) : 0;
};

template <int MU, int S, int ROW>
__device__ inline void add_hop(cd& sum, const real2 (*half)[6][64], int h, int c, int lane) {
    constexpr int src = Recon<MU, S, ROW>::src;
    if constexpr (src >= 0) {
        const real2 v = half[h][src * 3 + c][lane];
        sum = sum + mul_ipow<Recon<MU, S, ROW>::kpow>(mk(v.x, v.y));
    }
}

template <int J, bool DAG>
__device__ inline cd combine_comp(const real2 (*half)[6][64], int lane) {
    constexpr int ROW = J / 3, c = J % 3, SF = DAG ? -1 : 1;
    cd s0 = mk(0, 0), s1 = mk(0, 0);
    add_hop<0, SF, ROW>(s0, half, 0, c, lane); add_hop<0, -SF, ROW>(s1, half, 1, c, lane);
    add_hop<1, SF, ROW>(s0, half, 2, c, lane); add_hop<1, -SF, ROW>(s1, half, 3, c, lane);
    add_hop<2, SF, ROW>(s0, half, 4, c, lane); add_hop<2, -SF, ROW>(s1, half, 5, c, lane);
    add_hop<3, SF, ROW>(s0, half, 6, c, lane); add_hop<3, -SF, ROW>(s1, half, 7, c, lane);
    return s0 + s1;
}



// ------------------------------------------------------------------------------------------ Wilson, direction-split, neighbour spinors through LDS
// Variant 6: the x and y neighbours of a site of chunk c (64 consecutive checkerboard sites = 64 / XH x-rows) are sites of THE SAME chunk
// of the other parity -- all of them in x (the row wraps onto itself), all but one boundary row per direction in y.  The four waves load
// that chunk once (3 components each, 12 loads per workgroup), leave it in 12 KiB of LDS, and the x and y waves take both their hops from
// there (ds_read_b128 with the neighbour's lane index: the LDS moves 256 B/clk where the texture path moves 64); a lane of the y wave
// whose neighbour lies in the next / previous chunk has loaded that ONE spinor into registers beforehand.  Per workgroup 180 -> 150
// 16-B/lane global loads (x: 45 -> 24, y: 45 -> 36 of which 12 quarter-masked, z, t: +3 each).  Schedule: every wave issues its staging
// loads FIRST, then the loads that do not depend on the staged data (x, y: both links [+ the edge spinor]; z, t: link and spinor of the
// forward hop), writes its staged components and arrives at a raw s_barrier behind an lgkmcnt(0) only -- the other loads stay in flight
// across the barrier, nobody pays a second dependent memory round trip.  Partial sums as in variant 5 (own spin row in registers, 36 KiB);
// 48 KiB of LDS in ONE array, three workgroups per CU.  Needs 64 % XH == 0 and at least two rows per chunk (XH <= 32); the launcher
// falls back to variant 1 otherwise.  Same arithmetic and summation order as variant 1: bit-identical results.
__device__ inline void lds_stage_and_barrier(real2 (*nbr)[64], const cd (&stg)[3], int w, int lane) {
    __builtin_amdgcn_sched_barrier(0);     // arithmetic on the loads issued above must not be hoisted in front of the barrier (it would wait for them there)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) nbr[3 * w + cc][lane] = mk2(stg[cc].re, stg[cc].im);
    // LDS writes complete, then the workgroup barrier; global loads issued above stay in flight (no vmcnt wait here)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------------------------------ Wilson, direction-split, both hops in flight
// Variant 8: variant 1 issues the loads of the forward hop, waits, computes, and only then issues the loads of the backward hop -- the
// `if (sign != 0)` around each hop is control flow, and the compiler may not move a load across it -- so every wave pays TWO dependent
// memory round trips.  On an unpartitioned lattice no hop is ever skipped: this variant has no branch in the body, all 36 (t: 24) loads
// of the direction are issued back to back and the arithmetic follows (same operations in the same order as variant 1: bit-identical
// results).  More registers live at the peak (both half-sets of operands).
template <int MU, bool DAG, bool R12, bool NTB>
__device__ inline void dirsplit_hops_both(cd (&acc)[12], const KArgs& k, int p, int i) {
    Nbr n;
    int c[4];
    neighbours(k.g, p, i, n, c);
    const real2* __restrict__ psi = k.in[1 - p];
    const real2* __restrict__ Uf = R12 ? k.gauge12 + gl12_off(k.g, p, MU, i) : k.gauge + glink_off(k.g, p, MU, i);
    const real2* __restrict__ Ub = R12 ? k.gauge12 + gl12_off(k.g, 1 - p, MU, n.bwd[MU]) : k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]);
    const int Us = glink_stride(k.g);
    constexpr int SF = DAG ? -1 : 1;
    constexpr int NS = MU == 3 ? 6 : 12;                       // t: only the two rows the projector keeps
    constexpr int FF = MU == 3 ? (SF > 0 ? 6 : 0) : 0;         // first component of the forward / backward hop's rows
    constexpr int FB = MU == 3 ? (SF > 0 ? 0 : 6) : 0;
    cd sf[NS], sb[NS], uf[9], ub[9];
    load_comps12<FF, NS, false>(sf, psi + sp12_off(n.fwd[MU]));
    load_link_any<R12, false>(uf, Uf, Us);
    load_comps12<FB, NS, false>(sb, psi + sp12_off(n.bwd[MU]));
    load_link_any<R12, NTB>(ub, Ub, Us);
    cd chi0[3], chi1[3];
    finish_link<R12>(uf);
    hop_from_regs<MU, SF, false>(chi0, chi1, sf, uf, n.sf[MU]);
    reconstruct<MU, SF>(acc, chi0, chi1);
    finish_link<R12>(ub);
    hop_from_regs<MU, -SF, true>(chi0, chi1, sb, ub, n.sb[MU]);
    reconstruct<MU, -SF>(acc, chi0, chi1);
}
#ifndef LQCD_V8_OCC
#ifdef LQCD_F32
#define LQCD_V8_OCC 3
#else
#define LQCD_V8_OCC 2
#endif
#endif
template <bool DAG, bool R12, bool NTB>
__global__ __launch_bounds__(256, LQCD_V8_OCC) void wilson_dirsplit_both(KArgs k) {
    __shared__ real2 part[4][12][64];  // 48 KiB (fp32: 24)
    __shared__ double red[4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    const int ic = valid ? i : 0;          // no control flow around the loads: lanes past the end work on site 0 and do not store
    cd acc[12];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
    cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    cd rv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    if (k.upd_scal) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ld(k.upd[p] + sp12_off(ic) + co12(3 * w + cc));
    }
    if (k.a != 0.0) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ld(k.xin[p] + sp12_off(ic) + co12(3 * w + cc));
    }
    switch (w) {
    case 0: dirsplit_hops_both<0, DAG, R12, NTB>(acc, k, p, ic); break;
    case 1: dirsplit_hops_both<1, DAG, R12, NTB>(acc, k, p, ic); break;
    case 2: dirsplit_hops_both<2, DAG, R12, NTB>(acc, k, p, ic); break;
    default: dirsplit_hops_both<3, DAG, R12, NTB>(acc, k, p, ic); break;
    }
#pragma unroll
    for (int j = 0; j < 12; j++) part[w][j][lane] = mk2(acc[j].re, acc[j].im);
    __syncthreads();
    real nrm = 0.0;
    if (valid) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const int j = 3 * w + cc;
            const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
            cd s = mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
            cd v = k.b * s;
            v = mk(fma(k.a, xv[cc].re, v.re), fma(k.a, xv[cc].im, v.im));
            emit_pre(k, p, co12(j) + sp12_off(i), v, nrm, al_upd, rv[cc]);
        }
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <int MU, bool DAG, bool R12, bool NTB>
__device__ inline void dslds_body(const KArgs& k, int p, int chunk, int ic, int lane, real2* lds, cd (&own)[3], const cd (&stg)[3], cd (&xv)[3], int w) {
    // ic: the lane's site, clamped to a valid one -- between the staging loads and the barrier there is NO control flow (a branch around
    // a load makes the static vmcnt of the staging data wait for everything); skipped hops (off-rank neighbours) are multiplied by sign 0
    constexpr int SF = DAG ? -1 : 1;
    real2 (*part)[3][3][64] = reinterpret_cast<real2 (*)[3][3][64]>(lds);
    real2 (*nbr)[64] = reinterpret_cast<real2 (*)[64]>(lds + 4 * 3 * 3 * 64);
    cd f0[3], f1[3], b0[3], b1[3];
    Nbr n;
    int c[4];
    neighbours(k.g, p, ic, n, c);
    const int Vh = sp_stride(k.g);
    const int Us = glink_stride(k.g);
    const real2* __restrict__ psi = k.in[1 - p];
    const real2* __restrict__ Uf = R12 ? k.gauge12 + glink12_off(k.g, p, MU, ic) : k.gauge + glink_off(k.g, p, MU, ic);
    const real2* __restrict__ Ub = R12 ? k.gauge12 + glink12_off(k.g, 1 - p, MU, n.bwd[MU]) : k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]);
    auto load_xin = [&]() {       // the diagonal term's components: issued right behind the barrier (an a == 0 hop-only call has no xin)
        if (k.a != 0.0) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) xv[cc] = ld(k.xin[p] + sp_off(12, ic) + (size_t)(3 * w + cc) * Vh);
        }
    };
    if constexpr (MU < 2) {
        cd uf[9], ub[9];
        const bool f_in = (n.fwd[MU] >> 6) == chunk, b_in = (n.bwd[MU] >> 6) == chunk;
        load_link_raw<R12, false>(uf, Uf, Us);
        load_link_raw<R12, NTB>(ub, Ub, Us);
        lds_stage_and_barrier(nbr, stg, w, lane);
        load_xin();
        // spinor component j of the neighbour: from the staged chunk, or (y: the boundary row of the chunk) from the neighbouring chunk in memory
        auto hop = [&](auto adj, auto sgn, cd (&c0)[3], cd (&c1)[3], cd (&u)[9], int nb_, bool in_lds, real sign) {
            constexpr bool ADJ = decltype(adj)::value;
            constexpr int S = decltype(sgn)::value;
            constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
            constexpr int k0 = GK[MU][0] + (S > 0 ? 2 : 0), k1 = GK[MU][1] + (S > 0 ? 2 : 0);
            const int l = nb_ & 63;
            const real2* __restrict__ e = psi + sp_off(12, nb_);
            auto get = [&](int j) -> cd {
                if (MU == 1 && !in_lds) return ld(e + (size_t)j * Vh);
                const real2 t = nbr[j][l];
                return mk(t.x, t.y);
            };
            cd h0[3], h1[3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                h0[cc] = get(cc) + mul_ipow<k0>(get(p0 * 3 + cc));
                h1[cc] = get(3 + cc) + mul_ipow<k1>(get(p1 * 3 + cc));
            }
            finish_link<R12>(u);
#pragma unroll
            for (int cc = 0; cc < 3; cc++) { h0[cc] = sign * h0[cc]; h1[cc] = sign * h1[cc]; }
            su3_mv<ADJ>(c0, u, h0);
            su3_mv<ADJ>(c1, u, h1);
        };
        pin_after_barrier(uf, R12 ? 6 : 9);
        pin_after_barrier(ub, R12 ? 6 : 9);
        hop(std::false_type{}, std::integral_constant<int, SF>{}, f0, f1, uf, n.fwd[MU], f_in, n.sf[MU]);
        if constexpr (MU == 1) __builtin_amdgcn_sched_barrier(0);
        hop(std::true_type{}, std::integral_constant<int, -SF>{}, b0, b1, ub, n.bwd[MU], b_in, n.sb[MU]);
    } else {
        constexpr int NS = MU < 3 ? 12 : 6;
        constexpr int basef = (MU == 3 && SF > 0) ? 2 : 0;      // rows the t projector of the forward hop keeps
        cd uf[9], sp[NS];
        load_link_raw<R12, false>(uf, Uf, Us);
        {
            const real2* __restrict__ e = psi + sp_off(12, n.fwd[MU]);
#pragma unroll
            for (int j = 0; j < NS; j++) sp[j] = ld(e + (size_t)(basef * 3 + j) * Vh);
        }
        lds_stage_and_barrier(nbr, stg, w, lane);
        load_xin();
        pin_after_barrier(uf, R12 ? 6 : 9);
        pin_after_barrier(sp);
        finish_link<R12>(uf);
        hop_from_regs<MU, SF, false>(f0, f1, sp, uf, n.sf[MU]);
        // The backward hop's 21 loads must not be issued while the forward hop's 21 registers are still live (register budget of 3 waves
        // per SIMD): an empty asm makes the backward neighbour index depend on the forward result.
        int nbw = n.bwd[MU];
        asm volatile("" : "+v"(nbw), "+v"(f0[0].re), "+v"(f0[0].im), "+v"(f0[1].re), "+v"(f0[1].im), "+v"(f0[2].re), "+v"(f0[2].im),
                          "+v"(f1[0].re), "+v"(f1[0].im), "+v"(f1[1].re), "+v"(f1[1].im), "+v"(f1[2].re), "+v"(f1[2].im));
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { b0[cc] = b1[cc] = mk(0.0, 0.0); }
        const real2* __restrict__ Ub2 = R12 ? k.gauge12 + glink12_off(k.g, 1 - p, MU, nbw) : k.gauge + glink_off(k.g, 1 - p, MU, nbw);
        if (n.sb[MU] != 0.0) wilson_hop_chi<MU, -SF, true, R12>(b0, b1, psi + sp_off(12, nbw), Ub2, Vh, Us, n.sb[MU], NTB);
    }
    cd row[3];
#define LQ_ROW(R)                                                                      \
    recon_row<MU, SF, R>(row, f0, f1, b0, b1);                                         \
    if constexpr (R == MU) { own[0] = row[0]; own[1] = row[1]; own[2] = row[2]; }      \
    else {                                                                             \
        _Pragma("unroll") for (int cc = 0; cc < 3; cc++) part[R][MU < R ? MU : MU - 1][cc][lane] = mk2(row[cc].re, row[cc].im); \
    }
    LQ_ROW(0) LQ_ROW(1) LQ_ROW(2) LQ_ROW(3)
#undef LQ_ROW
}

#ifndef LQCD_V6_OCC
#define LQCD_V6_OCC 3
#endif
template <bool DAG, bool R12, bool NTB>
__global__ __launch_bounds__(256, LQCD_V6_OCC) void wilson_dirsplit_lds(KArgs k) {
    __shared__ real2 lds[4 * 3 * 3 * 64 + 12 * 64 + 4];   // partial rows (36 KiB) | staged neighbour chunk (12 KiB) | norm partials
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    const int ic = valid ? i : chunk * 64;       // a lane beyond the last site works on the first site of the chunk and stores nothing
    cd stg[3], xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
#pragma unroll
    for (int cc = 0; cc < 3; cc++) stg[cc] = ld(k.in[1 - p] + sp_off(12, ic) + (size_t)(3 * w + cc) * Vh);    // the staged chunk FIRST: its data must arrive first
    __builtin_amdgcn_sched_barrier(0);      // nothing is scheduled across this point: the staging loads stay the OLDEST in the in-order vmcnt queue
    cd own[3];
    switch (w) {
    case 0: dslds_body<0, DAG, R12, NTB>(k, p, chunk, ic, lane, lds, own, stg, xv, w); break;
    case 1: dslds_body<1, DAG, R12, NTB>(k, p, chunk, ic, lane, lds, own, stg, xv, w); break;
    case 2: dslds_body<2, DAG, R12, NTB>(k, p, chunk, ic, lane, lds, own, stg, xv, w); break;
    default: dslds_body<3, DAG, R12, NTB>(k, p, chunk, ic, lane, lds, own, stg, xv, w); break;
    }
    __syncthreads();
    real2 (*part)[3][3][64] = reinterpret_cast<real2 (*)[3][3][64]>(lds);
    real nrm = 0.0;
    if (valid) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            cd sv[4];
#pragma unroll
            for (int src = 0; src < 4; src++) {
                if (src == w) sv[src] = own[cc];
                else { const real2 t = part[w][src < w ? src : src - 1][cc][lane]; sv[src] = mk(t.x, t.y); }
            }
            cd s = mk((sv[0].re + sv[1].re) + (sv[2].re + sv[3].re), (sv[0].im + sv[1].im) + (sv[2].im + sv[3].im));
            cd v = k.b * s;
            v = mk(fma(k.a, xv[cc].re, v.re), fma(k.a, xv[cc].im, v.im));
            emit(k, p, (size_t)(3 * w + cc) * Vh + sp_off(12, i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) {
        double* red = reinterpret_cast<double*>(lds + 4 * 3 * 3 * 64 + 12 * 64);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// Buffer-addressed loads: the SRD (base, size) lives in SGPRs, every lane supplies ONE 32-bit byte offset and the component
// stride goes into the scalar offset -- no per-load 64-bit address VGPR pair / v_lshl_add_u64 (21 of them per hop otherwise).
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
__device__ inline __amdgpu_buffer_rsrc_t mkbuf(const real2* p, size_t elems) {
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const size_t bytes = elems * sizeof(real2);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0,
                                             bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)bytes, 0x00020000);
}
__device__ inline cd bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u4v v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    real2 d;
    __builtin_memcpy(&d, &v, sizeof(d));
    return mk(d.x, d.y);
}
// raw spinor components a hop needs (12, or the 6 the t projector keeps) and the link, all issued back to back
template <int MU, int S>
__device__ inline void load_hop_regs(cd* sp, cd (&u)[9], const real2* psi_block, const real2* gauge, size_t gauge_n, unsigned Vs,
                                     unsigned Us, unsigned psi_site, unsigned link_off) {
    const __amdgpu_buffer_rsrc_t rp = mkbuf(psi_block, (size_t)0x0FFFFFFF), ru = mkbuf(gauge, gauge_n);
    const unsigned vp = psi_site * 16u, vu = link_off * 16u, cs = Vs * 16u, us = Us * 16u;
    if constexpr (MU < 3) {
#pragma unroll
        for (int j = 0; j < 12; j++) sp[j] = bld(rp, vp, (unsigned)j * cs);
    } else {
        constexpr int base = S > 0 ? 2 : 0;
#pragma unroll
        for (int j = 0; j < 6; j++) sp[j] = bld(rp, vp, (unsigned)(base * 3 + j) * cs);
    }
#pragma unroll
    for (int j = 0; j < 9; j++) u[j] = bld(ru, vu, (unsigned)j * us);
}

template <int MU, bool BWD, bool DAG, bool NTG>
__device__ inline void hop_half(cd (&chi0)[3], cd (&chi1)[3], const KArgs& k, int p, int i) {
    Nbr n;
    int c[4];
    neighbours(k.g, p, i, n, c);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    constexpr int S = (DAG ? -1 : 1) * (BWD ? -1 : 1);
    const real sign = BWD ? n.sb[MU] : n.sf[MU];
    const int nb = BWD ? n.bwd[MU] : n.fwd[MU];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) { chi0[cc] = mk(0, 0); chi1[cc] = mk(0, 0); }
    if (sign != 0.0) {
        const real2* __restrict__ psi = k.in[1 - p] + sp_off(12, nb);
        const real2* __restrict__ U = k.gauge + (BWD ? glink_off(k.g, 1 - p, MU, nb) : glink_off(k.g, p, MU, i));
        const int Us = glink_stride(k.g);
        cd h0[3], h1[3], u[9];
        constexpr bool USE_BUF = false;   // measured: buffer-addressed loads are ~5 % slower than flat loads here (profiles/)
        if (USE_BUF && !(BWD && NTG)) {
            cd sp[MU < 3 ? 12 : 6];
            const int pp = BWD ? 1 - p : p;
            load_hop_regs<MU, S>(sp, u, k.in[1 - p], k.gauge, gauge_elems(k.g), (unsigned)Vh, (unsigned)Us, (unsigned)sp_off(12, nb),
                                 (unsigned)glink_off(k.g, pp, MU, BWD ? nb : i));
            project_regs<MU, S>(h0, h1, sp);
        } else {
            project<MU, S>(h0, h1, psi, Vh);
            // the backward hop is the LAST of the two uses of a link: optionally load it non-temporally
            if constexpr (BWD && NTG) load_link_nt(u, U, Us); else load_link(u, U, Us);
        }
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { h0[cc] = sign * h0[cc]; h1[cc] = sign * h1[cc]; }
        su3_mv<BWD>(chi0, u, h0);
        su3_mv<BWD>(chi1, u, h1);
    }
}

// NT bit 0: non-temporal backward-link loads; bit 1: non-temporal output stores
template <bool DAG, int NT>
__global__ __launch_bounds__(512) void wilson_hopsplit(KArgs k) {
    constexpr bool NTG = (NT & 1) != 0, NTS = (NT & 2) != 0;
    __shared__ real2 half[8][6][64];  // 48 KiB
    __shared__ double red[8];
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    cd xv[2] = {mk(0, 0), mk(0, 0)}, rv[2] = {mk(0, 0), mk(0, 0)};
    if (valid && w < 6 && k.a != 0.0) {
        xv[0] = ld(k.xin[p] + sp_off(12, i) + (size_t)(2 * w) * Vh);
        xv[1] = ld(k.xin[p] + sp_off(12, i) + (size_t)(2 * w + 1) * Vh);
    }
    if (valid && w < 6 && k.upd_scal) {
        rv[0] = ld(k.upd[p] + sp_off(12, i) + (size_t)(2 * w) * Vh);
        rv[1] = ld(k.upd[p] + sp_off(12, i) + (size_t)(2 * w + 1) * Vh);
    }
    cd chi0[3], chi1[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) { chi0[cc] = mk(0, 0); chi1[cc] = mk(0, 0); }
    if (valid) {
        switch (w) {
        case 0: hop_half<0, false, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 1: hop_half<0, true, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 2: hop_half<1, false, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 3: hop_half<1, true, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 4: hop_half<2, false, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 5: hop_half<2, true, DAG, NTG>(chi0, chi1, k, p, i); break;
        case 6: hop_half<3, false, DAG, NTG>(chi0, chi1, k, p, i); break;
        default: hop_half<3, true, DAG, NTG>(chi0, chi1, k, p, i); break;
        }
    }
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
        half[w][cc][lane] = mk2(chi0[cc].re, chi0[cc].im);
        half[w][3 + cc][lane] = mk2(chi1[cc].re, chi1[cc].im);
    }
    __syncthreads();
    real nrm = 0.0;
    if (valid && w < 6) {
        cd s0, s1;
        switch (w) {
        case 0: s0 = combine_comp<0, DAG>(half, lane); s1 = combine_comp<1, DAG>(half, lane); break;
        case 1: s0 = combine_comp<2, DAG>(half, lane); s1 = combine_comp<3, DAG>(half, lane); break;
        case 2: s0 = combine_comp<4, DAG>(half, lane); s1 = combine_comp<5, DAG>(half, lane); break;
        case 3: s0 = combine_comp<6, DAG>(half, lane); s1 = combine_comp<7, DAG>(half, lane); break;
        case 4: s0 = combine_comp<8, DAG>(half, lane); s1 = combine_comp<9, DAG>(half, lane); break;
        default: s0 = combine_comp<10, DAG>(half, lane); s1 = combine_comp<11, DAG>(half, lane); break;
        }
        cd v0 = mk(fma(k.a, xv[0].re, k.b * s0.re), fma(k.a, xv[0].im, k.b * s0.im));
        cd v1 = mk(fma(k.a, xv[1].re, k.b * s1.re), fma(k.a, xv[1].im, k.b * s1.im));
        if (k.upd_scal) {
            // CG update mode: r (prefetched at kernel start) <- r - alpha v ; q is never written
            const real al = k.upd_scal[S_ALPHA];
            cd r0 = mk(fma(-al, v0.re, rv[0].re), fma(-al, v0.im, rv[0].im));
            cd r1 = mk(fma(-al, v1.re, rv[1].re), fma(-al, v1.im, rv[1].im));
            nrm = r0.re * r0.re + r0.im * r0.im + r1.re * r1.re + r1.im * r1.im;
            real2* __restrict__ o = k.upd[p] + sp_off(12, i) + (size_t)(2 * w) * Vh;
            st(o, r0); st(o + Vh, r1);
        } else {
            real2* __restrict__ o = k.out[p] + sp_off(12, i) + (size_t)(2 * w) * Vh;
            nrm = v0.re * v0.re + v0.im * v0.im + v1.re * v1.re + v1.im * v1.im;
            if constexpr (NTS) { st_nt(o, v0); st_nt(o + Vh, v1); } else { st(o, v0); st(o + Vh, v1); }
        }
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0)
            k.norm_partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}

// ------------------------------------------------------------------------------------------ Wilson, hop-split, persistent
// Variant 3: the hop-split kernel as a persistent, software-pipelined loop.  The stencil is latency/MLP-bound (halving the
// resident workgroups costs only 1.3x, removing L2-miss traffic changes nothing -- profiles/), so the idle part of a
// workgroup's life matters: launch + index arithmetic before the first load, and barrier + LDS combine + store after the
// last one.  Here 2 workgroups per CU stay resident and walk the XCD's chunk sequence; every hop wave issues the 21 loads
// of its NEXT chunk before it enters the barrier/combine of the current one, so the memory system always has work.
template <int MU, bool BWD, bool DAG>
__device__ inline void hopsplit_persist_loop(const KArgs& k, real2 (*half)[6][64], double* red, int nvirt) {
    constexpr int W = 2 * MU + (BWD ? 1 : 0);
    constexpr int S = (DAG ? -1 : 1) * (BWD ? -1 : 1);
    constexpr int NS = (MU < 3) ? 12 : 6;   // spinor components this hop reads
    const int Vh = sp_stride(k.g);
    const int lane = threadIdx.x & 63;
    real nrm = 0.0;
    cd sp[NS], u[9];
    real sign = 0.0;
    int i = 0, p = 0;
    bool valid = false;

    auto issue = [&](int vb) {
        int chunk;
        map_block_v(k, vb, chunk, p);
        i = chunk * 64 + lane;
        valid = i < k.g.Vh;
        sign = 0.0;
        if (valid) {
            Nbr n;
            int c[4];
            neighbours(k.g, p, i, n, c);
            sign = BWD ? n.sb[MU] : n.sf[MU];
            const int nb = BWD ? n.bwd[MU] : n.fwd[MU];
            if (sign != 0.0) {
                const int pp = BWD ? 1 - p : p;
                load_hop_regs<MU, S>(sp, u, k.in[1 - p], k.gauge, gauge_elems(k.g), (unsigned)Vh, (unsigned)glink_stride(k.g), (unsigned)sp_off(12, nb),
                                     (unsigned)glink_off(k.g, pp, MU, BWD ? nb : i));
            }
        }
    };

    int vb = blockIdx.x;
    issue(vb);
    for (;;) {
        cd chi0[3], chi1[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { chi0[cc] = mk(0, 0); chi1[cc] = mk(0, 0); }
        if (valid && sign != 0.0) {
            cd h0[3], h1[3];
            project_regs<MU, S>(h0, h1, sp);
#pragma unroll
            for (int cc = 0; cc < 3; cc++) { h0[cc] = sign * h0[cc]; h1[cc] = sign * h1[cc]; }
            su3_mv<BWD>(chi0, u, h0);
            su3_mv<BWD>(chi1, u, h1);
        }
        const int ci = i, cp = p;
        const bool cvalid = valid;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            half[W][cc][lane] = mk2(chi0[cc].re, chi0[cc].im);
            half[W][3 + cc][lane] = mk2(chi1[cc].re, chi1[cc].im);
        }
        // operands of THIS chunk's epilogue first (they return first), then the NEXT chunk's hop loads
        cd xv[2] = {mk(0, 0), mk(0, 0)}, rv[2] = {mk(0, 0), mk(0, 0)};
        if constexpr (W < 6) {
            if (cvalid && k.a != 0.0) {
                xv[0] = ld(k.xin[cp] + sp_off(12, ci) + (size_t)(2 * W) * Vh);
                xv[1] = ld(k.xin[cp] + sp_off(12, ci) + (size_t)(2 * W + 1) * Vh);
            }
            if (cvalid && k.upd_scal) {
                rv[0] = ld(k.upd[cp] + sp_off(12, ci) + (size_t)(2 * W) * Vh);
                rv[1] = ld(k.upd[cp] + sp_off(12, ci) + (size_t)(2 * W + 1) * Vh);
            }
        }
        const int vb2 = vb + gridDim.x;
        const bool more = vb2 < nvirt;
        if (more) issue(vb2);
        __syncthreads();   // every hop of chunk vb is in LDS
        if constexpr (W < 6) {
            if (cvalid) {
                const cd s0 = combine_comp<2 * W, DAG>(half, lane), s1 = combine_comp<2 * W + 1, DAG>(half, lane);
                cd v0 = mk(fma(k.a, xv[0].re, k.b * s0.re), fma(k.a, xv[0].im, k.b * s0.im));
                cd v1 = mk(fma(k.a, xv[1].re, k.b * s1.re), fma(k.a, xv[1].im, k.b * s1.im));
                if (k.upd_scal) {
                    const real al = k.upd_scal[S_ALPHA];
                    v0 = mk(fma(-al, v0.re, rv[0].re), fma(-al, v0.im, rv[0].im));
                    v1 = mk(fma(-al, v1.re, rv[1].re), fma(-al, v1.im, rv[1].im));
                }
                nrm += v0.re * v0.re + v0.im * v0.im + v1.re * v1.re + v1.im * v1.im;
                real2* __restrict__ o = (k.upd_scal ? k.upd[cp] : k.out[cp]) + sp_off(12, ci) + (size_t)(2 * W) * Vh;
                st(o, v0);
                st(o + Vh, v1);
            }
        }
        __syncthreads();   // LDS is free again
        if (!more) break;
        vb = vb2;
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[W] = nrm;
        __syncthreads();
        if (W == 0 && lane == 0)
            k.norm_partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
    }
}

template <bool DAG>
__global__ __launch_bounds__(512, 4) void wilson_hopsplit_persist(KArgs k, int nvirt) {
    __shared__ real2 half[8][6][64];  // 48 KiB
    __shared__ double red[8];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (w) {
    case 0: hopsplit_persist_loop<0, false, DAG>(k, half, red, nvirt); break;
    case 1: hopsplit_persist_loop<0, true, DAG>(k, half, red, nvirt); break;
    case 2: hopsplit_persist_loop<1, false, DAG>(k, half, red, nvirt); break;
    case 3: hopsplit_persist_loop<1, true, DAG>(k, half, red, nvirt); break;
    case 4: hopsplit_persist_loop<2, false, DAG>(k, half, red, nvirt); break;
    case 5: hopsplit_persist_loop<2, true, DAG>(k, half, red, nvirt); break;
    case 6: hopsplit_persist_loop<3, false, DAG>(k, half, red, nvirt); break;
    default: hopsplit_persist_loop<3, true, DAG>(k, half, red, nvirt); break;
    }
}

// ------------------------------------------------------------------------------------------ launcher of the variants
bool launch_wilson_alt(lqcd_ctx_s* c, const StencilCall& s, const KArgs& k, size_t pad) {
    const int v = c->tun.dslash_variant;
    const bool partitioned = c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3];
    if (v == 8 && !k.clover && !partitioned) {
        dim3 grid(k.nblocks), block(256);
        const bool ntb = (k.nt & 1) != 0;
#define LQ_V8(D, R) do { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_both<D, R, true>), grid, block, pad, c->stream, k); \
                         else hipLaunchKernelGGL((wilson_dirsplit_both<D, R, false>), grid, block, pad, c->stream, k); } while (0)
        if (k.gauge12) { if (s.dagger) LQ_V8(true, true); else LQ_V8(false, true); }
        else { if (s.dagger) LQ_V8(true, false); else LQ_V8(false, false); }
#undef LQ_V8
        return true;
    }
    if (v == 7 && !k.clover && k.gauge12 && s.parity_mode == 2) {
        dim3 grid(k.nblocks / 2), block(512);
        if (s.dagger) hipLaunchKernelGGL((wilson_pair4<true, true>), grid, block, pad, c->stream, k);
        else hipLaunchKernelGGL((wilson_pair4<false, true>), grid, block, pad, c->stream, k);
        return true;
    }
    if (v == 6 && !k.clover && 64 % c->geom.XH == 0 && c->geom.XH <= 32) {
        dim3 grid(k.nblocks), block(256);
        const bool ntb = (k.nt & 1) != 0;     // backward-link loads non-temporal (tunable nt_gauge bit 0); bit 1 is not offered by this variant
#define LQ_V6(D, R) do { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_lds<D, R, true>), grid, block, pad, c->stream, k); \
                         else hipLaunchKernelGGL((wilson_dirsplit_lds<D, R, false>), grid, block, pad, c->stream, k); } while (0)
        if (k.gauge12) { if (s.dagger) LQ_V6(true, true); else LQ_V6(false, true); }
        else { if (s.dagger) LQ_V6(true, false); else LQ_V6(false, false); }
#undef LQ_V6
        return true;
    }
    if (v == 5 && !k.clover) {
        dim3 grid(k.nblocks), block(256);
        if (k.gauge12) {
            if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit4<true, true>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_dirsplit4<false, true>), grid, block, pad, c->stream, k);
        } else {
            if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit4<true, false>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_dirsplit4<false, false>), grid, block, pad, c->stream, k);
        }
        return true;
    }
    if (v == 4 && !k.clover) {
        dim3 grid(k.nblocks), block(256);
        if (k.gauge12) {
            if (s.dagger) hipLaunchKernelGGL((wilson_lanesplit<true, true>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_lanesplit<false, true>), grid, block, pad, c->stream, k);
        } else {
            if (s.dagger) hipLaunchKernelGGL((wilson_lanesplit<true, false>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_lanesplit<false, false>), grid, block, pad, c->stream, k);
        }
        return true;
    }
    if (v == 3) {
        dim3 grid(persist_grid(c, k.nblocks)), block(512);
        if (s.dagger) hipLaunchKernelGGL((wilson_hopsplit_persist<true>), grid, block, pad, c->stream, k, k.nblocks);
        else hipLaunchKernelGGL((wilson_hopsplit_persist<false>), grid, block, pad, c->stream, k, k.nblocks);
        return true;
    }
    if (v == 2) {
        dim3 grid(k.nblocks), block(512);
        const int nt = (c->tun.nt_gauge ? 1 : 0) | (c->tun.nt_store ? 2 : 0);
#define LQ_HS(D, N) hipLaunchKernelGGL((wilson_hopsplit<D, N>), grid, block, pad, c->stream, k)
        if (s.dagger) { switch (nt) { case 1: LQ_HS(true, 1); break; case 2: LQ_HS(true, 2); break; case 3: LQ_HS(true, 3); break; default: LQ_HS(true, 0); } }
        else { switch (nt) { case 1: LQ_HS(false, 1); break; case 2: LQ_HS(false, 2); break; case 3: LQ_HS(false, 3); break; default: LQ_HS(false, 0); } }
#undef LQ_HS
        return true;
    }
    return false;
}

}  // inline namespace (precision)
}  // namespace lqcd
