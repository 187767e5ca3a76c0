// stencil.hip -- Wilson and staggered Dslash for gfx950 (CDNA4), fp64.
//
// Replaces LinearAlgebra.mul!(y, D::Dirac_operator, x) / mul!(y, D', x) of LatticeDiracOperators.jl
// (operator built at /root/reference/src/system/universe.jl:106-116,137).
//
// Design (HBM-bandwidth-bound, 1.375 flop/B -> no MFMA):
//   * one lane per output site, lanes walk the checkerboard index, so every load is a 16-B/lane
//     global_load_dwordx4 over >=256-B contiguous runs (1 KiB per wave instruction in the bulk);
//   * spin projection (r = 1): only 6 of the 12 neighbour components enter the 3x3 colour mat-vec;
//     the t direction loads only the 6 components its projector keeps;
//   * workgroup -> lattice map is XCD-aware: XCD k (= blockIdx % 8) owns a contiguous range of the
//     checkerboard index (a t-slab) and processes the even and odd sites of a chunk back to back, so
//     the second use of every link (U_mu(n) forward from n, backward from n+mu) and the 8-fold spinor
//     reuse are served from that XCD's L2 / the Infinity Cache instead of HBM;
//   * optional fused |out|^2 block partials (CG: p.(D^+ D p) = |D p|^2) save a full pass over the field.
// Multi-GPU: hops that leave the rank are skipped by the interior kernel and added by the exterior
// kernels from spin-projected halos packed by pack kernels (see halo layout in lqcd_internal.h).
#include "lqcd_internal.h"
#include <algorithm>
#include <type_traits>

namespace lqcd {
inline namespace LQCD_PNS {

struct KArgs {
    Geom g;
    const real2* gauge;
    const real2* gauge12;   // 12-real links (rows 0,1) or nullptr
    const real2* clover;    // packed chiral clover blocks (clover.hip) or nullptr: the diagonal term becomes a * (A xin)
    real2* out[2];
    const real2* in[2];
    const real2* xin[2];
    real a, b, r;
    int parity_mode;
    int nblocks;
    int remap;
    int cps;   // chunks per t-slice per parity if the slice divides evenly into chunks and by nsub, else 0 (remap 2)
    int nsub;  // sub-domains per t-slice (multiple of 8): XCD k sweeps sub-domains k, k+8, ... one after the other
    int cpp;   // chunks per z-plane per parity
    int ysplit;  // sub-domains are (y,z) tiles: ysplit tiles across y (1 = plain z-slabs)
    int cpr, ty, tz;                              // derived tile sizes (chunks)
    FastDiv d_perpass, d_cpr, d_ysplit, d_ty;     // magic numbers for the block -> chunk map
#ifdef LQCD_ABLATE
    int dbg;     // timing ablations only (wrong results; -DLQCD_ABLATE builds): see dirsplit_hops / hop_half
#endif
    int nt;      // bit 0: non-temporal backward-link loads, bit 1: non-temporal forward-link loads, bit 2: non-temporal output stores
    double* norm_partial;
    const double* upd_scal;   // update mode (see StencilCall)
    real2* upd[2];
    const double* skip;       // scalar block whose S_DONE flag turns the launch into a no-op (the solver has converged)
    const double* alpha_partials;   // cg_small (see StencilCall): block partials of |D p|^2 to be summed in the prologue, or nullptr
    int alpha_n;
    double* scal_w;
};

typedef real v2d __attribute__((ext_vector_type(2)));
__device__ inline cd ld_nt(const real2* p) {
    v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
    return mk(v.x, v.y);
}
__device__ inline void load_link_nt(cd (&u)[9], const real2* __restrict__ U, int Vh) {
#pragma unroll
    for (int j = 0; j < 9; j++) u[j] = ld_nt(U + (size_t)j * Vh);
}
__device__ inline void st_nt(real2* p, cd v) {
    v2d t = {v.re, v.im};
    __builtin_nontemporal_store(t, reinterpret_cast<v2d*>(p));
}

// Addressing of a Wilson (12-component) spinor and of a 12-real link inside their 64-site chunks.
//   fp64 build: component j of a site at sp_off(12, i) + j * 64 -- one 16-byte element per lane and load instruction.
//   fp32 build: an element is 8 bytes, and 8-byte accesses run at 0.54-0.70x the 16-byte rate on this memory pipeline
//   (MI355X_MICROARCH.md, cache-policy table), so the fp32 fields of the mixed-precision solver keep TWO consecutive components
//   in one 16-byte word: [chunk][component pair][lane][2].  A hop then issues 6 + 3 loads of 16 bytes instead of 12 + 6 of 8.
//   (Staggered 3-component spinors and the 18-real fp32 links keep the fp64 arrangement.)
#ifdef LQCD_F32
__device__ inline size_t sp12_off(int i) { return (size_t)(i >> 6) * (12 * 64) + (size_t)(i & 63) * 2; }
__device__ constexpr size_t co12(int j) { return (size_t)(j >> 1) * 128 + (size_t)(j & 1); }
__device__ inline size_t gl12_off(const Geom& g, int p, int mu, int i) {
    return ((((size_t)p * g.nch + (size_t)(i >> 6)) * 4 + mu) * 6) * 64 + (size_t)(i & 63) * 2;
}
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ inline void ld_pair(cd& a, cd& b, const real2* p) {
    const v4f* q = reinterpret_cast<const v4f*>(p);
    v4f v;
    if constexpr (NT) v = __builtin_nontemporal_load(q); else v = *q;
    a = mk(v.x, v.y); b = mk(v.z, v.w);
}
#else
__device__ inline size_t sp12_off(int i) { return sp_off(12, i); }
__device__ constexpr size_t co12(int j) { return (size_t)j * 64; }
__device__ inline size_t gl12_off(const Geom& g, int p, int mu, int i) { return glink12_off(g, p, mu, i); }
#endif
// components FIRST .. FIRST+N-1 of a Wilson spinor (FIRST and N even)
template <int FIRST, int N, bool NT>
__device__ inline void load_comps12(cd* sp, const real2* __restrict__ psi) {
#ifdef LQCD_F32
#pragma unroll
    for (int q = 0; q < N / 2; q++) ld_pair<NT>(sp[2 * q], sp[2 * q + 1], psi + co12(FIRST + 2 * q));
#else
#pragma unroll
    for (int j = 0; j < N; j++) sp[j] = NT ? ld_nt(psi + co12(FIRST + j)) : ld(psi + co12(FIRST + j));
#endif
}

// final store of one output component: plain (out = v) or CG update mode (r -= alpha v); accumulates the squared norm
__device__ inline void emit(const KArgs& k, int p, size_t off, cd v, real& nrm, real al) {
    if (k.upd_scal) {
        real2* rp = k.upd[p] + off;
        cd r = ld(rp);
        r.re = fma(-al, v.re, r.re); r.im = fma(-al, v.im, r.im);
        nrm = fma(r.re, r.re, nrm); nrm = fma(r.im, r.im, nrm);
        st(rp, r);
    } else {
        nrm = fma(v.re, v.re, nrm); nrm = fma(v.im, v.im, nrm);
        if (k.nt & 4) st_nt(k.out[p] + off, v); else st(k.out[p] + off, v);
    }
}
// the same with the old value of r already in registers (its load was issued ahead of the hops: one dependent memory round trip less)
__device__ inline void emit_pre(const KArgs& k, int p, size_t off, cd v, real& nrm, real al, cd r) {
    if (k.upd_scal) {
        r.re = fma(-al, v.re, r.re); r.im = fma(-al, v.im, r.im);
        nrm = fma(r.re, r.re, nrm); nrm = fma(r.im, r.im, nrm);
        st(k.upd[p] + off, r);
    } else {
        nrm = fma(v.re, v.re, nrm); nrm = fma(v.im, v.im, nrm);
        if (k.nt & 4) st_nt(k.out[p] + off, v); else st(k.out[p] + off, v);
    }
}
__device__ inline bool upd_done(const KArgs& k) {
    const bool done = (k.upd_scal && k.upd_scal[S_DONE] != 0.0) || (k.skip && k.skip[S_DONE] != 0.0);
    // cg_small: the update-mode launch of an overshooting iteration tells the x/p update behind it that the converging iterate is complete
    if (done && k.scal_w && blockIdx.x == 0 && threadIdx.x == 0) k.scal_w[S_XDONE] = 1.0;
    return done;
}
// alpha of the CG update mode: from the scalar block, or (cg_small) rr / sum of the previous kernel's block partials, formed by every wave
__device__ inline real update_alpha(const KArgs& k) {
    if (!k.upd_scal) return real(0);
    if (k.alpha_partials) {
        const double pq = sum_partials_small(k.alpha_partials, k.alpha_n);
        const double rr = k.upd_scal[S_RR];
        const double al = rr / pq;
        if (blockIdx.x == 0 && threadIdx.x == 0) { k.scal_w[S_PQ] = pq; k.scal_w[S_ALPHA] = al; k.scal_w[S_RROLD] = rr; }
        return (real)al;
    }
    if (k.scal_w) {      // folded scalar step (several ranks): pq has been all-reduced into the scalar block, alpha is formed here instead of by a
        const double rr = k.upd_scal[S_RR];            // one-thread kernel between the all-reduce and this launch
        const double al = rr / k.upd_scal[S_PQ];
        if (blockIdx.x == 0 && threadIdx.x == 0) { k.scal_w[S_ALPHA] = al; k.scal_w[S_RROLD] = rr; }
        return (real)al;
    }
    return (real)k.upd_scal[S_ALPHA];
}

struct HArgs {  // halo kernels
    Geom g;
    const real2* gauge;
    real2* out[2];
    const real2* in[2];
    real b;
    int parity_mode;
    int dagger;
    real2* send_fwd[4];
    real2* send_bwd[4];
    const real2* recv_fwd[4];
    const real2* recv_bwd[4];
    real sign_fwd[4];  // bc sign if this rank sits on the global upper boundary, else 1
    real sign_bwd[4];
    double* norm_partial;     // if non-null: per-block CORRECTIONS sum(|v_after|^2 - |v_before|^2) go to norm_partial[partial_offset + block]
    int partial_offset;
    const double* upd_scal;   // CG update mode: target is upd (r) and the coefficient is -alpha*b
    real2* upd[2];
};

// workgroup -> (chunk of consecutive checkerboard sites, parity).  Observed (not contractual) dispatch: block b runs on
// XCD b % 8, blocks of one XCD start in increasing b.  The maps only change speed, never results.
//   remap 0: plain -- consecutive blocks = even/odd halves of consecutive chunks, round-robin over the XCDs
//   remap 1: XCD k owns a contiguous 1/8 of the chunk list (a t-slab), even/odd of a chunk back to back
//   remap 2: XCD k owns 1/8 of every t-slice (a z-slab) and sweeps t: the t-neighbour re-use distance is one slab step
//            (fits the 4 MiB L2) and the 8 XCDs advance through t together (z-halo lines are shared through the MALL)
__device__ inline void map_block_v(const KArgs& k, int b, int& chunk, int& p) {
    const int nb = k.nblocks;
    const bool both = k.parity_mode == 2;
    if (k.remap == 2 && k.cps > 0) {
        const int cpr = k.cpr;                      // chunks per sub-domain per t-slice (per parity)
        const int xcd = b & 7;
        int j = b >> 3;
        if (both) { p = j & 1; j >>= 1; } else p = k.parity_mode;
        const int per_pass = cpr * k.g.L[3];
        const int pass = fdiv(j, k.d_perpass);
        j -= pass * per_pass;
        const int t = fdiv(j, k.d_cpr), m = j - t * cpr, sd = xcd + 8 * pass;
        int s;
        if (k.ysplit > 1) {
            // 2-D tiling of the (y-chunk, z) grid of a t-slice: sub-domain sd = (sy, sz), tile ty x tz chunks
            const int sz = fdiv(sd, k.d_ysplit), sy = sd - sz * k.ysplit;
            const int ty = k.ty, tz = k.tz;
            const int zz = fdiv(m, k.d_ty), yy = m - zz * ty;
            s = (sz * tz + zz) * k.cpp + sy * ty + yy;
        } else {
            s = sd * cpr + m;
        }
        chunk = t * k.cps + s;
        return;
    }
    int lb = b;
    if (k.remap && !(nb & 7)) lb = (b & 7) * (nb >> 3) + (b >> 3);
    if (both) { chunk = lb >> 1; p = lb & 1; } else { chunk = lb; p = k.parity_mode; }
}

__device__ inline void map_block(const KArgs& k, int& chunk, int& p) { map_block_v(k, blockIdx.x, chunk, p); }

// gamma_mu (mu = 0,1,2) has one entry per row: row a -> column PERM[mu][a], value i^GK[mu][a]
// (SURVEY.md Appendix A).  gamma_4 = diag(1,1,-1,-1).
// (tables PERM / GK: lqcd_internal.h)

// first term of a complex accumulation chain: the operation sequence of cfma / cfma_conj on a zero accumulator with the leading
// fma(x, y, 0) written as x * y (the same value up to the sign of an exact zero) -- no register clear, one v_mul instead of v_mov + v_fma
__device__ inline cd cmul_first(cd a, cd b) {
    cd t;
    t.re = a.re * b.re; t.re = fma(-a.im, b.im, t.re);
    t.im = a.re * b.im; t.im = fma(a.im, b.re, t.im);
    return t;
}
__device__ inline cd cmul_conj_first(cd a, cd b) {
    cd t;
    t.re = a.re * b.re; t.re = fma(a.im, b.im, t.re);
    t.im = a.re * b.im; t.im = fma(-a.im, b.re, t.im);
    return t;
}
template <bool ADJ>
__device__ inline void su3_mv(cd (&chi)[3], const cd (&u)[9], const cd (&h)[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        cd t = ADJ ? cmul_conj_first(u[a], h[0]) : cmul_first(u[a * 3], h[0]);
#pragma unroll
        for (int b = 1; b < 3; b++) {
            if constexpr (ADJ) cfma_conj(t, u[b * 3 + a], h[b]);
            else cfma(t, u[a * 3 + b], h[b]);
        }
        chi[a] = t;
    }
}

__device__ inline void load_link(cd (&u)[9], const real2* __restrict__ U, int Vh) {
#pragma unroll
    for (int k = 0; k < 9; k++) u[k] = ld(U + (size_t)k * Vh);
}

// third row of an SU(3) matrix from the first two: u[6+b] = conj(u[b1] u[3+b2] - u[b2] u[3+b1]), one accumulation chain per real part
// (8 fp64 instructions per element; every 12-real path of this file uses this one function, so the variants agree bit for bit)
__device__ inline void recon_row2(cd (&u)[9]) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const cd a = u[b1], bb = u[3 + b2], c = u[b2], d = u[3 + b1];
        real wr = a.re * bb.re;
        wr = fma(-a.im, bb.im, wr);
        wr = fma(-c.re, d.re, wr);
        wr = fma(c.im, d.im, wr);
        real wi = a.re * bb.im;
        wi = fma(a.im, bb.re, wi);
        wi = fma(-c.re, d.im, wi);
        wi = fma(-c.im, d.re, wi);
        u[6 + b] = mk(wr, -wi);
    }
}

// 12-real links: rows 0 and 1 from memory, row 2 = conj(row 0 x row 1)  (exact for SU(3) to rounding)
__device__ inline void load_link12(cd (&u)[9], const real2* __restrict__ U, bool nt = false) {
#ifdef LQCD_F32
    if (nt) {
#pragma unroll
        for (int q = 0; q < 3; q++) ld_pair<true>(u[2 * q], u[2 * q + 1], U + (size_t)q * 128);
    } else {
#pragma unroll
        for (int q = 0; q < 3; q++) ld_pair<false>(u[2 * q], u[2 * q + 1], U + (size_t)q * 128);
    }
#else
    if (nt) {
#pragma unroll
        for (int k = 0; k < 6; k++) u[k] = ld_nt(U + (size_t)k * 64);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) u[k] = ld(U + (size_t)k * 64);
    }
#endif
    recon_row2(u);
}

// spin projection h = rows 0,1 of (1 - S*gamma_mu) psi   (mu = 3: the two non-zero rows, factor 2 included)
template <int MU, int S>
__device__ inline void project(cd (&h0)[3], cd (&h1)[3], const real2* __restrict__ psi, int Vh, bool nt = false) {
#ifdef LQCD_F32
    if constexpr (MU < 3) {
        constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
        constexpr int k0 = GK[MU][0] + (S > 0 ? 2 : 0), k1 = GK[MU][1] + (S > 0 ? 2 : 0);
        cd sp[12];
        load_comps12<0, 12, false>(sp, psi);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            h0[c] = sp[c] + mul_ipow<k0>(sp[p0 * 3 + c]);
            h1[c] = sp[3 + c] + mul_ipow<k1>(sp[p1 * 3 + c]);
        }
    } else {
        constexpr int base = S > 0 ? 2 : 0;
        cd sp[6];
        if (nt) load_comps12<base * 3, 6, true>(sp, psi); else load_comps12<base * 3, 6, false>(sp, psi);
#pragma unroll
        for (int c = 0; c < 3; c++) { h0[c] = 2.0 * sp[c]; h1[c] = 2.0 * sp[3 + c]; }
    }
    return;
#endif
    if constexpr (MU == 3) {
        if (nt) {       // last use of these spinor lines in the t-sweep of the workgroup map: stream them
            constexpr int base = S > 0 ? 2 : 0;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                h0[c] = 2.0 * ld_nt(psi + (size_t)(base * 3 + c) * Vh);
                h1[c] = 2.0 * ld_nt(psi + (size_t)((base + 1) * 3 + c) * Vh);
            }
            return;
        }
    }
    if constexpr (MU < 3) {
        constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
        constexpr int k0 = GK[MU][0] + (S > 0 ? 2 : 0), k1 = GK[MU][1] + (S > 0 ? 2 : 0);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            h0[c] = ld(psi + (size_t)(0 * 3 + c) * Vh) + mul_ipow<k0>(ld(psi + (size_t)(p0 * 3 + c) * Vh));
            h1[c] = ld(psi + (size_t)(1 * 3 + c) * Vh) + mul_ipow<k1>(ld(psi + (size_t)(p1 * 3 + c) * Vh));
        }
    } else {
        constexpr int base = S > 0 ? 2 : 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            h0[c] = 2.0 * ld(psi + (size_t)(base * 3 + c) * Vh);
            h1[c] = 2.0 * ld(psi + (size_t)((base + 1) * 3 + c) * Vh);
        }
    }
}

// acc += (1 - S*gamma_mu) reconstructed from the two rows chi0, chi1
template <int MU, int S>
__device__ inline void reconstruct(cd (&acc)[12], const cd (&chi0)[3], const cd (&chi1)[3]) {
    if constexpr (MU < 3) {
        constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
        constexpr int k0 = -GK[MU][0] + (S > 0 ? 2 : 0) + 8, k1 = -GK[MU][1] + (S > 0 ? 2 : 0) + 8;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            acc[c] = acc[c] + chi0[c];
            acc[3 + c] = acc[3 + c] + chi1[c];
            acc[p0 * 3 + c] = acc[p0 * 3 + c] + mul_ipow<k0>(chi0[c]);
            acc[p1 * 3 + c] = acc[p1 * 3 + c] + mul_ipow<k1>(chi1[c]);
        }
    } else {
        constexpr int base = S > 0 ? 2 : 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            acc[base * 3 + c] = acc[base * 3 + c] + chi0[c];
            acc[(base + 1) * 3 + c] = acc[(base + 1) * 3 + c] + chi1[c];
        }
    }
}

// one hop, r = 1:  acc += (1 - S gamma_mu) [U or U^+] psi(nb) * sign
// The boundary sign is +1 for every lane of almost every wave (periodic directions; interior of the antiperiodic one): the twelve
// multiplications by it are skipped under a wave-uniform test.  x * 1 = x, so the results do not change.
template <int MU, int S, bool ADJ, bool R12 = false>
__device__ inline void wilson_hop(cd (&acc)[12], const real2* __restrict__ psi, const real2* __restrict__ U,
                                  int Vh, int Us, real sign, bool nt = false, bool nt_psi = false) {
    cd h0[3], h1[3], chi0[3], chi1[3], u[9];
    project<MU, S>(h0, h1, psi, Vh, nt_psi);
    if constexpr (R12) load_link12(u, U, nt);
    else { if (nt) load_link_nt(u, U, Us); else load_link(u, U, Us); }
    if (__builtin_amdgcn_ballot_w64(sign != real(1.0)) != 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { h0[c] = sign * h0[c]; h1[c] = sign * h1[c]; }
    }
    su3_mv<ADJ>(chi0, u, h0);
    su3_mv<ADJ>(chi1, u, h1);
    reconstruct<MU, S>(acc, chi0, chi1);
}

// one hop, general r:  acc += (r - S gamma_mu) [U or U^+] psi(nb) * sign
template <int MU, int S, bool ADJ>
__device__ inline void wilson_hop_rgen(cd (&acc)[12], const real2* __restrict__ psi, const real2* __restrict__ U,
                                       int Vh, int Us, real sign, real r) {
    cd u[9], t[4][3];
    load_link(u, U, Us);
#pragma unroll
    for (int s = 0; s < 4; s++) {
        cd h[3];
#pragma unroll
        for (int c = 0; c < 3; c++) h[c] = sign * ld(psi + co12(s * 3 + c));
        su3_mv<ADJ>(t[s], u, h);
    }
    if constexpr (MU < 3) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            // -S * g(s) * t[perm(s)]
#pragma unroll
            for (int c = 0; c < 3; c++) {
                cd v = r * t[s][c];
                cd w;
                if (s == 0) w = mul_ipow<GK[MU][0] + (S > 0 ? 2 : 0)>(t[PERM[MU][0]][c]);
                else if (s == 1) w = mul_ipow<GK[MU][1] + (S > 0 ? 2 : 0)>(t[PERM[MU][1]][c]);
                else if (s == 2) w = mul_ipow<GK[MU][2] + (S > 0 ? 2 : 0)>(t[PERM[MU][2]][c]);
                else w = mul_ipow<GK[MU][3] + (S > 0 ? 2 : 0)>(t[PERM[MU][3]][c]);
                acc[s * 3 + c] = acc[s * 3 + c] + v + w;
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            real d = (s < 2) ? 1.0 : -1.0;
            real f = r - (real)S * d;
#pragma unroll
            for (int c = 0; c < 3; c++) acc[s * 3 + c] = acc[s * 3 + c] + f * t[s][c];
        }
    }
}

// neighbour bookkeeping for one site
struct Nbr {
    int fwd[4], bwd[4];
    real sf[4], sb[4];   // sign (0 => hop is off-rank, skipped by the interior kernel)
};

__device__ inline void neighbours(const Geom& g, int p, int i, Nbr& n, int c[4]) {
    cb_to_coords(g, p, i, c);
    const int q = c[0] & 1;
    const int s1 = g.XH, s2 = g.XH * g.L[1], s3 = s2 * g.L[2];
    // x
    {
        bool wf = c[0] == g.L[0] - 1, wb = c[0] == 0;
        n.fwd[0] = q ? (wf ? i - (g.XH - 1) : i + 1) : i;
        n.bwd[0] = q ? i : (wb ? i + (g.XH - 1) : i - 1);
        n.sf[0] = wf ? (g.part[0] ? 0.0 : g.bc_fwd[0]) : 1.0;
        n.sb[0] = wb ? (g.part[0] ? 0.0 : g.bc_bwd[0]) : 1.0;
    }
    const int strides[4] = {0, s1, s2, s3};
#pragma unroll
    for (int mu = 1; mu < 4; mu++) {
        bool wf = c[mu] == g.L[mu] - 1, wb = c[mu] == 0;
        n.fwd[mu] = wf ? i - (g.L[mu] - 1) * strides[mu] : i + strides[mu];
        n.bwd[mu] = wb ? i + (g.L[mu] - 1) * strides[mu] : i - strides[mu];
        n.sf[mu] = wf ? (g.part[mu] ? 0.0 : g.bc_fwd[mu]) : 1.0;
        n.sb[mu] = wb ? (g.part[mu] ? 0.0 : g.bc_bwd[mu]) : 1.0;
    }
}

template <int TB>
__device__ inline void block_norm_partial(double v, double* partial) {
    __shared__ double red[TB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < TB / 64; w++) s += red[w];
        partial[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------ Wilson
template <int TB, bool DAG, bool RGEN>
__global__ __launch_bounds__(TB) void wilson_interior(KArgs k) {
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    const int i = chunk * TB + threadIdx.x;
    const bool valid = i < k.g.Vh;
    real nrm = 0.0;
    if (valid) {
        Nbr n;
        int c[4];
        neighbours(k.g, p, i, n, c);
        cd acc[12], xv[12];
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
        // diagonal term: issue its loads first so they overlap the hops instead of forming a ninth dependent round trip
        if (k.a != 0.0) {
#pragma unroll
            for (int j = 0; j < 12; j++) xv[j] = ld(k.xin[p] + sp12_off(i) + co12(j));
        } else {
#pragma unroll
            for (int j = 0; j < 12; j++) xv[j] = mk(0.0, 0.0);
        }
        const real2* __restrict__ psi = k.in[1 - p];
        const int Us = glink_stride(k.g);
        constexpr int SF = DAG ? -1 : 1;  // forward hop: (r - gamma) for D, (r + gamma) for D^+
#define HOP(MU)                                                                                                  \
    if (n.sf[MU] != 0.0) {                                                                                       \
        if constexpr (RGEN) wilson_hop_rgen<MU, SF, false>(acc, psi + sp12_off(n.fwd[MU]), k.gauge + glink_off(k.g, p, MU, i), Vh, Us, n.sf[MU], k.r); \
        else wilson_hop<MU, SF, false>(acc, psi + sp12_off(n.fwd[MU]), k.gauge + glink_off(k.g, p, MU, i), Vh, Us, n.sf[MU]);            \
    }                                                                                                            \
    if (n.sb[MU] != 0.0) {                                                                                       \
        if constexpr (RGEN) wilson_hop_rgen<MU, -SF, true>(acc, psi + sp12_off(n.bwd[MU]), k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]), Vh, Us, n.sb[MU], k.r); \
        else wilson_hop<MU, -SF, true>(acc, psi + sp12_off(n.bwd[MU]), k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]), Vh, Us, n.sb[MU]); \
    }
        HOP(0) HOP(1) HOP(2) HOP(3)
#undef HOP
#pragma unroll
        for (int j = 0; j < 12; j++) {
            cd v = mk(fma(k.b, acc[j].re, k.a * xv[j].re), fma(k.b, acc[j].im, k.a * xv[j].im));
            emit(k, p, co12(j) + sp12_off(i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) block_norm_partial<TB>(nrm, k.norm_partial);
}

// ------------------------------------------------------------------------------------------ Wilson, direction-split
// Variant 1 ("dirsplit"): a workgroup of 4 waves owns 64 consecutive checkerboard sites; wave w computes the two hops
// of direction mu = w (all 42 loads of the direction issued as two short bursts), the four partial spinors are combined
// through LDS and wave w writes spin row w.  Compared with one-lane-does-all-8-hops this cuts the wave lifetime ~6x and
// phase-aligns the waves that touch the same lines, so the re-use of psi (9x) and of the links (2x) falls inside the
// residency time of the XCD's 4 MiB L2 (measured: profiles/).  r = 1 only.
template <int MU, bool DAG, bool R12>
__device__ inline void dirsplit_hops(cd (&acc)[12], const KArgs& k, int p, int i) {
    Nbr n;
    int c[4];
    neighbours(k.g, p, i, n, c);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    const real2* __restrict__ psi = k.in[1 - p];
    const real2* __restrict__ Uf = R12 ? k.gauge12 + gl12_off(k.g, p, MU, i) : k.gauge + glink_off(k.g, p, MU, i);
    const real2* __restrict__ Ub = R12 ? k.gauge12 + gl12_off(k.g, 1 - p, MU, n.bwd[MU]) : k.gauge + glink_off(k.g, 1 - p, MU, n.bwd[MU]);
    const int Us = glink_stride(k.g);
    constexpr int SF = DAG ? -1 : 1;
#ifdef LQCD_ABLATE
    if (k.dbg & (16 << MU)) return;   // traffic ablation (wrong results): drop both hops of direction MU
    if (k.dbg >= 256) {               // traffic ablations (wrong results): redirect one stream of direction MU to chunk 0 (always L2-hot)
        const int hot = i & 63;
        if (k.dbg & (256 << MU)) { n.fwd[MU] = hot; n.bwd[MU] = hot; }
        if (k.dbg & (4096 << MU)) Uf = R12 ? k.gauge12 + gl12_off(k.g, p, MU, hot) : k.gauge + glink_off(k.g, p, MU, hot);
        if (k.dbg & (65536 << MU)) Ub = R12 ? k.gauge12 + gl12_off(k.g, 1 - p, MU, hot) : k.gauge + glink_off(k.g, 1 - p, MU, hot);
    }
#endif
#ifdef LQCD_ABLATE
    if (MU < 2 && (k.dbg & ((1 << 20) << MU))) {     // ablation (wrong results): NO neighbour-spinor loads for this direction (upper bound of what LDS staging can save)
        cd h0[3], h1[3], chi0[3], chi1[3], u[9];
#pragma unroll
        for (int hop = 0; hop < 2; hop++) {
            if constexpr (R12) load_link12(u, hop ? Ub : Uf, hop ? (k.nt & 1) != 0 : false); else load_link(u, hop ? Ub : Uf, Us);
#pragma unroll
            for (int cc = 0; cc < 3; cc++) { h0[cc] = mk(real(1 + cc + i), real(2)); h1[cc] = mk(real(3), real(4 + hop)); }
            if (hop) { su3_mv<true>(chi0, u, h0); su3_mv<true>(chi1, u, h1); reconstruct<MU, -SF>(acc, chi0, chi1); }
            else { su3_mv<false>(chi0, u, h0); su3_mv<false>(chi1, u, h1); reconstruct<MU, SF>(acc, chi0, chi1); }
        }
        return;
    }
#endif
    if (n.sf[MU] != 0.0) wilson_hop<MU, SF, false, R12>(acc, psi + sp12_off(n.fwd[MU]), Uf, Vh, Us, n.sf[MU], (k.nt & 2) != 0);
    if (n.sb[MU] != 0.0) wilson_hop<MU, -SF, true, R12>(acc, psi + sp12_off(n.bwd[MU]), Ub, Vh, Us, n.sb[MU], (k.nt & 1) != 0, (k.nt & 8) != 0);
}

// rows 3*W .. 3*W+2 of A x for the packed clover field (clover.hip: two Hermitian 6x6 blocks in the chiral basis chi_(-+) =
// psi_upper -+ psi_lower; block b at 18 b: 3 elements = 6 real diagonals, then the upper triangle, entry (r,q) at
// 3 + 5 r - r(r-1)/2 + (q - r - 1)).  Spin row W needs rows S1 = W & 1 of both blocks:
//   (A x)_W = 1/2 [ -+ (A_+ chi_+)_{S1} + (A_- chi_-)_{S1} ]   (upper sign for W >= 2).
template <int S1>
__device__ inline void clover_rows(cd (&out)[3], const real2* __restrict__ a, const cd (&psi)[12], bool lower) {
#pragma unroll
    for (int cc = 0; cc < 3; cc++) out[cc] = mk(0.0, 0.0);
#pragma unroll
    for (int b = 0; b < 2; b++) {
        const real sg = b == 0 ? real(-1.0) : real(1.0);
        cd chi[6];
#pragma unroll
        for (int q = 0; q < 6; q++) chi[q] = mk(psi[q].re + sg * psi[6 + q].re, psi[q].im + sg * psi[6 + q].im);
        const real2* __restrict__ ab = a + (size_t)(18 * b) * 64;
        const real wgt = real(0.5) * ((b == 0 && lower) ? real(-1.0) : real(1.0));
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            constexpr int dummy = 0; (void)dummy;
            const int r = 3 * S1 + cc;
            const cd dd = ld(ab + (size_t)(r >> 1) * 64);
            const real d = (r & 1) ? dd.im : dd.re;
            cd y = mk(d * chi[r].re, d * chi[r].im);
#pragma unroll
            for (int q = 0; q < 6; q++) {
                if (q == r) continue;
                const int lo = q < r ? q : r, hi = q < r ? r : q;
                const cd m = ld(ab + (size_t)(3 + 5 * lo - (lo * (lo - 1)) / 2 + (hi - lo - 1)) * 64);
                if (q > r) cfma(y, m, chi[q]); else cfma_conj(y, m, chi[q]);
            }
            out[cc] = mk(out[cc].re + wgt * y.re, out[cc].im + wgt * y.im);
        }
    }
}

// fp32 build: half-size registers and 24 KiB of LDS leave room for 5 workgroups per CU if the compiler stays within 96 VGPRs (it does,
// without spilling, except for the clover instances, which keep the default)
#ifdef LQCD_F32
#ifndef LQCD_DS_OCC
#define LQCD_DS_OCC 5
#endif
#define LQCD_DS_BOUNDS __launch_bounds__(256, CLOV ? 1 : LQCD_DS_OCC)
#else
#define LQCD_DS_BOUNDS __launch_bounds__(256)
#endif
template <bool DAG, bool R12 = false, bool CLOV = false>
__global__ LQCD_DS_BOUNDS void wilson_dirsplit(KArgs k) {
    __shared__ real2 part[4][12][64];  // 48 KiB
    __shared__ double red[4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int i = chunk * 64 + lane;
    const bool valid = i < k.g.Vh;
    cd acc[12];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
    // the diagonal term's loads are issued first so they are not a third dependent memory round trip after the barrier
    cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    cd cpsi[CLOV ? 12 : 1];
    cd rv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
#ifndef LQCD_UPD_PREFETCH      // fp32 build: the three extra live values spill under its 96-VGPR cap (LQCD_DS_OCC) -- off there (LQCD_UPD_PREFETCH32)
#ifdef LQCD_F32
#ifndef LQCD_UPD_PREFETCH32
#define LQCD_UPD_PREFETCH32 0
#endif
#define LQCD_UPD_PREFETCH LQCD_UPD_PREFETCH32
#else
#define LQCD_UPD_PREFETCH 1
#endif
#endif
    if (LQCD_UPD_PREFETCH && valid && k.upd_scal) {      // CG update mode: the old r is read now, not after the barrier
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ld(k.upd[p] + sp12_off(i) + co12(3 * w + cc));
    }
    if (valid && k.a != 0.0) {
        if constexpr (CLOV) {       // Wilson-clover: all 12 components now (the loads overlap the hops), A xin after the hops
#pragma unroll
            for (int j = 0; j < 12; j++) cpsi[j] = ld(k.xin[p] + sp12_off(i) + co12(j));
        } else {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) xv[cc] = ld(k.xin[p] + sp12_off(i) + co12(3 * w + cc));
        }
    }
    if (valid) {
        switch (w) {
        case 0: dirsplit_hops<0, DAG, R12>(acc, k, p, i); break;
        case 1: dirsplit_hops<1, DAG, R12>(acc, k, p, i); break;
        case 2: dirsplit_hops<2, DAG, R12>(acc, k, p, i); break;
        default: dirsplit_hops<3, DAG, R12>(acc, k, p, i); break;
        }
    }
#pragma unroll
    for (int j = 0; j < 12; j++) part[w][j][lane] = mk2(acc[j].re, acc[j].im);
    if constexpr (CLOV) if (valid && k.a != 0.0) {   // this wave's rows of A xin (its partial sums are already on their way to LDS)
        const real2* __restrict__ ca = k.clover + (((size_t)p * k.g.nch + (size_t)(i >> 6)) * 36) * 64 + (i & 63);
        if (w & 1) clover_rows<1>(xv, ca, cpsi, w >= 2);
        else clover_rows<0>(xv, ca, cpsi, w >= 2);
    }
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
            if (LQCD_UPD_PREFETCH) emit_pre(k, p, co12(j) + sp12_off(i), v, nrm, al_upd, rv[cc]);
            else emit(k, p, co12(j) + sp12_off(i), v, nrm, al_upd);
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

template <int MU, int S>
__device__ inline void project_regs(cd (&h0)[3], cd (&h1)[3], const cd* sp) {
    if constexpr (MU < 3) {
        constexpr int p0 = PERM[MU][0], p1 = PERM[MU][1];
        constexpr int k0 = GK[MU][0] + (S > 0 ? 2 : 0), k1 = GK[MU][1] + (S > 0 ? 2 : 0);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            h0[c] = sp[c] + mul_ipow<k0>(sp[p0 * 3 + c]);
            h1[c] = sp[3 + c] + mul_ipow<k1>(sp[p1 * 3 + c]);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; c++) { h0[c] = 2.0 * sp[c]; h1[c] = 2.0 * sp[3 + c]; }
    }
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
template <bool R12, bool NT>      // NT is a compile-time choice: a run-time branch around the loads would end in register copies that wait for them
__device__ inline void load_link_raw(cd (&u)[9], const real2* __restrict__ U, int Us) {
    constexpr int N = R12 ? 6 : 9;
    const int st = R12 ? 64 : Us;
#pragma unroll
    for (int j = 0; j < N; j++) u[j] = NT ? ld_nt(U + (size_t)j * st) : ld(U + (size_t)j * st);
}
template <bool R12>
__device__ inline void finish_link(cd (&u)[9]) {      // 12-real links: row 2 = conj(row 0 x row 1), the arithmetic of load_link12
    if constexpr (R12) recon_row2(u);
}
__device__ inline void lds_stage_and_barrier(real2 (*nbr)[64], const cd (&stg)[3], int w, int lane) {
    __builtin_amdgcn_sched_barrier(0);     // arithmetic on the loads issued above must not be hoisted in front of the barrier (it would wait for them there)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) nbr[3 * w + cc][lane] = mk2(stg[cc].re, stg[cc].im);
    // LDS writes complete, then the workgroup barrier; global loads issued above stay in flight (no vmcnt wait here)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
// identity the optimiser cannot see through: arithmetic on values loaded BEFORE the barrier must not be scheduled in front of it
// (pure arithmetic is not ordered by the barrier's memory clobber; it would drag the wait for those loads in front of the barrier)
template <int N>
__device__ inline void pin_after_barrier(cd (&a)[N], int n = N) {
#pragma unroll
    for (int j = 0; j < N; j++)
        if (j < n) asm volatile("" : "+v"(a[j].re), "+v"(a[j].im));
}
template <int MU, int S, bool ADJ>
__device__ inline void hop_from_regs(cd (&chi0)[3], cd (&chi1)[3], const cd* sp, const cd (&u)[9], real sign) {
    cd h0[3], h1[3];
    project_regs<MU, S>(h0, h1, sp);
#pragma unroll
    for (int c = 0; c < 3; c++) { h0[c] = sign * h0[c]; h1[c] = sign * h1[c]; }
    su3_mv<ADJ>(chi0, u, h0);
    su3_mv<ADJ>(chi1, u, h1);
}

// ------------------------------------------------------------------------------------------ Wilson, direction-split, both hops in flight
// Variant 8: variant 1 issues the loads of the forward hop, waits, computes, and only then issues the loads of the backward hop -- the
// `if (sign != 0)` around each hop is control flow, and the compiler may not move a load across it -- so every wave pays TWO dependent
// memory round trips.  On an unpartitioned lattice no hop is ever skipped: this variant has no branch in the body, all 36 (t: 24) loads
// of the direction are issued back to back and the arithmetic follows (same operations in the same order as variant 1: bit-identical
// results).  More registers live at the peak (both half-sets of operands).
template <bool R12, bool NT>
__device__ inline void load_link_any(cd (&u)[9], const real2* __restrict__ U, int Us) {
#ifdef LQCD_F32
    if constexpr (R12) {
#pragma unroll
        for (int q = 0; q < 3; q++) ld_pair<NT>(u[2 * q], u[2 * q + 1], U + (size_t)q * 128);
        return;
    }
#endif
    load_link_raw<R12, NT>(u, U, Us);
}
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

// ------------------------------------------------------------------------------------------ staggered
template <bool R12 = false>
__device__ inline void stag_hop(cd (&acc)[3], const real2* __restrict__ psi, const real2* __restrict__ U, int Vh, int Us,
                                real coef, bool adj, bool nt = false) {
    cd h[3], u[9], chi[3];
#pragma unroll
    for (int c = 0; c < 3; c++) h[c] = coef * ld(psi + (size_t)c * Vh);
    if constexpr (R12) load_link12(u, U, nt);
    else { if (nt) load_link_nt(u, U, Us); else load_link(u, U, Us); }
    if (adj) su3_mv<true>(chi, u, h); else su3_mv<false>(chi, u, h);
#pragma unroll
    for (int c = 0; c < 3; c++) acc[c] = acc[c] + chi[c];
}

// eta_mu(n) = (-1)^(x_0+...+x_{mu-1}), global coordinates (local == global parity since extents/origins are even)
__device__ inline real stag_eta(const int c[4], int mu) {
    int e = 0;
    for (int j = 0; j < mu; j++) e += c[j];
    return (e & 1) ? -1.0 : 1.0;
}

template <int TB>
__global__ __launch_bounds__(TB) void staggered_interior(KArgs k) {
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    const int Vh = sp_stride(k.g);  // spinor component stride in elements
    const int i = chunk * TB + threadIdx.x;
    const bool valid = i < k.g.Vh;
    real nrm = 0.0;
    if (valid) {
        Nbr n;
        int c[4];
        neighbours(k.g, p, i, n, c);
        cd acc[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
        const real2* __restrict__ psi = k.in[1 - p];
        const int Us = glink_stride(k.g);
#pragma unroll
        for (int mu = 0; mu < 4; mu++) {
            const real eta = stag_eta(c, mu);
            if (n.sf[mu] != 0.0) stag_hop(acc, psi + sp_off(3, n.fwd[mu]), k.gauge + glink_off(k.g, p, mu, i), Vh, Us, eta * n.sf[mu], false);
            if (n.sb[mu] != 0.0) stag_hop(acc, psi + sp_off(3, n.bwd[mu]), k.gauge + glink_off(k.g, 1 - p, mu, n.bwd[mu]), Vh, Us, -eta * n.sb[mu], true);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
            cd v = k.b * acc[j];
            if (k.a != 0.0) {
                cd xv = ld(k.xin[p] + sp_off(3, i) + (size_t)j * Vh);
                v = mk(fma(k.a, xv.re, v.re), fma(k.a, xv.im, v.im));
            }
            emit(k, p, (size_t)j * Vh + sp_off(3, i), v, nrm, al_upd);
        }
    }
    if (k.norm_partial) block_norm_partial<TB>(nrm, k.norm_partial);
}

// direction-split form (dslash_variant >= 1): 4 waves per 64 sites, wave = direction (forward + backward hop = 18 link + 6
// spinor loads issued as one burst), the four colour-vector partials are combined through LDS and waves 0..2 write one
// colour component each.  Same reasoning as wilson_dirsplit: short-lived, phase-aligned waves keep the 2x link and 8x
// spinor re-use inside the L2 residency time, and the XCD tile sweep of map_block applies to 64-site chunks.
// BOTH (unpartitioned lattices: no hop is ever skipped): no branch in the body, the loads of the forward AND the backward hop are issued
// back to back -- one memory round trip per wave instead of two -- and the arithmetic follows in stag_hop's order (bit-identical results).
// The staggered kernel is light on registers, so unlike the Wilson variant 8 this costs little occupancy.
template <bool R12, bool BOTH = false, bool NTB = false>
__global__ __launch_bounds__(256) void staggered_dirsplit(KArgs k) {
    __shared__ real2 part[4][3][64];
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
    cd acc[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    cd xv = mk(0, 0);
    if (valid && k.a != 0.0 && w < 3) xv = ld(k.xin[p] + sp_off(3, i) + (size_t)w * Vh);
    cd rv = mk(0, 0);
    if (valid && k.upd_scal && w < 3) rv = ld(k.upd[p] + sp_off(3, i) + (size_t)w * Vh);    // CG update mode: the old r is read ahead of the hops
    if constexpr (BOTH) {
        const int ic = valid ? i : 0;      // lanes past the end work on site 0 and do not store
        Nbr n;
        int c[4];
        neighbours(k.g, p, ic, n, c);
        const real2* __restrict__ psi = k.in[1 - p];
        const int Us = glink_stride(k.g);
        const real eta = stag_eta(c, w);
        const int nf = w == 0 ? n.fwd[0] : w == 1 ? n.fwd[1] : w == 2 ? n.fwd[2] : n.fwd[3];
        const int nb = w == 0 ? n.bwd[0] : w == 1 ? n.bwd[1] : w == 2 ? n.bwd[2] : n.bwd[3];
        const real sf = w == 0 ? n.sf[0] : w == 1 ? n.sf[1] : w == 2 ? n.sf[2] : n.sf[3];
        const real sb = w == 0 ? n.sb[0] : w == 1 ? n.sb[1] : w == 2 ? n.sb[2] : n.sb[3];
        const real2* __restrict__ Uf = R12 ? k.gauge12 + gl12_off(k.g, p, w, ic) : k.gauge + glink_off(k.g, p, w, ic);
        const real2* __restrict__ Ub = R12 ? k.gauge12 + gl12_off(k.g, 1 - p, w, nb) : k.gauge + glink_off(k.g, 1 - p, w, nb);
        cd hf[3], hb[3], uf[9], ub[9], chi[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) hf[cc] = ld(psi + sp_off(3, nf) + (size_t)cc * Vh);
        load_link_any<R12, false>(uf, Uf, Us);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) hb[cc] = ld(psi + sp_off(3, nb) + (size_t)cc * Vh);
        load_link_any<R12, NTB>(ub, Ub, Us);
        const real cf = eta * sf, cb = -eta * sb;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) hf[cc] = cf * hf[cc];
        finish_link<R12>(uf);
        su3_mv<false>(chi, uf, hf);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) acc[cc] = acc[cc] + chi[cc];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) hb[cc] = cb * hb[cc];
        finish_link<R12>(ub);
        su3_mv<true>(chi, ub, hb);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) acc[cc] = acc[cc] + chi[cc];
    } else
    if (valid) {
        Nbr n;
        int c[4];
        neighbours(k.g, p, i, n, c);
        const real2* __restrict__ psi = k.in[1 - p];
        const int Us = glink_stride(k.g);
        const real eta = stag_eta(c, w);
        const int nf = w == 0 ? n.fwd[0] : w == 1 ? n.fwd[1] : w == 2 ? n.fwd[2] : n.fwd[3];
        const int nb = w == 0 ? n.bwd[0] : w == 1 ? n.bwd[1] : w == 2 ? n.bwd[2] : n.bwd[3];
        const real sf = w == 0 ? n.sf[0] : w == 1 ? n.sf[1] : w == 2 ? n.sf[2] : n.sf[3];
        const real sb = w == 0 ? n.sb[0] : w == 1 ? n.sb[1] : w == 2 ? n.sb[2] : n.sb[3];
        const real2* __restrict__ Uf = R12 ? k.gauge12 + gl12_off(k.g, p, w, i) : k.gauge + glink_off(k.g, p, w, i);
        const real2* __restrict__ Ub = R12 ? k.gauge12 + gl12_off(k.g, 1 - p, w, nb) : k.gauge + glink_off(k.g, 1 - p, w, nb);
        if (sf != 0.0) stag_hop<R12>(acc, psi + sp_off(3, nf), Uf, Vh, Us, eta * sf, false, (k.nt & 2) != 0);
        if (sb != 0.0) stag_hop<R12>(acc, psi + sp_off(3, nb), Ub, Vh, Us, -eta * sb, true, (k.nt & 1) != 0);
    }
#pragma unroll
    for (int j = 0; j < 3; j++) part[w][j][lane] = mk2(acc[j].re, acc[j].im);
    __syncthreads();
    real nrm = 0.0;
    if (valid && w < 3) {
        const real2 s0 = part[0][w][lane], s1 = part[1][w][lane], s2 = part[2][w][lane], s3 = part[3][w][lane];
        cd v = k.b * mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
        v = mk(fma(k.a, xv.re, v.re), fma(k.a, xv.im, v.im));
        emit_pre(k, p, (size_t)w * Vh + sp_off(3, i), v, nrm, al_upd, rv);
    }
    if (k.norm_partial) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_down(nrm, off, 64);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ------------------------------------------------------------------------------------------ halo: pack
// blockIdx.y = 2*mu + side.  side 0: lower face (x_mu = 0) -> send_bwd[mu] = P psi   (receiver's forward hop)
//                            side 1: upper face (x_mu = L-1) -> send_fwd[mu] = U^+ P psi (receiver's backward hop)
// Buffers are [slot][ncomp_half][Fh] with slot = output parity of the RECEIVING site (0 when a single parity is computed).
// Everything is templated on MU and per-lane parities select pointers with ?: -- a per-lane run-time index into the
// by-value argument struct would force a private (scratch) copy of it.
template <int MU>
__device__ inline void wilson_pack_dir(const HArgs& k, int side) {
    const Geom& g = k.g;
    if (!g.part[MU]) return;
    const int Fh = g.Vh / g.L[MU], Vh = sp_stride(g);
    const int nslots = k.parity_mode == 2 ? 2 : 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslots * Fh) return;
    const int slot = t / Fh, f = t - slot * Fh;
    const int pout = k.parity_mode == 2 ? slot : k.parity_mode;
    const int ps = 1 - pout;  // parity of the site being packed
    int c[4];
    face_to_coords(g, MU, side ? g.L[MU] - 1 : 0, ps, f, c);
    const int i = coords_to_cb(g, c);
    const real2* __restrict__ psi = (ps ? k.in[1] : k.in[0]) + sp12_off(i);
    cd h0[3], h1[3];
    real2* dst;
    if (side == 0) {
        // receiver forward hop uses (1 - SF gamma), SF = dagger ? -1 : +1
        if (k.dagger) project<MU, -1>(h0, h1, psi, Vh); else project<MU, 1>(h0, h1, psi, Vh);
        dst = k.send_bwd[MU];
    } else {
        // receiver backward hop uses (1 + SF gamma) and U^+ of the sender's link
        if (k.dagger) project<MU, 1>(h0, h1, psi, Vh); else project<MU, -1>(h0, h1, psi, Vh);
        cd u[9], x0[3], x1[3];
        load_link(u, k.gauge + glink_off(g, ps, MU, i), glink_stride(g));
        su3_mv<true>(x0, u, h0);
        su3_mv<true>(x1, u, h1);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { h0[cc] = x0[cc]; h1[cc] = x1[cc]; }
        dst = k.send_fwd[MU];
    }
    dst += (size_t)slot * 6 * Fh + f;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
        st(dst + (size_t)cc * Fh, h0[cc]);
        st(dst + (size_t)(3 + cc) * Fh, h1[cc]);
    }
}

__global__ __launch_bounds__(128) void wilson_pack(HArgs k) {
    const int side = blockIdx.y & 1;
    switch (blockIdx.y >> 1) {
    case 0: wilson_pack_dir<0>(k, side); break;
    case 1: wilson_pack_dir<1>(k, side); break;
    case 2: wilson_pack_dir<2>(k, side); break;
    default: wilson_pack_dir<3>(k, side); break;
    }
}

// ------------------------------------------------------------------------------------------ halo: exterior (fused)
// out(n) += b * (all hop contributions that crossed a rank boundary).  ONE launch: blockIdx.y = 2*mu + side enumerates the
// faces; a boundary site that lies on several faces (edges, corners) is OWNED by its lowest partitioned direction (and,
// there, by its face), whose thread gathers the contributions of every face the site touches and does a single
// read-modify-write -- no inter-direction ordering, no atomics, deterministic.
template <int NU, bool DAG>
__device__ __forceinline__ void wilson_ext_add(cd (&acc)[12], const HArgs& k, const int (&c)[4], int slot, int pout, int i) {
    const Geom& g = k.g;
    if (!g.part[NU]) return;
    const int Fh = g.Vh / g.L[NU], Vh = sp_stride(g);
    if (c[NU] == g.L[NU] - 1) {
        // forward hop at the upper face: ghost = P psi(n+nu) from the +nu neighbour; multiply by own U_nu(n)
        const int f = coords_to_face(g, NU, c);
        const real2* __restrict__ src = k.recv_fwd[NU] + (size_t)slot * 6 * Fh + f;
        const real sg = k.sign_fwd[NU];
        cd h0[3], h1[3], u[9], x0[3], x1[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            h0[cc] = sg * ld(src + (size_t)cc * Fh);
            h1[cc] = sg * ld(src + (size_t)(3 + cc) * Fh);
        }
        load_link(u, k.gauge + glink_off(g, pout, NU, i), glink_stride(g));
        su3_mv<false>(x0, u, h0);
        su3_mv<false>(x1, u, h1);
        reconstruct<NU, DAG ? -1 : 1>(acc, x0, x1);
    }
    if (c[NU] == 0) {
        // backward hop at the lower face: ghost = U^+ P psi(n-nu) from the -nu neighbour
        const int f = coords_to_face(g, NU, c);
        const real2* __restrict__ src = k.recv_bwd[NU] + (size_t)slot * 6 * Fh + f;
        const real sg = k.sign_bwd[NU];
        cd h0[3], h1[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            h0[cc] = sg * ld(src + (size_t)cc * Fh);
            h1[cc] = sg * ld(src + (size_t)(3 + cc) * Fh);
        }
        reconstruct<NU, DAG ? 1 : -1>(acc, h0, h1);
    }
}

// true if the site is already owned by a face of a lower partitioned direction, or (same direction) by the lower face
template <int MU>
__device__ inline bool ext_not_owner(const Geom& g, const int (&c)[4], int side) {
    bool lower = false;
    if (MU > 0) lower = lower || (g.part[0] && (c[0] == 0 || c[0] == g.L[0] - 1));
    if (MU > 1) lower = lower || (g.part[1] && (c[1] == 0 || c[1] == g.L[1] - 1));
    if (MU > 2) lower = lower || (g.part[2] && (c[2] == 0 || c[2] == g.L[2] - 1));
    // extent 2 in direction MU: a site cannot be on both faces, nothing to do; side is only used for clarity
    (void)side;
    return lower;
}

template <int MU, bool DAG>
__device__ __forceinline__ real wilson_ext_face(const HArgs& k, int side) {
    const Geom& g = k.g;
    if (!g.part[MU]) return 0.0;
    const int Fh = g.Vh / g.L[MU], Vh = sp_stride(g);
    const int nslots = k.parity_mode == 2 ? 2 : 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslots * Fh) return 0.0;
    const int slot = t / Fh, f = t - slot * Fh;
    const int pout = k.parity_mode == 2 ? slot : k.parity_mode;
    int c[4];
    face_to_coords(g, MU, side ? g.L[MU] - 1 : 0, pout, f, c);
    if (ext_not_owner<MU>(g, c, side)) return 0.0;
    const int i = coords_to_cb(g, c);
    cd acc[12];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
    if (MU <= 0) wilson_ext_add<0, DAG>(acc, k, c, slot, pout, i);
    if (MU <= 1) wilson_ext_add<1, DAG>(acc, k, c, slot, pout, i);
    if (MU <= 2) wilson_ext_add<2, DAG>(acc, k, c, slot, pout, i);
    wilson_ext_add<3, DAG>(acc, k, c, slot, pout, i);
    const real coef = k.upd_scal ? -k.upd_scal[S_ALPHA] * k.b : k.b;
    real2* __restrict__ o = (k.upd_scal ? (pout ? k.upd[1] : k.upd[0]) : (pout ? k.out[1] : k.out[0])) + sp12_off(i);
    real corr = 0.0;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        cd v = ld(o + co12(j));
        const real before = v.re * v.re + v.im * v.im;
        v.re = fma(coef, acc[j].re, v.re);
        v.im = fma(coef, acc[j].im, v.im);
        corr += (v.re * v.re + v.im * v.im) - before;
        st(o + co12(j), v);
    }
    return corr;
}

// block-level sum of the per-thread norm corrections of an exterior kernel (128 threads)
__device__ inline void ext_partial(const HArgs& k, real corr) {
    if (!k.norm_partial) return;
    __shared__ double red[2];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) corr += __shfl_down(corr, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = corr;
    __syncthreads();
    if (threadIdx.x == 0) k.norm_partial[k.partial_offset + blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1];
}

template <bool DAG>
__global__ __launch_bounds__(128) void wilson_exterior(HArgs k) {
    if (k.upd_scal && k.upd_scal[S_DONE] != 0.0) return;
    const int side = blockIdx.y & 1;
    real corr;
    switch (blockIdx.y >> 1) {
    case 0: corr = wilson_ext_face<0, DAG>(k, side); break;
    case 1: corr = wilson_ext_face<1, DAG>(k, side); break;
    case 2: corr = wilson_ext_face<2, DAG>(k, side); break;
    default: corr = wilson_ext_face<3, DAG>(k, side); break;
    }
    ext_partial(k, corr);
}

// ------------------------------------------------------------------------------------------ staggered halos
// 3 components; eta and the +/- sign are applied by the receiver
template <int MU>
__device__ inline void staggered_pack_dir(const HArgs& k, int side) {
    const Geom& g = k.g;
    if (!g.part[MU]) return;
    const int Fh = g.Vh / g.L[MU], Vh = sp_stride(g);
    const int nslots = k.parity_mode == 2 ? 2 : 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslots * Fh) return;
    const int slot = t / Fh, f = t - slot * Fh;
    const int pout = k.parity_mode == 2 ? slot : k.parity_mode;
    const int ps = 1 - pout;
    int c[4];
    face_to_coords(g, MU, side ? g.L[MU] - 1 : 0, ps, f, c);
    const int i = coords_to_cb(g, c);
    cd h[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) h[cc] = ld((ps ? k.in[1] : k.in[0]) + sp_off(3, i) + (size_t)cc * Vh);
    real2* dst;
    if (side == 0) {
        dst = k.send_bwd[MU];
    } else {
        cd u[9], x[3];
        load_link(u, k.gauge + glink_off(g, ps, MU, i), glink_stride(g));
        su3_mv<true>(x, u, h);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) h[cc] = x[cc];
        dst = k.send_fwd[MU];
    }
    dst += (size_t)slot * 3 * Fh + f;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) st(dst + (size_t)cc * Fh, h[cc]);
}

__global__ __launch_bounds__(128) void staggered_pack(HArgs k) {
    const int side = blockIdx.y & 1;
    switch (blockIdx.y >> 1) {
    case 0: staggered_pack_dir<0>(k, side); break;
    case 1: staggered_pack_dir<1>(k, side); break;
    case 2: staggered_pack_dir<2>(k, side); break;
    default: staggered_pack_dir<3>(k, side); break;
    }
}

template <int NU>
__device__ __forceinline__ void staggered_ext_add(cd (&acc)[3], const HArgs& k, const int (&c)[4], int slot, int pout, int i) {
    const Geom& g = k.g;
    if (!g.part[NU]) return;
    const int Fh = g.Vh / g.L[NU], Vh = sp_stride(g);
    int e = 0;
#pragma unroll
    for (int j = 0; j < NU; j++) e += c[j];
    const real eta = (e & 1) ? -1.0 : 1.0;
    if (c[NU] == g.L[NU] - 1) {
        const int f = coords_to_face(g, NU, c);
        const real2* src = k.recv_fwd[NU] + (size_t)slot * 3 * Fh + f;
        const real cf = eta * k.sign_fwd[NU];
        cd h[3], u[9], x[3];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) h[cc] = cf * ld(src + (size_t)cc * Fh);
        load_link(u, k.gauge + glink_off(g, pout, NU, i), glink_stride(g));
        su3_mv<false>(x, u, h);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) acc[cc] = acc[cc] + x[cc];
    }
    if (c[NU] == 0) {
        const int f = coords_to_face(g, NU, c);
        const real2* src = k.recv_bwd[NU] + (size_t)slot * 3 * Fh + f;
        const real cf = -eta * k.sign_bwd[NU];
#pragma unroll
        for (int cc = 0; cc < 3; cc++) acc[cc] = acc[cc] + cf * ld(src + (size_t)cc * Fh);
    }
}

template <int MU>
__device__ inline real staggered_ext_face(const HArgs& k, int side) {
    const Geom& g = k.g;
    if (!g.part[MU]) return 0.0;
    const int Fh = g.Vh / g.L[MU], Vh = sp_stride(g);
    const int nslots = k.parity_mode == 2 ? 2 : 1;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslots * Fh) return 0.0;
    const int slot = t / Fh, f = t - slot * Fh;
    const int pout = k.parity_mode == 2 ? slot : k.parity_mode;
    int c[4];
    face_to_coords(g, MU, side ? g.L[MU] - 1 : 0, pout, f, c);
    if (ext_not_owner<MU>(g, c, side)) return 0.0;
    const int i = coords_to_cb(g, c);
    cd acc[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    if (MU <= 0) staggered_ext_add<0>(acc, k, c, slot, pout, i);
    if (MU <= 1) staggered_ext_add<1>(acc, k, c, slot, pout, i);
    if (MU <= 2) staggered_ext_add<2>(acc, k, c, slot, pout, i);
    staggered_ext_add<3>(acc, k, c, slot, pout, i);
    const real coef = k.upd_scal ? -k.upd_scal[S_ALPHA] * k.b : k.b;
    real2* o = (k.upd_scal ? (pout ? k.upd[1] : k.upd[0]) : (pout ? k.out[1] : k.out[0])) + sp_off(3, i);
    real corr = 0.0;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        cd v = ld(o + (size_t)j * Vh);
        const real before = v.re * v.re + v.im * v.im;
        v.re = fma(coef, acc[j].re, v.re);
        v.im = fma(coef, acc[j].im, v.im);
        corr += (v.re * v.re + v.im * v.im) - before;
        st(o + (size_t)j * Vh, v);
    }
    return corr;
}

__global__ __launch_bounds__(128) void staggered_exterior(HArgs k) {
    if (k.upd_scal && k.upd_scal[S_DONE] != 0.0) return;
    const int side = blockIdx.y & 1;
    real corr;
    switch (blockIdx.y >> 1) {
    case 0: corr = staggered_ext_face<0>(k, side); break;
    case 1: corr = staggered_ext_face<1>(k, side); break;
    case 2: corr = staggered_ext_face<2>(k, side); break;
    default: corr = staggered_ext_face<3>(k, side); break;
    }
    ext_partial(k, corr);
}

// ------------------------------------------------------------------------------------------ host launchers
static KArgs make_kargs(lqcd_ctx_s* c, const StencilCall& s, int TB) {
    KArgs k;
    k.g = c->geom;
    k.gauge = (const real2*)s.gauge;
    k.gauge12 = (const real2*)s.gauge12;   // elements of this build's precision (the caller matches prec)
    k.clover = (const real2*)s.clover;     // elements of this build's precision (the caller matches prec)
    for (int p = 0; p < 2; p++) { k.out[p] = (real2*)s.out[p]; k.in[p] = (const real2*)s.in[p]; k.xin[p] = (const real2*)s.xin[p]; }
    k.a = s.a; k.b = s.b; k.r = s.r;
    k.parity_mode = s.parity_mode;
    const int chunks = (c->geom.Vh + TB - 1) / TB;
    k.nblocks = chunks * (s.parity_mode == 2 ? 2 : 1);
    k.remap = c->tun.xcd_remap;
    const int slice = c->geom.XH * c->geom.L[1] * c->geom.L[2];  // sites per parity per t-slice
    k.nsub = (c->tun.xcd_nsub >= 8 && c->tun.xcd_nsub % 8 == 0) ? c->tun.xcd_nsub : 8;
    k.cps = (slice % TB == 0 && (slice / TB) % k.nsub == 0) ? slice / TB : 0;
    const int plane = c->geom.XH * c->geom.L[1];
    k.cpp = (plane % TB == 0) ? plane / TB : 0;
    k.ysplit = 1;
#ifdef LQCD_ABLATE
    k.dbg = c->tun.dbg;
#endif
    k.nt = (c->tun.nt_gauge & 3) | (c->tun.nt_store ? 4 : 0) | ((c->tun.nt_gauge & 4) ? 8 : 0);
    const int ys = c->tun.xcd_ysplit;
    if (ys > 1 && k.cps > 0 && k.cpp > 0 && k.cpp % ys == 0 && k.nsub % ys == 0 && c->geom.L[2] % (k.nsub / ys) == 0) k.ysplit = ys;
    k.cpr = k.cps > 0 ? k.cps / k.nsub : 1;
    k.ty = k.ysplit > 1 ? k.cpp / k.ysplit : 1;
    k.tz = k.cpr / k.ty;
    k.d_perpass = make_fastdiv(std::max(1, k.cpr * c->geom.L[3]));
    k.d_cpr = make_fastdiv(std::max(1, k.cpr));
    k.d_ysplit = make_fastdiv(std::max(1, k.ysplit));
    k.d_ty = make_fastdiv(std::max(1, k.ty));
    k.norm_partial = s.norm_partial;
    k.upd_scal = s.upd_scal;
    k.upd[0] = (real2*)s.upd[0]; k.upd[1] = (real2*)s.upd[1];
    k.skip = s.skip_flag;
    k.alpha_partials = s.alpha_partials; k.alpha_n = s.alpha_n; k.scal_w = s.scal_w;
    return k;
}

static bool use_dirsplit(lqcd_ctx_s* c, int kind, real r) {   // variants 1/2/3 work on 64-site chunks
    if (!(c->tun.dslash_variant >= 1 && c->tun.dslash_variant <= 8)) return false;
    // Wilson: the split kernels use the r = 1 projectors.  On a partitioned lattice a general-r application runs as two r = 1 calls
    // (apply.hip, split_general_r), so the launch geometry (number of |.|^2 partials) is the r = 1 one there for every r.
    return kind == LQCD_STAGGERED || r == 1.0 || c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3];
}
static int persist_grid(lqcd_ctx_s* c, int nvirt) {
    int g = c->num_cu * (c->tun.persist_per_cu > 0 ? c->tun.persist_per_cu : 2);
    g -= g % 8;
    if (g < 8) g = 8;
    return std::min(g, nvirt);
}


template <int TB>
static int launch_interior_tb(lqcd_ctx_s* c, const StencilCall& s) {
    KArgs k = make_kargs(c, s, TB);
    dim3 grid(k.nblocks), block(TB);
    const size_t pad = (size_t)c->tun.lds_pad_kb * 1024;  // occupancy limiter (experiments)
    if (s.kind == LQCD_WILSON) {
        const bool rgen = (s.r != 1.0);
        if (!rgen) {
            if (s.dagger) hipLaunchKernelGGL((wilson_interior<TB, true, false>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_interior<TB, false, false>), grid, block, pad, c->stream, k);
        } else {
            if (s.dagger) hipLaunchKernelGGL((wilson_interior<TB, true, true>), grid, block, 0, c->stream, k);
            else hipLaunchKernelGGL((wilson_interior<TB, false, true>), grid, block, 0, c->stream, k);
        }
    } else {
        hipLaunchKernelGGL((staggered_interior<TB>), grid, block, 0, c->stream, k);
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int launch_stencil_interior(lqcd_ctx_s* c, const StencilCall& s) {
    if (use_dirsplit(c, s.kind, s.r)) {
        KArgs k = make_kargs(c, s, 64);
        const size_t pad = (size_t)c->tun.lds_pad_kb * 1024;
        if (s.kind == LQCD_STAGGERED) {
            const bool both = c->tun.stag_both && !(c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3]);
            const dim3 sg(k.nblocks), sb_(256);
            if (both) {
                const bool ntb = (k.nt & 1) != 0;
                if (k.gauge12) { if (ntb) hipLaunchKernelGGL((staggered_dirsplit<true, true, true>), sg, sb_, pad, c->stream, k);
                                 else hipLaunchKernelGGL((staggered_dirsplit<true, true, false>), sg, sb_, pad, c->stream, k); }
                else { if (ntb) hipLaunchKernelGGL((staggered_dirsplit<false, true, true>), sg, sb_, pad, c->stream, k);
                       else hipLaunchKernelGGL((staggered_dirsplit<false, true, false>), sg, sb_, pad, c->stream, k); }
            }
            else if (k.gauge12) hipLaunchKernelGGL((staggered_dirsplit<true>), sg, sb_, pad, c->stream, k);
            else hipLaunchKernelGGL((staggered_dirsplit<false>), sg, sb_, pad, c->stream, k);
        } else if (c->tun.dslash_variant == 8 && !k.clover && !(c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3])) {
            dim3 grid(k.nblocks), block(256);
            const bool ntb = (k.nt & 1) != 0;
#define LQ_V8(D, R) do { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_both<D, R, true>), grid, block, pad, c->stream, k); \
                         else hipLaunchKernelGGL((wilson_dirsplit_both<D, R, false>), grid, block, pad, c->stream, k); } while (0)
            if (k.gauge12) { if (s.dagger) LQ_V8(true, true); else LQ_V8(false, true); }
            else { if (s.dagger) LQ_V8(true, false); else LQ_V8(false, false); }
#undef LQ_V8
#ifndef LQCD_F32   // the fp32 build (paired-component fields, see sp12_off) has the direction-split and site-per-lane kernels only;
                   // the mixed-precision solver pins dslash_variant to 0/1 for the duration of a solve (mixed.hip)
        } else if (c->tun.dslash_variant == 7 && !k.clover && k.gauge12 && s.parity_mode == 2) {
            dim3 grid(k.nblocks / 2), block(512);
            if (s.dagger) hipLaunchKernelGGL((wilson_pair4<true, true>), grid, block, pad, c->stream, k);
            else hipLaunchKernelGGL((wilson_pair4<false, true>), grid, block, pad, c->stream, k);
        } else if (c->tun.dslash_variant == 6 && !k.clover && 64 % c->geom.XH == 0 && c->geom.XH <= 32) {
            dim3 grid(k.nblocks), block(256);
            const bool ntb = (k.nt & 1) != 0;     // backward-link loads non-temporal (tunable nt_gauge bit 0); bit 1 is not offered by this variant
#define LQ_V6(D, R) do { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_lds<D, R, true>), grid, block, pad, c->stream, k); \
                         else hipLaunchKernelGGL((wilson_dirsplit_lds<D, R, false>), grid, block, pad, c->stream, k); } while (0)
            if (k.gauge12) { if (s.dagger) LQ_V6(true, true); else LQ_V6(false, true); }
            else { if (s.dagger) LQ_V6(true, false); else LQ_V6(false, false); }
#undef LQ_V6
        } else if (c->tun.dslash_variant == 5 && !k.clover) {
            dim3 grid(k.nblocks), block(256);
            if (k.gauge12) {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit4<true, true>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit4<false, true>), grid, block, pad, c->stream, k);
            } else {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit4<true, false>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit4<false, false>), grid, block, pad, c->stream, k);
            }
        } else if (c->tun.dslash_variant == 4 && !k.clover) {
            dim3 grid(k.nblocks), block(256);
            if (k.gauge12) {
                if (s.dagger) hipLaunchKernelGGL((wilson_lanesplit<true, true>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_lanesplit<false, true>), grid, block, pad, c->stream, k);
            } else {
                if (s.dagger) hipLaunchKernelGGL((wilson_lanesplit<true, false>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_lanesplit<false, false>), grid, block, pad, c->stream, k);
            }
        } else if (c->tun.dslash_variant == 3) {
            dim3 grid(persist_grid(c, k.nblocks)), block(512);
            if (s.dagger) hipLaunchKernelGGL((wilson_hopsplit_persist<true>), grid, block, pad, c->stream, k, k.nblocks);
            else hipLaunchKernelGGL((wilson_hopsplit_persist<false>), grid, block, pad, c->stream, k, k.nblocks);
        } else if (c->tun.dslash_variant == 2) {
            dim3 grid(k.nblocks), block(512);
            const int nt = (c->tun.nt_gauge ? 1 : 0) | (c->tun.nt_store ? 2 : 0);
#define LQ_HS(D, N) hipLaunchKernelGGL((wilson_hopsplit<D, N>), grid, block, pad, c->stream, k)
            if (s.dagger) { switch (nt) { case 1: LQ_HS(true, 1); break; case 2: LQ_HS(true, 2); break; case 3: LQ_HS(true, 3); break; default: LQ_HS(true, 0); } }
            else { switch (nt) { case 1: LQ_HS(false, 1); break; case 2: LQ_HS(false, 2); break; case 3: LQ_HS(false, 3); break; default: LQ_HS(false, 0); } }
#undef LQ_HS
#endif
        } else {
            dim3 grid(k.nblocks), block(256);
            if (k.clover && k.gauge12) {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, true, true>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit<false, true, true>), grid, block, pad, c->stream, k);
            } else if (k.clover) {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, false, true>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit<false, false, true>), grid, block, pad, c->stream, k);
            } else if (k.gauge12) {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, true>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit<false, true>), grid, block, pad, c->stream, k);
            } else {
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, false>), grid, block, pad, c->stream, k);
                else hipLaunchKernelGGL((wilson_dirsplit<false, false>), grid, block, pad, c->stream, k);
            }
        }
        HIPCHK(hipGetLastError());
        return LQCD_OK;
    }
    switch (c->tun.dslash_block) {
    case 64: return launch_interior_tb<64>(c, s);
    case 256: return launch_interior_tb<256>(c, s);
    default: return launch_interior_tb<128>(c, s);
    }
}

static int max_face_threads(lqcd_ctx_s* c, int parity_mode);

static HArgs make_hargs(lqcd_ctx_s* c, const StencilCall& s) {
    HArgs h;
    h.g = c->geom;
    h.gauge = (const real2*)s.gauge;
    for (int p = 0; p < 2; p++) { h.out[p] = (real2*)s.out[p]; h.in[p] = (const real2*)s.in[p]; }
    h.b = s.b;
    h.parity_mode = s.parity_mode;
    h.dagger = s.dagger;
    for (int mu = 0; mu < 4; mu++) {
        // [send_fwd | send_bwd] and [recv_bwd | recv_fwd] are packed back to back for THIS call's message size (elements of this
        // build's precision), so that a pair of faces bound for the same rank is one contiguous message (ops.hip)
        const size_t cnt = (size_t)(s.parity_mode == 2 ? 2 : 1) * (s.kind == LQCD_WILSON ? 6 : 3) * face_half_sites(c->geom, mu);
        h.send_fwd[mu] = (real2*)c->send_fwd[mu]; h.send_bwd[mu] = (real2*)c->send_fwd[mu] + cnt;
        h.recv_bwd[mu] = (const real2*)c->recv_bwd[mu]; h.recv_fwd[mu] = (const real2*)c->recv_bwd[mu] + cnt;
        h.sign_fwd[mu] = (c->coord[mu] == c->pe[mu] - 1) ? c->geom.bc_fwd[mu] : 1.0;
        h.sign_bwd[mu] = (c->coord[mu] == 0) ? c->geom.bc_bwd[mu] : 1.0;
    }
    h.norm_partial = s.norm_partial;
    h.partial_offset = stencil_num_blocks(c, s.kind, s.r, s.parity_mode);   // corrections are appended to the interior's partials
    h.upd_scal = s.upd_scal;
    h.upd[0] = (real2*)s.upd[0]; h.upd[1] = (real2*)s.upd[1];
    return h;
}

static int max_face_threads(lqcd_ctx_s* c, int parity_mode) {
    int m = 0;
    for (int mu = 0; mu < 4; mu++)
        if (c->geom.part[mu]) m = std::max(m, face_half_sites(c->geom, mu));
    return m * (parity_mode == 2 ? 2 : 1);
}

int launch_stencil_pack(lqcd_ctx_s* c, const StencilCall& s) {
    const int nt = max_face_threads(c, s.parity_mode);
    if (nt == 0) return LQCD_OK;
    HArgs h = make_hargs(c, s);
    dim3 grid((nt + 127) / 128, 8), block(128);
    if (s.kind == LQCD_WILSON) hipLaunchKernelGGL(wilson_pack, grid, block, 0, c->stream, h);
    else hipLaunchKernelGGL(staggered_pack, grid, block, 0, c->stream, h);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int launch_stencil_exterior(lqcd_ctx_s* c, const StencilCall& s) {
    const int nt = max_face_threads(c, s.parity_mode);
    if (nt == 0) return LQCD_OK;
    HArgs h = make_hargs(c, s);
    dim3 grid((nt + 127) / 128, 8), block(128);
    if (s.kind == LQCD_WILSON) {
        if (s.dagger) hipLaunchKernelGGL(wilson_exterior<true>, grid, block, 0, c->stream, h);
        else hipLaunchKernelGGL(wilson_exterior<false>, grid, block, 0, c->stream, h);
    }
    else hipLaunchKernelGGL(staggered_exterior, grid, block, 0, c->stream, h);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

}  // inline namespace (precision)

#ifndef LQCD_F32
// precision-independent launch geometry (shared by both builds of this file)
// number of |.|^2 block partials the interior kernel writes
int stencil_num_blocks(lqcd_ctx_s* c, int kind, double r, int parity_mode) {
    const int TB = use_dirsplit(c, kind, r) ? 64 : c->tun.dslash_block;
    const int nvirt = ((c->geom.Vh + TB - 1) / TB) * (parity_mode == 2 ? 2 : 1);
    if (use_dirsplit(c, kind, r) && kind == LQCD_WILSON && c->tun.dslash_variant == 3) return persist_grid(c, nvirt);
    return nvirt;
}
// interior block partials + (partitioned lattice) the exterior kernel's correction partials
int stencil_num_partials(lqcd_ctx_s* c, int kind, double r, int parity_mode) {
    const int nt = max_face_threads(c, parity_mode);
    return stencil_num_blocks(c, kind, r, parity_mode) + (nt > 0 ? ((nt + 127) / 128) * 8 : 0);
}
#endif

}  // namespace lqcd
