// stencil_common.h -- argument structs, addressing, load/store and SU(3) helpers shared by the stencil translation units
// (stencil.hip: the default kernels, both precisions; experiments/stencil_alt/stencil_alt.hip: the opt-in Wilson variants 2-8, fp64, experiment builds only).  Everything lives in the
// precision namespace (lqcd::p64 / lqcd::p32) the including file is compiled for.
#pragma once
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
    const real2* dotz2[2];    // StencilCall::dot_z2 (scalar-addressing kernel only)
    const real2* dotz[2];     // dot mode (StencilCall::dot_z): Re / Im <z, out> and |out|^2 per workgroup -> dot_partial[3 b ..]; dot_conj: <out, z> instead
    double* dot_partial;
    int dot_conj;
    const int* vlist;         // folded bulk / boundary launches: virtual block of workgroup b (the launch covers only the chunks of its kind, in the order of the map); null: b itself
    int fsel;                 // folded launches (round 6, overlapping schedules): 0 every chunk, 1 only the chunks with no site on a partitioned face ("bulk": runs
                              // beside the exchange), 2 only the others ("boundary": after arrival) -- the two launches write disjoint sites and disjoint |.|^2 partials
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
    // fused tail of the launch (partitioned CG, critical path at small local volumes): the LAST block to finish sums every |.|^2 partial
    // of this application (interior + exterior corrections, red_n of them) into red_out -- no separate one-block reduction launch
    double* red_out;
    unsigned* red_ctr;        // arrival counter, zero between launches (reset by the last block)
    int red_n;
    // >= 0: every owner thread also packs the faces of the site it just finished for the NEXT application (its dagger flag) -- the
    // following stencil call skips its pack launch (StencilCall::prepacked)
    int pack_next;
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

__device__ inline int vblock_of(const KArgs& k) { return k.vlist ? k.vlist[blockIdx.x] : (int)blockIdx.x; }      // (workgroup-uniform: a scalar load)
__device__ inline void map_block(const KArgs& k, int& chunk, int& p) { map_block_v(k, vblock_of(k), chunk, p); }

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
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < TB / 64; w++) s += red[w];
        partial[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------ register-level hop helpers
// (operands already in registers: used where the loads of a hop are issued apart from its arithmetic)
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

// "12 + delta" links (fp64 build): rows 0, 1 in fp64 and delta = row 2 - conj(row 0 x row 1) in fp32, three complex floats in words 6, 7 of the 8-word link
// (word 6 = delta_0, delta_1; word 7 = delta_2, unused).  For a link within 1e-9 of SU(3) -- the reference's text and ILDG configurations: 8.8e-11 -- the fp32
// rounding of delta is below 1e-16 of the O(1) matrix elements, i.e. row 2 comes back to fp64 rounding: 128 B per link instead of 144.
#ifndef LQCD_F32
__device__ inline void add_delta_row2(cd (&u)[9], const cd (&dw)[2]) {
    const float d0r = __int_as_float(__double2loint(dw[0].re)), d0i = __int_as_float(__double2hiint(dw[0].re));
    const float d1r = __int_as_float(__double2loint(dw[0].im)), d1i = __int_as_float(__double2hiint(dw[0].im));
    const float d2r = __int_as_float(__double2loint(dw[1].re)), d2i = __int_as_float(__double2hiint(dw[1].re));
    u[6].re += (double)d0r; u[6].im += (double)d0i;
    u[7].re += (double)d1r; u[7].im += (double)d1i;
    u[8].re += (double)d2r; u[8].im += (double)d2i;
}
#endif

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

// pack the faces a finished site lies on, from its 12 components in registers, for an application with dagger flag `dag`: the arithmetic of
// wilson_pack_dir (lower face: P psi -> send_bwd; upper face: U^+ P psi -> send_fwd), so a prepacked application sees the same bits
template <int NU>
__device__ __forceinline__ void wilson_pack_site(const HArgs& k, const cd (&v)[12], const int (&c)[4], int slot, int ps, int i, int dag) {
    const Geom& g = k.g;
    if (!g.part[NU]) return;
    const bool lo = c[NU] == 0, hi = c[NU] == g.L[NU] - 1;
    if (!lo && !hi) return;
    const int Fh = g.Vh / g.L[NU];
    const int f = coords_to_face(g, NU, c);
    if (lo) {
        cd h0[3], h1[3];
        if (NU == 3) { if (dag) project_regs<NU, -1>(h0, h1, v); else project_regs<NU, 1>(h0, h1, v + 6); }
        else { if (dag) project_regs<NU, -1>(h0, h1, v); else project_regs<NU, 1>(h0, h1, v); }
        real2* dst = k.send_bwd[NU] + (size_t)slot * 6 * Fh + f;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { st(dst + (size_t)cc * Fh, h0[cc]); st(dst + (size_t)(3 + cc) * Fh, h1[cc]); }
    }
    if (hi) {
        cd h0[3], h1[3], u[9], x0[3], x1[3];
        if (NU == 3) { if (dag) project_regs<NU, 1>(h0, h1, v + 6); else project_regs<NU, -1>(h0, h1, v); }
        else { if (dag) project_regs<NU, 1>(h0, h1, v); else project_regs<NU, -1>(h0, h1, v); }
        load_link(u, k.gauge + glink_off(g, ps, NU, i), glink_stride(g));
        su3_mv<true>(x0, u, h0);
        su3_mv<true>(x1, u, h1);
        real2* dst = k.send_fwd[NU] + (size_t)slot * 6 * Fh + f;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { st(dst + (size_t)cc * Fh, x0[cc]); st(dst + (size_t)(3 + cc) * Fh, x1[cc]); }
    }
}

// pack blocks appended to the CG's x/p update launch (solvers.hip): thread -> (direction, side, slot, face site) like wilson_pack_dir, the new
// search direction p' = r + beta p at that site is formed in registers with the update kernel's own fma (same bits as the value the flat
// part of the launch stores) and packed for the next D p.  k.in = p_k, k.upd = r (read only), pb = index of the pack block, npx = pack
// blocks per (direction, side).
template <int MU>
__device__ inline void wilson_pack_axpy_dir(const HArgs& k, int side, int t, double be) {
    const Geom& g = k.g;
    if (!g.part[MU]) return;
    const int Fh = g.Vh / g.L[MU];
    if (t >= 2 * Fh) return;
    const int slot = t / Fh, f = t - slot * Fh;       // slot = parity of the RECEIVING site
    const int ps = 1 - slot;
    int c[4];
    face_to_coords(g, MU, side ? g.L[MU] - 1 : 0, ps, f, c);
    const int i = coords_to_cb(g, c);
    const real2* __restrict__ pp = (ps ? k.in[1] : k.in[0]) + sp12_off(i);
    const real2* __restrict__ rp = (ps ? k.upd[1] : k.upd[0]) + sp12_off(i);
    cd v[12];
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const cd pv = ld(pp + co12(j)), rv = ld(rp + co12(j));
        v[j] = mk(fma(be, pv.re, rv.re), fma(be, pv.im, rv.im));
    }
    // only this face: the site's other faces are packed by the threads of those faces
    cd h0[3], h1[3];
    if (side == 0) {
        if (MU == 3) project_regs<MU, 1>(h0, h1, v + 6); else project_regs<MU, 1>(h0, h1, v);
        real2* dst = k.send_bwd[MU] + (size_t)slot * 6 * Fh + f;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { st(dst + (size_t)cc * Fh, h0[cc]); st(dst + (size_t)(3 + cc) * Fh, h1[cc]); }
    } else {
        project_regs<MU, -1>(h0, h1, v);
        cd u[9], x0[3], x1[3];
        load_link(u, k.gauge + glink_off(g, ps, MU, i), glink_stride(g));
        su3_mv<true>(x0, u, h0);
        su3_mv<true>(x1, u, h1);
        real2* dst = k.send_fwd[MU] + (size_t)slot * 6 * Fh + f;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { st(dst + (size_t)cc * Fh, x0[cc]); st(dst + (size_t)(3 + cc) * Fh, x1[cc]); }
    }
}
__device__ inline void wilson_pack_axpy_block(const HArgs& k, int pb, int npx, double be) {
    const int y = pb / npx, x = pb - y * npx;
    const int side = y & 1, t = x * (int)blockDim.x + (int)threadIdx.x;
    switch (y >> 1) {
    case 0: wilson_pack_axpy_dir<0>(k, side, t, be); break;
    case 1: wilson_pack_axpy_dir<1>(k, side, t, be); break;
    case 2: wilson_pack_axpy_dir<2>(k, side, t, be); break;
    default: wilson_pack_axpy_dir<3>(k, side, t, be); break;
    }
}

inline int persist_grid(lqcd_ctx_s* c, int nvirt) {
    int g = c->num_cu * (c->tun.persist_per_cu > 0 ? c->tun.persist_per_cu : 2);
    g -= g % 8;
    if (g < 8) g = 8;
    return std::min(g, nvirt);
}

// stencil_alt.hip (fp64 build only): launches the opt-in Wilson variant selected by dslash_variant if it applies to this call;
// returns false if the default direction-split kernel should run instead
bool launch_wilson_alt(lqcd_ctx_s* c, const StencilCall& s, const KArgs& k, size_t pad);

}  // inline namespace (precision)
}  // namespace lqcd
