// stencil_pair32.hip -- fp32 Wilson Dslash on SITE PAIRS for the inner solver of the mixed-precision CG (BASELINE configs[4]).
//
// Why.  The fp32 build of the direction-split kernel (stencil.hip, one site per lane) moves exactly its compulsory bytes (PMC: 0.79 GB per
// launch at 32^3 x 64 for 0.81 GB compulsory, L2 hit 0.69) but only at 4 TB/s: a workgroup lives as long as in the fp64 kernel (7.8 us vs
// 8.4 us) while it moves half the bytes, the SIMD issue ports are 82 % busy (profiles/r03_pmc_mixed_fp32.csv) -- a quarter of its VALU
// instructions are v_mov shuffles that line the (re, im) halves of a float2 up for v_pk_fma_f32.  The kernel is bound by instruction issue
// and by per-workgroup latencies, not by bandwidth.
//
// What.  A lane holds TWO sites, n and n + T/2 in the t direction: `real` is a 2-vector, every field element is 16 bytes
// (re_A, re_B, im_A, im_B), every arithmetic instruction is a packed fp32 operation over two independent sites with no shuffles, and the
// index arithmetic, the barrier, the LDS exchange and the dispatch of a workgroup are amortised over 128 sites.  Translation by T/2 commutes
// with every hop, so the neighbour of a pair is a pair; only the hop across t = T/2 - 1 -> T/2 (and 0 -> -1) finds its two neighbours in the
// pair of the other end of the half lattice with the slots swapped, and there the slot that wraps the full lattice takes the boundary
// sign.  Registers and LDS per workgroup are those of the fp64 kernel (3 workgroups per CU), so is the number of bytes each wave keeps
// in flight.  Layout of a pair field: the checkerboard layout of lqcd_internal.h on the half lattice L0 x L1 x L2 x T/2 with 16-byte
// elements -- spinor [parity][chunk][12][64], links (rows 0, 1) [parity][chunk][mu][6][64]; pair (p, i) holds sites (p, i) and
// (p, i + Vh/2) of the full lattice (T/2 even keeps the parity).
//
// Scope: Wilson r = 1 without clover term on an unpartitioned lattice whose z-planes are whole 64-site chunks and whose T is a multiple
// of 4, links unitary to 1e-14 (12-real rule of the fp64 path); anything else keeps the one-site-per-lane fp32 kernels.  Same scalar
// addressing as wilson_dirsplit_s (stencil.hip): t, z, y-chunk wave-uniform, scalar base + 32-bit lane offset + immediate.
#include "lqcd_internal.h"

#include <algorithm>
#include <type_traits>

namespace lqcd {
namespace pair32 {

typedef float v2f __attribute__((ext_vector_type(2)));
struct cx { v2f re, im; };
__device__ __forceinline__ cx mkx(v2f a, v2f b) { cx r; r.re = a; r.im = b; return r; }
__device__ __forceinline__ v2f splat(float a) { v2f r = {a, a}; return r; }
__device__ __forceinline__ v2f vfma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ cx ldx(const float4* p) { const float4 v = *p; cx r; r.re = v2f{v.x, v.y}; r.im = v2f{v.z, v.w}; return r; }
__device__ __forceinline__ cx ldx_nt(const float4* p) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    cx r; r.re = v2f{v.x, v.y}; r.im = v2f{v.z, v.w}; return r;
}
// 16-bit links (tunable mixed_links16): a link element of a pair is one 8-byte word of four fixed-point int16, (re_A, re_B) in .x and (im_A, im_B) in .y, value = n / 32767
// (|u| <= 1 for a unitary matrix: an absolute rounding error of 1.5e-5 per real, where fp16 would leave 2.4e-4 next to 1).  Two converts per word half and one packed multiply per two reals.
constexpr float L16_SCALE = 1.f / 32767.f;
template <bool NT>
__device__ __forceinline__ cx ldx16(const uint2* p) {
    uint2 v;
    if constexpr (NT) {
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u t = __builtin_nontemporal_load(reinterpret_cast<const v2u*>(p));
        v.x = t.x; v.y = t.y;
    } else v = *p;
    const v2f sc = {L16_SCALE, L16_SCALE};
    cx r;
    r.re = v2f{(float)(short)(v.x & 0xffffu), (float)((int)v.x >> 16)} * sc;
    r.im = v2f{(float)(short)(v.y & 0xffffu), (float)((int)v.y >> 16)} * sc;
    return r;
}
__device__ __forceinline__ void stx(float4* p, cx v) { *p = make_float4(v.re.x, v.re.y, v.im.x, v.im.y); }
__device__ __forceinline__ void stx_nt(float4* p, cx v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f t = {v.re.x, v.re.y, v.im.x, v.im.y};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}
__device__ __forceinline__ cx operator+(cx a, cx b) { return mkx(a.re + b.re, a.im + b.im); }
__device__ __forceinline__ cx swap_slots(cx a) { return mkx(a.re.yx, a.im.yx); }
template <int K> __device__ __forceinline__ cx mul_ipow(cx a) {      // multiply by i^K
    constexpr int k = ((K % 4) + 4) % 4;
    if constexpr (k == 0) return a;
    else if constexpr (k == 1) return mkx(-a.im, a.re);
    else if constexpr (k == 2) return mkx(-a.re, -a.im);
    else return mkx(a.im, -a.re);
}
__device__ __forceinline__ void cfma(cx& acc, cx a, cx b) {           // acc += a b
    acc.re = vfma(a.re, b.re, acc.re); acc.re = vfma(-a.im, b.im, acc.re);
    acc.im = vfma(a.re, b.im, acc.im); acc.im = vfma(a.im, b.re, acc.im);
}
__device__ __forceinline__ void cfma_conj(cx& acc, cx a, cx b) {      // acc += conj(a) b
    acc.re = vfma(a.re, b.re, acc.re); acc.re = vfma(a.im, b.im, acc.re);
    acc.im = vfma(a.re, b.im, acc.im); acc.im = vfma(-a.im, b.re, acc.im);
}
template <bool ADJ>
__device__ __forceinline__ void su3_mv(cx (&chi)[3], const cx (&u)[9], const cx (&h)[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        cx t = mkx(splat(0.f), splat(0.f));
#pragma unroll
        for (int b = 0; b < 3; b++) {
            if constexpr (ADJ) cfma_conj(t, u[b * 3 + a], h[b]);
            else cfma(t, u[a * 3 + b], h[b]);
        }
        chi[a] = t;
    }
}
// row 2 = conj(row 0 x row 1)
__device__ __forceinline__ void recon_row2(cx (&u)[9]) {
#pragma unroll
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const cx a = u[b1], bb = u[3 + b2], c = u[b2], d = u[3 + b1];
        v2f wr = a.re * bb.re;
        wr = vfma(-a.im, bb.im, wr); wr = vfma(-c.re, d.re, wr); wr = vfma(c.im, d.im, wr);
        v2f wi = a.re * bb.im;
        wi = vfma(a.im, bb.re, wi); wi = vfma(-c.re, d.im, wi); wi = vfma(-c.im, d.re, wi);
        u[6 + b] = mkx(wr, -wi);
    }
}
// h = rows 0, 1 of (1 - S gamma_mu) psi (mu = 3: the two rows the projector keeps, factor 2 included; sp then points at those six components)
template <int MU, int S>
__device__ __forceinline__ void project(cx (&h0)[3], cx (&h1)[3], const cx* sp) {
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
        for (int c = 0; c < 3; c++) { h0[c] = mkx(2.f * sp[c].re, 2.f * sp[c].im); h1[c] = mkx(2.f * sp[3 + c].re, 2.f * sp[3 + c].im); }
    }
}
template <int MU, int S>
__device__ __forceinline__ void reconstruct(cx (&acc)[12], const cx (&chi0)[3], const cx (&chi1)[3]) {
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

struct PairArgs {
    const float4* gauge;      // rows 0, 1 of every link, pair layout
    const uint2* gauge16;     // H16 instances: the same elements as four int16 (ldx16)
    float4* dst[2];           // out, or r in update mode (read and written)
    const float4* in[2];
    const float4* xin[2];
    float4* xacc[2];          // update mode: x += alpha p rides in the epilogue (nullptr: not asked for)
    const float4* pacc[2];
    double* norm_partial;
    const double* upd_scal;
    const double* skip;
    const float4* dotz[2];    // dot mode (StencilCall::dot_z): Re / Im <z, out> and |out|^2 per workgroup -> dot_partial[3 b ..]
    const float4* dotz2[2];   // StencilCall::dot_z2 (z = xin only): a second inner product <z2, out> -> five values per workgroup
    double* dot_partial;
    int dot_conj;
    float a, b;
    int nt_store;
    int both, pmode;
    int XH, L1, L2, LTh, nchp;     // LTh = T/2, nchp = chunks of pairs per parity
    FastDiv dXH;
    float sgn_f[4], sgn_b[4];      // sign of a hop that wraps the (full) lattice
    int cps, cpp, cpr, per_pass, ty, tz, ysplit;
    FastDiv d_perpass, d_cpr, d_ysplit, d_ty, d_cpp;
};

__device__ __forceinline__ int fdiv_nb(int n, const FastDiv& f) {
    const int q = (int)(__umulhi((unsigned)n, f.m) >> f.sh);
    return f.d == 1 ? n : q;
}
// virtual block -> parity, t (of the half lattice), z, chunk inside the z-plane: the XCD tile sweep of stencil.hip (map_block_v, remap 2)
__device__ __forceinline__ void pair_map(const PairArgs& a, int b, int& p, int& t, int& z, int& yc) {
    const int xcd = b & 7;
    int j = b >> 3;
    p = a.both ? (j & 1) : a.pmode;
    j = a.both ? (j >> 1) : j;
    const int pass = fdiv_nb(j, a.d_perpass);
    j -= pass * a.per_pass;
    t = fdiv_nb(j, a.d_cpr);
    const int m = j - t * a.cpr, sd = xcd + 8 * pass;
    const int sz = fdiv_nb(sd, a.d_ysplit), sy = sd - sz * a.ysplit;
    const int zz = fdiv_nb(m, a.d_ty), yy = m - zz * a.ty;
    const int s = a.ysplit > 1 ? (sz * a.tz + zz) * a.cpp + sy * a.ty + yy : sd * a.cpr + m;
    z = fdiv_nb(s, a.d_cpp);
    yc = s - z * a.cpp;
}

struct PairSite {
    unsigned own, nf, nb;      // byte offsets inside a parity block of a spinor field
    unsigned uf, ub;           // byte offsets inside a parity block of the link field
    v2f sf, sb;                // boundary signs per slot
    int p;
    bool swf, swb;             // t direction: the neighbour pair holds the two neighbours in swapped slots
};
template <int MU>
__device__ __forceinline__ PairSite pair_site(const PairArgs& a, int b, int lane) {
    constexpr unsigned SPC = 12 * 64 * 16, LKC = 4 * 6 * 64 * 16, LKM = 6 * 64 * 16;
    PairSite s;
    int t, z, yc;
    pair_map(a, b, s.p, t, z, yc);
    const int chunk = t * a.cps + z * a.cpp + yc;
    s.own = (unsigned)chunk * SPC + (unsigned)lane * 16u;
    s.uf = (unsigned)chunk * LKC + MU * LKM + (unsigned)lane * 16u;
    s.swf = false; s.swb = false;
    if constexpr (MU >= 2) {
        const int c = MU == 2 ? z : t, Lc = MU == 2 ? a.L2 : a.LTh, st = MU == 2 ? a.cpp : a.cps;
        const bool wf = c == Lc - 1, wb = c == 0;
        const int cf = wf ? chunk - (Lc - 1) * st : chunk + st;
        const int cb = wb ? chunk + (Lc - 1) * st : chunk - st;
        s.nf = (unsigned)cf * SPC + (unsigned)lane * 16u;
        s.nb = (unsigned)cb * SPC + (unsigned)lane * 16u;
        s.ub = (unsigned)cb * LKC + MU * LKM + (unsigned)lane * 16u;
        if constexpr (MU == 2) {
            s.sf = splat(wf ? a.sgn_f[2] : 1.f);
            s.sb = splat(wb ? a.sgn_b[2] : 1.f);
        } else {
            // t = T/2 - 1: slot A (site T/2 - 1) hops to site T/2 = slot B of the pair at t = 0 (no wrap), slot B (site T - 1) to site 0 = slot A
            // of that pair across the boundary; t = 0, backward: slot A (site 0) to site T - 1 = slot B of the pair at T/2 - 1 across the
            // boundary, slot B (site T/2) to site T/2 - 1 = its slot A
            s.swf = wf; s.swb = wb;
            s.sf = wf ? v2f{1.f, a.sgn_f[3]} : splat(1.f);
            s.sb = wb ? v2f{a.sgn_b[3], 1.f} : splat(1.f);
        }
    } else {
        const int cbp = yc * 64 + lane;
        const int y = fdiv(cbp, a.dXH), xh = cbp - y * a.XH;
        const int i = chunk * 64 + lane;
        int nf, nb;
        bool wf, wb;
        if constexpr (MU == 0) {
            const int q = (y + z + t + s.p) & 1;
            wf = q && xh == a.XH - 1; wb = !q && xh == 0;
            nf = q ? (wf ? i - (a.XH - 1) : i + 1) : i;
            nb = q ? i : (wb ? i + (a.XH - 1) : i - 1);
        } else {
            wf = y == a.L1 - 1; wb = y == 0;
            nf = wf ? i - (a.L1 - 1) * a.XH : i + a.XH;
            nb = wb ? i + (a.L1 - 1) * a.XH : i - a.XH;
        }
        s.nf = (unsigned)(nf >> 6) * SPC + (unsigned)(nf & 63) * 16u;
        s.nb = (unsigned)(nb >> 6) * SPC + (unsigned)(nb & 63) * 16u;
        s.ub = (unsigned)(nb >> 6) * LKC + MU * LKM + (unsigned)(nb & 63) * 16u;
        s.sf = splat(wf ? a.sgn_f[MU] : 1.f);
        s.sb = splat(wb ? a.sgn_b[MU] : 1.f);
    }
    return s;
}
template <typename T>
__device__ __forceinline__ const float4* boff(const T* base, unsigned bytes) { return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + bytes); }

__device__ __forceinline__ void apply_sign(cx (&h0)[3], cx (&h1)[3], v2f sign) {
    if (__builtin_amdgcn_ballot_w64(sign.x != 1.f || sign.y != 1.f) != 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { h0[c] = mkx(sign * h0[c].re, sign * h0[c].im); h1[c] = mkx(sign * h1[c].re, sign * h1[c].im); }
    }
}

template <int MU, bool DAG, bool NTB, bool DOT = false, bool H16 = false>
__device__ __forceinline__ void pair_wave(const PairArgs& a, float4 (*part)[12][64], int lane, float al_upd, v2f& nrm, v2f& dre, v2f& dim, v2f& dre2, v2f& dim2) {
    constexpr int SF = DAG ? -1 : 1;
    constexpr int NS = MU == 3 ? 6 : 12;
    constexpr int FF = MU == 3 ? (SF > 0 ? 6 : 0) : 0;
    constexpr int FB = MU == 3 ? (SF > 0 ? 0 : 6) : 0;
    const size_t gpar = (size_t)a.nchp * 4 * 6 * 64;           // float4 elements of one parity block of the link field
    const PairSite s = pair_site<MU>(a, blockIdx.x, lane);
    cx xv[3], rv[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) { xv[cc] = mkx(splat(0.f), splat(0.f)); rv[cc] = xv[cc]; }
    const bool z_is_x = DOT && (s.p ? a.dotz[1] == a.xin[1] : a.dotz[0] == a.xin[0]) && a.a != 0.f;
    const bool two = DOT && z_is_x && (s.p ? a.dotz2[1] : a.dotz2[0]) != nullptr;      // second inner product: z = xin leaves the registers of z to z2
    if constexpr (DOT) {        // dot mode: z takes the registers the old r has in update mode
        if (!z_is_x) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ldx(boff(s.p ? a.dotz[1] : a.dotz[0], s.own) + (3 * MU + cc) * 64);
        } else if (two) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ldx(boff(s.p ? a.dotz2[1] : a.dotz2[0], s.own) + (3 * MU + cc) * 64);
        }
    } else if (a.upd_scal) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ldx(boff(s.p ? a.dst[1] : a.dst[0], s.own) + (3 * MU + cc) * 64);
    }
    if (a.a != 0.f) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ldx(boff(s.p ? a.xin[1] : a.xin[0], s.own) + (3 * MU + cc) * 64);
    }
    cx acc[12], chi0[3], chi1[3], h0[3], h1[3];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = mkx(splat(0.f), splat(0.f));
    {
        cx sp[NS], u[9];
        const float4* ps = boff(s.p ? a.in[0] : a.in[1], s.nf);
        const float4* pu = boff(a.gauge + (s.p ? gpar : 0), s.uf);
        const uint2* pu16 = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.gauge16 + (s.p ? gpar : 0)) + (s.uf >> 1));
#pragma unroll
        for (int j = 0; j < NS; j++) sp[j] = ldx(ps + (FF + j) * 64);
#pragma unroll
        for (int j = 0; j < 6; j++) u[j] = H16 ? ldx16<false>(pu16 + j * 64) : ldx(pu + j * 64);
        if (MU == 3 && s.swf) {
#pragma unroll
            for (int j = 0; j < NS; j++) sp[j] = swap_slots(sp[j]);
        }
        recon_row2(u);
        project<MU, SF>(h0, h1, sp);
        apply_sign(h0, h1, s.sf);
        su3_mv<false>(chi0, u, h0);
        su3_mv<false>(chi1, u, h1);
        reconstruct<MU, SF>(acc, chi0, chi1);
    }
    __builtin_amdgcn_sched_barrier(0);      // the backward operands take the registers of the forward ones (3 waves per SIMD)
    {
        cx sp[NS], u[9];
        const float4* ps = boff(s.p ? a.in[0] : a.in[1], s.nb);
        const float4* pu = boff(a.gauge + (s.p ? 0 : gpar), s.ub);
        const uint2* pu16 = reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.gauge16 + (s.p ? 0 : gpar)) + (s.ub >> 1));
#pragma unroll
        for (int j = 0; j < NS; j++) sp[j] = ldx(ps + (FB + j) * 64);
#pragma unroll
        for (int j = 0; j < 6; j++) u[j] = H16 ? ldx16<NTB>(pu16 + j * 64) : (NTB ? ldx_nt(pu + j * 64) : ldx(pu + j * 64));
        if (MU == 3 && s.swb) {
#pragma unroll
            for (int j = 0; j < NS; j++) sp[j] = swap_slots(sp[j]);
#pragma unroll
            for (int j = 0; j < 6; j++) u[j] = swap_slots(u[j]);
        }
        recon_row2(u);
        project<MU, -SF>(h0, h1, sp);
        apply_sign(h0, h1, s.sb);
        su3_mv<true>(chi0, u, h0);
        su3_mv<true>(chi1, u, h1);
        reconstruct<MU, -SF>(acc, chi0, chi1);
    }
#pragma unroll
    for (int j = 0; j < 12; j++) part[MU][j][lane] = make_float4(acc[j].re.x, acc[j].re.y, acc[j].im.x, acc[j].im.y);
    __syncthreads();
    float4* dstp = const_cast<float4*>(boff(s.p ? a.dst[1] : a.dst[0], s.own));
    const v2f av = splat(a.a), bv = splat(a.b), mal = splat(-al_upd);
    // update mode with an x accumulator: x += alpha p for this wave's three components.  Its six loads are issued here, behind the barrier --
    // the kernel sits at its register limit while the hops are in flight -- and land while the LDS partials are summed.
    const bool xupd = a.upd_scal && a.xacc[0];
    cx xs[3], ps[3];
    float4* xsp = nullptr;
    if (xupd) {
        xsp = const_cast<float4*>(boff(s.p ? a.xacc[1] : a.xacc[0], s.own));
        const float4* pp = boff(s.p ? a.pacc[1] : a.pacc[0], s.own);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) { xs[cc] = ldx(xsp + (3 * MU + cc) * 64); ps[cc] = ldx(pp + (3 * MU + cc) * 64); }
    }
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
        const int j = 3 * MU + cc;
        const float4 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
        const v2f sre = (v2f{s0.x, s0.y} + v2f{s1.x, s1.y}) + (v2f{s2.x, s2.y} + v2f{s3.x, s3.y});
        const v2f sim = (v2f{s0.z, s0.w} + v2f{s1.z, s1.w}) + (v2f{s2.z, s2.w} + v2f{s3.z, s3.w});
        cx v = mkx(vfma(av, xv[cc].re, bv * sre), vfma(av, xv[cc].im, bv * sim));
        if constexpr (DOT) {        // <z, v> = conj(z) v per slot, next to |v|^2
            const cx z = z_is_x ? xv[cc] : rv[cc];
            nrm = vfma(v.re, v.re, nrm); nrm = vfma(v.im, v.im, nrm);
            if (a.nt_store) stx_nt(dstp + j * 64, v); else stx(dstp + j * 64, v);
            dre = vfma(z.re, v.re, dre); dre = vfma(z.im, v.im, dre);
            dim = vfma(z.re, v.im, dim); dim = vfma(-z.im, v.re, dim);
            if (two) {
                const cx z2 = rv[cc];
                dre2 = vfma(z2.re, v.re, dre2); dre2 = vfma(z2.im, v.im, dre2);
                dim2 = vfma(z2.re, v.im, dim2); dim2 = vfma(-z2.im, v.re, dim2);
            }
        } else if (a.upd_scal) {
            cx r = rv[cc];
            r.re = vfma(mal, v.re, r.re); r.im = vfma(mal, v.im, r.im);
            nrm = vfma(r.re, r.re, nrm); nrm = vfma(r.im, r.im, nrm);
            stx(dstp + j * 64, r);
            if (xupd) {
                const v2f al = splat(al_upd);
                cx xn = mkx(vfma(al, ps[cc].re, xs[cc].re), vfma(al, ps[cc].im, xs[cc].im));
                stx(xsp + j * 64, xn);
            }
        } else {
            nrm = vfma(v.re, v.re, nrm); nrm = vfma(v.im, v.im, nrm);
            if (a.nt_store) stx_nt(dstp + j * 64, v); else stx(dstp + j * 64, v);
        }
    }
}

template <bool DAG, bool NTB, bool DOT = false, bool H16 = false>
__global__ __launch_bounds__(256, 3) void wilson_dirsplit_pair32(PairArgs a) {
    __shared__ float4 part[4][12][64];  // 48 KiB
    __shared__ double red[DOT ? 20 : 4];
    if ((a.upd_scal && a.upd_scal[S_DONE] != 0.0) || (a.skip && a.skip[S_DONE] != 0.0)) return;
    const float al_upd = a.upd_scal ? (float)a.upd_scal[S_ALPHA] : 0.f;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    v2f nrm = splat(0.f), dre = splat(0.f), dim = splat(0.f), dre2 = splat(0.f), dim2 = splat(0.f);
    switch (w) {
    case 0: pair_wave<0, DAG, NTB, DOT, H16>(a, part, lane, al_upd, nrm, dre, dim, dre2, dim2); break;
    case 1: pair_wave<1, DAG, NTB, DOT, H16>(a, part, lane, al_upd, nrm, dre, dim, dre2, dim2); break;
    case 2: pair_wave<2, DAG, NTB, DOT, H16>(a, part, lane, al_upd, nrm, dre, dim, dre2, dim2); break;
    default: pair_wave<3, DAG, NTB, DOT, H16>(a, part, lane, al_upd, nrm, dre, dim, dre2, dim2); break;
    }
    if constexpr (DOT) {                // three sums per workgroup (both slots of a lane), the order of the fp64 kernels' dot epilogue
        const double di = (double)dim.x + (double)dim.y;
        const bool five = a.dotz2[0] != nullptr || a.dotz2[1] != nullptr;
        double t3[5] = {(double)dre.x + (double)dre.y, (a.dot_conj & 1) ? -di : di, (double)nrm.x + (double)nrm.y, (double)dre2.x + (double)dre2.y, (double)dim2.x + (double)dim2.y};
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (q < 3 || five) {
                t3[q] = wave_sum(t3[q]);
                if (lane == 0) red[4 * q + w] = t3[q];
            }
        }
        __syncthreads();
        const int nv = five ? 5 : 3;
        if ((int)threadIdx.x < nv) a.dot_partial[(a.dot_conj & 2) ? (size_t)threadIdx.x * gridDim.x + blockIdx.x : nv * (size_t)blockIdx.x + threadIdx.x] =
            (red[4 * threadIdx.x] + red[4 * threadIdx.x + 1]) + (red[4 * threadIdx.x + 2] + red[4 * threadIdx.x + 3]);
        return;
    }
    if (a.norm_partial) {
        double s = (double)nrm.x + (double)nrm.y;
        s = wave_sum(s);
        if (lane == 0) red[w] = s;
        __syncthreads();
        if (threadIdx.x == 0) a.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// ------------------------------------------------------------------------------------------ layout conversions
// pair (p, i) of the half lattice <-> sites (p, i) and (p, i + Vh/2) of the full lattice; blk64 = fp64 elements of one parity block
// (a parity block of either field is blk64 elements of its own type apart: the padding chunk of the fp64 layout stays unused in the pair field)
// dst2 / dst3 (may be null): two more copies of the converted field, xzero (may be null): a field of the same shape set to zero -- the start of an fp32 BiCGStab chain
// (r, r0 = r, p = r, x = 0) in one pass over the fp64 residual
__global__ __launch_bounds__(256) void cvt_wilson_to_pair32(float4* __restrict__ dst, const double2* __restrict__ src, int Vh, int nchp, size_t blk64, double scale, int npar,
                                                            float4* __restrict__ dst2 = nullptr, float4* __restrict__ dst3 = nullptr, float4* __restrict__ xzero = nullptr) {
    const size_t n = (size_t)npar * nchp * 768;       // npar = 1: ONE parity block (the even-odd solver's half-lattice vectors; dst / src point at it)
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), j = (int)((t >> 6) % 12), chp = (int)((t / 768) % nchp), p = (int)(t / ((size_t)768 * nchp));
        const int iA = chp * 64 + lane, iB = iA + Vh / 2;
        const double2 a = src[p * blk64 + (size_t)(iA >> 6) * 768 + j * 64 + (iA & 63)], b = src[p * blk64 + (size_t)(iB >> 6) * 768 + j * 64 + (iB & 63)];
        const size_t o = p * (blk64 / 2) + ((size_t)chp * 12 + j) * 64 + lane;
        const float4 v = make_float4((float)(a.x * scale), (float)(b.x * scale), (float)(a.y * scale), (float)(b.y * scale));
        dst[o] = v;
        if (dst2) dst2[o] = v;
        if (dst3) dst3[o] = v;
        if (xzero) xzero[o] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
// the fp64 residual rhs - q of ONE parity block, scaled, straight into the pair layout (+ the copies / zero field of cvt_wilson_to_pair32) with |rhs - q|^2 block partials:
// the true residual behind a correction step of the mixed-precision even-odd BiCGStab and the start of the next fp32 chain in one pass (mixed.hip)
__global__ __launch_bounds__(256) void residual_to_pair32(float4* __restrict__ dst, const double2* __restrict__ rhs, const double2* __restrict__ q, int Vh, int nchp, double scale,
                                                          float4* __restrict__ dst2, float4* __restrict__ dst3, float4* __restrict__ xzero, double* __restrict__ partial) {
    __shared__ double red[4];
    const size_t n = (size_t)nchp * 768;
    double acc = 0.0;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), j = (int)((t >> 6) % 12), chp = (int)(t / 768);
        const int iA = chp * 64 + lane, iB = iA + Vh / 2;
        const size_t oA = (size_t)(iA >> 6) * 768 + j * 64 + (iA & 63), oB = (size_t)(iB >> 6) * 768 + j * 64 + (iB & 63);
        const double2 ra = rhs[oA], qa = q[oA], rb = rhs[oB], qb = q[oB];
        const double ax = ra.x - qa.x, ay = ra.y - qa.y, bx = rb.x - qb.x, by = rb.y - qb.y;
        acc = fma(ax, ax, acc); acc = fma(ay, ay, acc); acc = fma(bx, bx, acc); acc = fma(by, by, acc);
        const float4 v = make_float4((float)(ax * scale), (float)(bx * scale), (float)(ay * scale), (float)(by * scale));
        dst[t] = v;
        if (dst2) dst2[t] = v;
        if (dst3) dst3[t] = v;
        if (xzero) xzero[t] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// y (fp64, full lattice) += a * x (pairs)
__global__ __launch_bounds__(256) void axpy_from_pair32(double2* __restrict__ y, const float4* __restrict__ x, int Vh, int nchp, size_t blk64, double a, int npar) {
    const size_t n = (size_t)npar * nchp * 768;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), j = (int)((t >> 6) % 12), chp = (int)((t / 768) % nchp), p = (int)(t / ((size_t)768 * nchp));
        const int iA = chp * 64 + lane, iB = iA + Vh / 2;
        const float4 v = x[p * (blk64 / 2) + ((size_t)chp * 12 + j) * 64 + lane];
        double2* ya = y + p * blk64 + (size_t)(iA >> 6) * 768 + j * 64 + (iA & 63);
        double2* yb = y + p * blk64 + (size_t)(iB >> 6) * 768 + j * 64 + (iB & 63);
        double2 u = *ya, w = *yb;
        u.x = fma(a, (double)v.x, u.x); u.y = fma(a, (double)v.z, u.y);
        w.x = fma(a, (double)v.y, w.x); w.y = fma(a, (double)v.w, w.y);
        *ya = u; *yb = w;
    }
}
// links: from the fp64 12-real copy [parity][chunk][mu][6][64]
__global__ __launch_bounds__(256) void cvt_gauge12_pair32(float4* __restrict__ dst, const double2* __restrict__ src12, int Vh, int nch, int nchp) {
    const size_t n = (size_t)2 * nchp * 4 * 6 * 64;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), e = (int)((t >> 6) % 6), mu = (int)((t / 384) % 4), chp = (int)((t / 1536) % nchp), p = (int)(t / ((size_t)1536 * nchp));
        const int iA = chp * 64 + lane, iB = iA + Vh / 2;
        const double2 a = src12[((((size_t)p * nch + (iA >> 6)) * 4 + mu) * 6 + e) * 64 + (iA & 63)];
        const double2 b = src12[((((size_t)p * nch + (iB >> 6)) * 4 + mu) * 6 + e) * 64 + (iB & 63)];
        dst[t] = make_float4((float)a.x, (float)b.x, (float)a.y, (float)b.y);
    }
}

// links as int16: the same element order, 8 bytes per element
__global__ __launch_bounds__(256) void cvt_gauge12_pair16(uint2* __restrict__ dst, const double2* __restrict__ src12, int Vh, int nch, int nchp) {
    const size_t n = (size_t)2 * nchp * 4 * 6 * 64;
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < n; t += (size_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), e = (int)((t >> 6) % 6), mu = (int)((t / 384) % 4), chp = (int)((t / 1536) % nchp), p = (int)(t / ((size_t)1536 * nchp));
        const int iA = chp * 64 + lane, iB = iA + Vh / 2;
        const double2 a = src12[((((size_t)p * nch + (iA >> 6)) * 4 + mu) * 6 + e) * 64 + (iA & 63)];
        const double2 b = src12[((((size_t)p * nch + (iB >> 6)) * 4 + mu) * 6 + e) * 64 + (iB & 63)];
        auto q = [](double v) { return (unsigned)(__double2int_rn(fmin(fmax(v, -1.0), 1.0) * 32767.0) & 0xffff); };
        dst[t] = make_uint2(q(a.x) | (q(b.x) << 16), q(a.y) | (q(b.y) << 16));
    }
}

}  // namespace pair32

// ------------------------------------------------------------------------------------------ host side
bool pair32_geometry_ok(lqcd_ctx_s* c) {
    const Geom& g = c->geom;
    if (any_partitioned(c) || c->tun.xcd_remap != 2) return false;
    const int plane = g.XH * g.L[1];
    if (plane % 64 != 0 || (plane * g.L[2] / 64) % 8 != 0 || g.L[3] % 4 != 0) return false;
    if ((size_t)g.nch * 64 * 768 >= ((size_t)1 << 32)) return false;       // 32-bit byte offsets inside a parity block of the largest field
    return g.nch % 2 == 0 && g.Vs >= g.nch * 64 && g.Vs % 2 == 0;     // the pair field of a parity fits the parity block of the plain fp32 field
}
int pair32_num_blocks(lqcd_ctx_s* c) { return c->geom.nch; }      // 2 parities x nch / 2 chunks of pairs

int pair32_cvt_spinor(lqcd_ctx_s* c, float2* dst, const double2* src, double scale, int npar, float2* dst2, float2* dst3, float2* xzero) {
    const Geom& g = c->geom;
    hipLaunchKernelGGL(pair32::cvt_wilson_to_pair32, dim3(stream_grid(c, (size_t)g.nch * 768)), dim3(256), 0, c->stream, (float4*)dst, src, g.Vh, g.nch / 2,
                       (size_t)12 * g.Vs, scale, npar, (float4*)dst2, (float4*)dst3, (float4*)xzero);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
// one parity block; returns the number of block partials in *nb (c->d_partial)
int pair32_residual(lqcd_ctx_s* c, float2* dst, const double2* rhs, const double2* q, double scale, float2* dst2, float2* dst3, float2* xzero, int* nb) {
    const Geom& g = c->geom;
    *nb = stream_grid(c, (size_t)g.nch / 2 * 768);
    hipLaunchKernelGGL(pair32::residual_to_pair32, dim3(*nb), dim3(256), 0, c->stream, (float4*)dst, rhs, q, g.Vh, g.nch / 2, scale, (float4*)dst2, (float4*)dst3, (float4*)xzero,
                       c->d_partial);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
int pair32_axpy_to_f64(lqcd_ctx_s* c, double2* y, const float2* x, double a, int npar) {
    const Geom& g = c->geom;
    hipLaunchKernelGGL(pair32::axpy_from_pair32, dim3(stream_grid(c, (size_t)g.nch * 768)), dim3(256), 0, c->stream, y, (const float4*)x, g.Vh, g.nch / 2,
                       (size_t)12 * g.Vs, a, npar);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
int pair32_cvt_gauge12(lqcd_ctx_s* c, float2* dst, const double2* src12) {
    const Geom& g = c->geom;
    hipLaunchKernelGGL(pair32::cvt_gauge12_pair32, dim3(stream_grid(c, (size_t)g.nch * 1536)), dim3(256), 0, c->stream, (float4*)dst, src12, g.Vh, g.nch, g.nch / 2);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int pair32_cvt_gauge16(lqcd_ctx_s* c, void* dst, const double2* src12) {
    const Geom& g = c->geom;
    hipLaunchKernelGGL(pair32::cvt_gauge12_pair16, dim3(stream_grid(c, (size_t)g.nch * 1536)), dim3(256), 0, c->stream, (uint2*)dst, src12, g.Vh, g.nch, g.nch / 2);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// the launcher behind stencil_apply for StencilCall::prec == 2 (fields in pair layout; full-lattice applications only)
int launch_pair32_interior(lqcd_ctx_s* c, const StencilCall& s) {
    using namespace pair32;
    ARGCHK(s.kind == LQCD_WILSON && s.r == 1.0 && s.gauge12 && !s.clover && !s.alpha_partials && pair32_geometry_ok(c),
           "pair32 stencil: Wilson r = 1 with 12-real links on an unpartitioned lattice only");
    ARGCHK(!(s.dot_partial && s.upd_scal), "pair32 stencil: dot mode and update mode exclude each other");
    const Geom& g = c->geom;
    PairArgs a;
    a.gauge = (const float4*)s.gauge12;
    a.gauge16 = (const uint2*)s.gauge16;
    const bool upd = s.upd_scal != nullptr;
    for (int p = 0; p < 2; p++) { a.dst[p] = (float4*)(upd ? s.upd[p] : s.out[p]); a.in[p] = (const float4*)s.in[p]; a.xin[p] = (const float4*)s.xin[p]; }
    for (int p = 0; p < 2; p++) { a.xacc[p] = upd ? (float4*)s.xacc[p] : nullptr; a.pacc[p] = upd ? (const float4*)s.pacc[p] : nullptr; }
    a.norm_partial = s.norm_partial; a.upd_scal = s.upd_scal; a.skip = s.skip_flag;
    for (int p = 0; p < 2; p++) { a.dotz[p] = (const float4*)s.dot_z[p]; a.dotz2[p] = (const float4*)s.dot_z2[p]; }
    a.dot_partial = s.dot_partial; a.dot_conj = s.dot_conj;
    a.a = (float)s.a; a.b = (float)s.b;
    a.nt_store = c->tun.nt_store != 0;
    a.both = s.parity_mode == 2 ? 1 : 0; a.pmode = s.parity_mode == 2 ? 0 : s.parity_mode;      // parity hops: the pairs of ONE parity (even-odd solvers)
    a.XH = g.XH; a.L1 = g.L[1]; a.L2 = g.L[2]; a.LTh = g.L[3] / 2; a.nchp = g.nch / 2; a.dXH = g.dXH;
    for (int mu = 0; mu < 4; mu++) { a.sgn_f[mu] = (float)g.bc_fwd[mu]; a.sgn_b[mu] = (float)g.bc_bwd[mu]; }
    // the XCD tile sweep of stencil.hip make_kargs on the half lattice
    const int slice = g.XH * g.L[1] * g.L[2], plane = g.XH * g.L[1];
    int nsub = (c->tun.xcd_nsub >= 8 && c->tun.xcd_nsub % 8 == 0) ? c->tun.xcd_nsub : 8;
    while (nsub > 8 && (slice / 64) % nsub != 0) nsub -= 8;
    a.cps = slice / 64; a.cpp = plane / 64; a.ysplit = 1;
    for (int ys = c->tun.xcd_ysplit; ys > 1; ys--)
        if (a.cpp % ys == 0 && nsub % ys == 0 && g.L[2] % (nsub / ys) == 0) { a.ysplit = ys; break; }
    a.cpr = a.cps / nsub;
    a.ty = a.ysplit > 1 ? a.cpp / a.ysplit : 1;
    a.tz = a.cpr / a.ty;
    a.per_pass = std::max(1, a.cpr * a.LTh);
    a.d_perpass = make_fastdiv(a.per_pass); a.d_cpr = make_fastdiv(std::max(1, a.cpr)); a.d_ysplit = make_fastdiv(std::max(1, a.ysplit));
    a.d_ty = make_fastdiv(std::max(1, a.ty)); a.d_cpp = make_fastdiv(std::max(1, a.cpp));
    const dim3 grid(s.parity_mode == 2 ? pair32_num_blocks(c) : pair32_num_blocks(c) / 2), block(256);
    const bool ntb = (c->tun.nt_gauge & 1) != 0;
    auto go = [&](auto dag, auto nt, auto dot, auto h16) {
        hipLaunchKernelGGL((wilson_dirsplit_pair32<decltype(dag)::value, decltype(nt)::value, decltype(dot)::value, decltype(h16)::value>), grid, block, 0, c->stream, a);
    };
    auto pick = [&](auto dag, auto nt, auto dot) { if (a.gauge16) go(dag, nt, dot, std::true_type()); else go(dag, nt, dot, std::false_type()); };
    auto pick2 = [&](auto dag, auto nt) { if (s.dot_partial) pick(dag, nt, std::true_type()); else pick(dag, nt, std::false_type()); };
    auto pick3 = [&](auto dag) { if (ntb) pick2(dag, std::true_type()); else pick2(dag, std::false_type()); };
    if (s.dagger) pick3(std::true_type()); else pick3(std::false_type());
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

}  // namespace lqcd
