// stencil.hip -- Wilson and staggered Dslash for gfx950 (CDNA4): the default kernels (both precisions), the halo kernels and the launchers.
// Shared device helpers: stencil_common.h.  Opt-in Wilson variants 2-8 (measured alternatives, fp64 only): experiments/stencil_alt/stencil_alt.hip (LQCD_VARIANTS builds).
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
#include "stencil_common.h"

#include <cstring>
namespace lqcd {
inline namespace LQCD_PNS {

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
// (defined with the halo kernels below) out(n) += the hops of direction NU that cross a rank boundary, taken from the ghost buffers
template <int NU, bool DAG>
__device__ __forceinline__ void wilson_ext_add(cd (&acc)[12], const HArgs& k, const int (&c)[4], int slot, int pout, int i);
template <int NU>
__device__ __forceinline__ void staggered_ext_add(cd (&acc)[3], const HArgs& k, const int (&c)[4], int slot, int pout, int i);

// FOLD (folded one-stream halo schedule, apply.hip): the exchange is complete when the kernel starts, so the hops that leave the rank -- skipped by sign 0 above --
// are added from the ghost buffers by the direction wave itself (wilson_ext_add: the arithmetic of the exterior kernel).  Any partitioned direction, any instance of
// this kernel (clover epilogue, inverse clover blocks on the hop sum, fp32 build); the scalar-addressing kernel has its own FOLD instances (sdir_wave).
template <int MU, bool DAG, bool R12, bool FOLD = false>
__device__ inline void dirsplit_hops(cd (&acc)[12], const KArgs& k, int p, int i, const HArgs* h = nullptr) {
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
    if constexpr (FOLD) wilson_ext_add<MU, DAG>(acc, *h, c, k.parity_mode == 2 ? p : 0, p, i);
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
#define LQCD_DS_BOUNDS __launch_bounds__(256, (CLOV || CINV) ? 1 : LQCD_DS_OCC)
#else
#define LQCD_DS_BOUNDS __launch_bounds__(256)
#endif
// CINV (even-odd Wilson-clover solver): the packed matrix k.clover (the INVERSE clover blocks of the output parity) is applied to the HOP SUM,
// out = a xin + b C (H in), where CLOV applies it to the diagonal term -- the Schur operator 1 - k^2 A_ee^-1 H_eo A_oo^-1 H_oe becomes two launches
// with no intermediate field (every wave rebuilds the 12 summed components from the four direction partials: 48 LDS reads instead of 12).
// folded twins, bulk / boundary launches (KArgs::fsel): does this workgroup's chunk sit out this launch?  Every wave of the workgroup holds the same 64 sites, so
// the ballot is the same in all of them (no LDS, no barrier).  A chunk with a site on a partitioned face is a boundary chunk.
__device__ __forceinline__ bool fold_chunk_skipped(const KArgs& k, int chunk, int p) {
    if (!k.fsel) return false;
    const int i = chunk * 64 + (int)(threadIdx.x & 63);
    bool b = false;
    if (i < k.g.Vh) {
        int c[4];
        cb_to_coords(k.g, p, i, c);
#pragma unroll
        for (int mu = 0; mu < 4; mu++) b = b || (k.g.part[mu] && (c[mu] == 0 || c[mu] == k.g.L[mu] - 1));
    }
    const bool bnd = __any(b);
    return (k.fsel == 1) == bnd;
}

template <bool DAG, bool R12, bool CLOV, bool DOT, bool CINV, bool FOLD>
__device__ __forceinline__ void wilson_dirsplit_body(const KArgs& k, const HArgs* hf) {
    __shared__ real2 part[4][12][64];  // 48 KiB
    __shared__ double red[DOT ? 20 : 4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    if constexpr (FOLD) { if (fold_chunk_skipped(k, chunk, p)) return; }
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
#ifndef LQCD_UPD_PREFETCH      // fp32 build: fits under its 96-VGPR cap (LQCD_DS_OCC) since the kernel lost a tenth of its instructions (88 VGPRs, no spill)
#ifdef LQCD_F32
#ifndef LQCD_UPD_PREFETCH32
#define LQCD_UPD_PREFETCH32 1
#endif
#define LQCD_UPD_PREFETCH LQCD_UPD_PREFETCH32
#else
#define LQCD_UPD_PREFETCH 1
#endif
#endif
    // 1: before the hops (fp64: +12 VGPRs, still 3 waves/SIMD); 2: after the hops, in front of the LDS exchange (no long live range: the
    // fp32 build under its 96-VGPR cap), the barrier and the LDS round trip then cover the load
    if (LQCD_UPD_PREFETCH == 1 && valid && k.upd_scal) {      // CG update mode: the old r is read now, not after the barrier
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
        case 0: dirsplit_hops<0, DAG, R12, FOLD>(acc, k, p, i, hf); break;
        case 1: dirsplit_hops<1, DAG, R12, FOLD>(acc, k, p, i, hf); break;
        case 2: dirsplit_hops<2, DAG, R12, FOLD>(acc, k, p, i, hf); break;
        default: dirsplit_hops<3, DAG, R12, FOLD>(acc, k, p, i, hf); break;
        }
    }
    if (LQCD_UPD_PREFETCH == 2 && valid && k.upd_scal) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ld(k.upd[p] + sp12_off(i) + co12(3 * w + cc));
    }
    cd zv[DOT ? 3 : 1];
    if constexpr (DOT) {            // dot mode: this wave's three components of z, requested behind the hops (no long live range: the kernel stays at
#pragma unroll                      // 3 waves/SIMD); the LDS exchange and the barrier cover the load
        for (int cc = 0; cc < 3; cc++) zv[cc] = !valid ? mk(0, 0) : (k.dotz[p] == k.xin[p] && k.a != 0.0) ? xv[cc] : ld(k.dotz[p] + sp12_off(i) + co12(3 * w + cc));
    }                               // (z = the diagonal term's field -- <t, s> with t = M s -- is in registers already)
    // second inner product (StencilCall::dot_z2, only beside z = xin): <z2, out> -> five values per workgroup (merged BiCGStab chain: <r0, t> beside <t, s>, |t|^2)
    const bool two = DOT && k.dotz2[p] != nullptr && k.dotz[p] == k.xin[p] && k.a != 0.0;
    cd z2[DOT ? 3 : 1];
    if constexpr (DOT) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) z2[cc] = (valid && two) ? ld(k.dotz2[p] + sp12_off(i) + co12(3 * w + cc)) : mk(0, 0);
    }
#pragma unroll
    for (int j = 0; j < 12; j++) part[w][j][lane] = mk2(acc[j].re, acc[j].im);
    if constexpr (CLOV) if (valid && k.a != 0.0) {   // this wave's rows of A xin (its partial sums are already on their way to LDS)
        const real2* __restrict__ ca = k.clover + (((size_t)p * k.g.nch + (size_t)(i >> 6)) * 36) * 64 + (i & 63);
        if (w & 1) clover_rows<1>(xv, ca, cpsi, w >= 2);
        else clover_rows<0>(xv, ca, cpsi, w >= 2);
    }
    __syncthreads();
    real nrm = 0.0, dre = 0.0, dim = 0.0, dre2 = 0.0, dim2 = 0.0;
    cd hs[CINV ? 3 : 1];
    if constexpr (CINV) if (valid) {        // this wave's rows of C (H in): all 12 summed components, then the packed 6x6 blocks
        cd v12[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
            v12[j] = mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
        }
        const real2* __restrict__ ca = k.clover + (((size_t)p * k.g.nch + (size_t)(i >> 6)) * 36) * 64 + (i & 63);
        if (w & 1) clover_rows<1>(hs, ca, v12, w >= 2);
        else clover_rows<0>(hs, ca, v12, w >= 2);
    }
    if (valid) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            const int j = 3 * w + cc;
            const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
            cd s = CINV ? hs[cc] : mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
            cd v = k.b * s;
            v = mk(fma(k.a, xv[cc].re, v.re), fma(k.a, xv[cc].im, v.im));
            if (LQCD_UPD_PREFETCH) emit_pre(k, p, co12(j) + sp12_off(i), v, nrm, al_upd, rv[cc]);
            else emit(k, p, co12(j) + sp12_off(i), v, nrm, al_upd);
            if constexpr (DOT) {        // <z, v> = conj(z) v
                dre = fma(zv[cc].re, v.re, dre); dre = fma(zv[cc].im, v.im, dre);
                dim = fma(zv[cc].re, v.im, dim); dim = fma(-zv[cc].im, v.re, dim);
                dre2 = fma(z2[cc].re, v.re, dre2); dre2 = fma(z2[cc].im, v.im, dre2);
                dim2 = fma(z2[cc].re, v.im, dim2); dim2 = fma(-z2[cc].im, v.re, dim2);
            }
        }
    }
    if constexpr (DOT) {                // three sums per workgroup: a wave tree each, then the four waves in a fixed order; five with a second inner product
        const bool five = k.dotz2[0] != nullptr || k.dotz2[1] != nullptr;
        double t3[5] = {(double)dre, (double)((k.dot_conj & 1) ? -dim : dim), (double)nrm, (double)dre2, (double)dim2};
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (q < 3 || five) {
                t3[q] = wave_sum(t3[q]);
                if (lane == 0) red[4 * q + w] = t3[q];
            }
        }
        __syncthreads();
        const int nv = five ? 5 : 3;
        if ((int)threadIdx.x < nv) k.dot_partial[(k.dot_conj & 2) ? (size_t)threadIdx.x * gridDim.x + blockIdx.x : nv * (size_t)blockIdx.x + threadIdx.x] =
            (red[4 * threadIdx.x] + red[4 * threadIdx.x + 1]) + (red[4 * threadIdx.x + 2] + red[4 * threadIdx.x + 3]);      // (bit 1 of dot_conj: [value][workgroup] layout)
        return;
    }
    if (k.norm_partial) {
        nrm = wave_sum(nrm);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[vblock_of(k)] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <bool DAG, bool R12 = false, bool CLOV = false, bool DOT = false, bool CINV = false>
__global__ LQCD_DS_BOUNDS void wilson_dirsplit(KArgs k) { wilson_dirsplit_body<DAG, R12, CLOV, DOT, CINV, false>(k, nullptr); }
// the same launch with the boundary hops folded in (no exterior kernel): the HArgs carry the ghost buffers and the boundary signs of this rank
#ifdef LQCD_F32
#define LQCD_DS_BOUNDS_FOLD LQCD_DS_BOUNDS
#else
#define LQCD_DS_BOUNDS_FOLD __launch_bounds__(256, CLOV ? 2 : 3)      // (the ghost path adds ~4 VGPRs: without the cap two 18-real instances land on 169 and lose a wave per SIMD)
#endif
template <bool DAG, bool R12 = false, bool CLOV = false, bool CINV = false>
__global__ LQCD_DS_BOUNDS_FOLD void wilson_dirsplit_fold(KArgs k, HArgs h) { wilson_dirsplit_body<DAG, R12, CLOV, false, CINV, true>(k, &h); }


// ------------------------------------------------------------------------------------------ Wilson, direction-split, persistent + software-pipelined
// The persistent form of variant 1 (tunable dslash_pipe).  Variant 1's workgroup lives ~4 us and spends it in one dependent chain: index arithmetic -> forward-hop loads ->
// arithmetic -> backward-hop loads -> arithmetic -> LDS -> barrier -> LDS -> store; with every stream L2-hot the kernel still takes 0.27 ms of
// its 0.36 (profiles/r02_hot_ablations_and_occupancy.log) -- a CU-side floor made of phases that do not overlap at 3 waves/SIMD.
// Here a workgroup is persistent (3 per CU) and pulls virtual blocks of the same XCD-aware map IN ORDER from the queue of the XCD it runs on
// (one device-scope atomic per chunk, issued alongside the backward-hop loads of the chunk before; an XCD whose queue is empty helps the
// next one, so results never depend on placement).  A static walk b, b + grid, ... was measured first (profiles/r03_pipe_static_walk.log): workgroups drift apart, the set in
// flight on an XCD stops being a contiguous window of the sweep, the L2 hit rate falls from 0.64 to 0.46 and the kernel becomes
// HBM-bound on 1.6x the traffic.  The in-order queue reproduces what the hardware dispatcher gives variant 1 for free.
//   * the forward-hop operands of chunk c+1 (and the diagonal term / old r of chunk c) are issued BEFORE the barrier, the LDS combine and the
//     stores of chunk c, and land in the registers chunk c's partial sums just left -- one of the two dependent memory round trips of a chunk
//     overlaps the workgroup's synchronisation phase;
//   * the map and neighbour arithmetic of chunk c+1 runs while the backward-hop loads of chunk c are in flight, most of it on the scalar unit:
//     the kernel requires t-slices and z-planes made of whole 64-site chunks, so t, z and the y-chunk of a workgroup's sites are wave-uniform,
//     the z and t neighbours of a chunk are whole chunks (scalar base address, scalar boundary sign) and only the x / y waves do per-lane
//     index arithmetic (one magic-number division by XH);
//   * no workgroup launch, kernel-argument load or 100-SGPR argument block between two chunks: the kernel takes a compact argument struct
//     (PipeArgs), every index into it is static, every load is scalar base + 32-bit lane offset + immediate.
// No branch around a load (a hop that leaves the rank is multiplied by its sign 0, like variant 8); same operations in the same order as variant 1 per
// site => bit-identical Dslash output.  The |out|^2 partial of a workgroup is accumulated per lane over its chunks in double precision (one
// partial per persistent workgroup: stencil_num_blocks).
#ifndef LQCD_PIPE_DEARLY
#define LQCD_PIPE_DEARLY 0      // 1: the diagonal term / old r of a chunk are issued ahead of its backward-hop operands (live across the hop)
#endif
#ifndef LQCD_PIPE_NOPF
#define LQCD_PIPE_NOPF 0        // 1 (experiment): no prefetch across the barrier
#endif
struct PipeArgs {
    const real2* gauge;       // the 18-real field, or the 12-real copy (template R12)
    const real2* clover;      // CINV instances (StencilCall::clover_on_hop): packed 6x6 blocks applied to the hop sum
    real2* dst[2];            // out, or r in update mode (read and written)
    const real2* in[2];
    const real2* xin[2];
    double* norm_partial;
    const double* upd_scal;   // update mode (see StencilCall)
    const double* skip;
    double* scal_w;
    real a, b;
    int nt_store;
    int nvirt, both, pmode;
    int XH, L1, L2, LT, nch;
    FastDiv dXH;
    real sgn_f[4], sgn_b[4];  // sign a hop takes when it wraps the local lattice (0: the neighbour is on another rank)
    int cps, cpp, cpr, per_pass, ty, tz, ysplit;
    FastDiv d_perpass, d_cpr, d_ysplit, d_ty, d_cpp;
    unsigned* ctr;            // queue heads (ctr[32 q], q = 0..7) and exit counter (ctr[256]); zero at launch, reset by the last workgroup;
                              // nullptr: no queue -- workgroup b walks the per_wg consecutive virtual blocks b * per_wg .. (dslash_pipe = 3)
    int per_wg;
    const real2* dotz[2];     // dot mode of the scalar-addressing kernel (StencilCall::dot_z, see KArgs)
    const real2* dotz2[2];    // StencilCall::dot_z2: second inner product of the dot epilogue (five values per workgroup)
    double* dot_partial;
    int dot_conj;
    // DW5 instance (Domainwall operator): L5 slices per launch, block b -> (virtual block, slice) with the XCD of b kept (see wilson_dirsplit_s)
    int ls;
    unsigned long long slice_bytes;      // bytes between the same parity block of consecutive slices
    FastDiv d_ls;
    real dw_mass;
    // FOLD instance (partitioned lattice, one-stream schedule: the exchange is complete when the kernel starts): the hops that leave the rank are taken
    // from the ghost buffers by the SAME launch -- no exterior kernel, complete |.|^2 partials.  Bit mu of fold: direction mu (1, 2, 3) is partitioned.
    int fold;
    int fsel;                 // 0: every chunk; 1: bulk chunks only (no site on a partitioned face), 2: boundary chunks only (KArgs::fsel)
    const int* vlist;         // fsel launches: virtual block of workgroup b (KArgs::vlist); null: b itself
    const real2* gh_f[4];     // ghost of the forward hop at the upper face: P psi(n + mu) packed by the +mu neighbour ([slot][6][Fh], HArgs::recv_fwd)
    const real2* gh_b[4];     // ghost of the backward hop at the lower face: U^+ P psi(n - mu) from the -mu neighbour (HArgs::recv_bwd)
    int Fh[4];                // sites of a face per parity
    real gsf[4], gsb[4];      // sign of a ghost hop: the boundary condition where this rank sits on the global boundary, else 1
};

// next virtual block for this workgroup: from queue q (virtual blocks 8 j + q, j = 0 .. nvirt/8 - 1, handed out in order), moving on to the
// next queue when one is exhausted; -1 when all eight are.  Called by ONE lane.
__device__ inline int pipe_fetch(unsigned* ctr, int nper, int& q, int& tries) {
    while (tries < 8) {
        const unsigned j = __hip_atomic_fetch_add(ctr + 32 * q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j < (unsigned)nper) return 8 * (int)j + q;
        q = (q + 1) & 7; tries++;
    }
    return -1;
}

__device__ inline int fdiv_nb(int n, const FastDiv& f) {      // branch-free form for the scalar unit (s_mul_hi + shift + select)
    const int q = (int)(__umulhi((unsigned)n, f.m) >> f.sh);
    return f.d == 1 ? n : q;
}

// virtual block -> parity, t, z, chunk index inside the z-plane: the arithmetic of map_block_v (remap 2), all on wave-uniform values
__device__ inline void pipe_map(const PipeArgs& a, int b, int& p, int& t, int& z, int& yc) {
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

// one chunk's addressing for the wave of direction MU: byte offsets (32-bit, inside a parity block) of the lane's own site, of its two
// neighbours and of the two links, and the two boundary signs
struct PipeSite {
    unsigned own, nf, nb;      // spinor byte offsets: own site, forward / backward neighbour (other parity)
    unsigned uf, ub;           // link byte offsets inside the gauge field's parity block
    real sf, sb;
    int p;
    bool wf, wb;               // the forward / backward hop of this direction wraps the local lattice (MU >= 2: wave-uniform)
    int fidx;                  // index of the site inside the face of direction MU (coords_to_face), MU >= 1
    int chunk;                 // 64-site chunk inside the parity block (wave-uniform)
};
template <int MU, int NL>      // NL: 16-byte elements per link (9: 18 reals, 6: 12 reals)
__device__ inline PipeSite pipe_site(const PipeArgs& a, int b, int lane) {
    constexpr unsigned SPC = 12 * 64 * sizeof(real2);       // bytes of a spinor chunk
    constexpr unsigned LKC = 4 * NL * 64 * sizeof(real2);   // bytes of a gauge chunk (4 directions)
    constexpr unsigned LKM = NL * 64 * sizeof(real2);       // bytes of one direction inside it
    // bytes per lane inside a link component: 16 (fp64 element; fp32 pair of the 12-real copy) or 8 (fp32 element of the 18-real field)
    constexpr unsigned LKL = (NL == 6 || sizeof(real2) == 16) ? 16u : 8u;
    PipeSite s;
    int t, z, yc;
    pipe_map(a, b, s.p, t, z, yc);
    const int chunk = t * a.cps + z * a.cpp + yc;
    s.own = (unsigned)chunk * SPC + (unsigned)lane * 16u;
    s.chunk = chunk;
    s.uf = (unsigned)chunk * LKC + MU * LKM + (unsigned)lane * LKL;
    if constexpr (MU >= 2) {      // the neighbour of a chunk is a chunk: everything but the lane term is wave-uniform
        const int c = MU == 2 ? z : t, Lc = MU == 2 ? a.L2 : a.LT, st = MU == 2 ? a.cpp : a.cps;
        const bool wf = c == Lc - 1, wb = c == 0;
        const int cf = wf ? chunk - (Lc - 1) * st : chunk + st;
        const int cb = wb ? chunk + (Lc - 1) * st : chunk - st;
        s.nf = (unsigned)cf * SPC + (unsigned)lane * 16u;
        s.nb = (unsigned)cb * SPC + (unsigned)lane * 16u;
        s.ub = (unsigned)cb * LKC + MU * LKM + (unsigned)lane * LKL;
        s.sf = wf ? a.sgn_f[MU] : real(1.0);
        s.sb = wb ? a.sgn_b[MU] : real(1.0);
        s.wf = wf; s.wb = wb;
        s.fidx = ((MU == 2 ? t : z) * a.cpp + yc) * 64 + lane;      // z face: (x, y, t); t face: (x, y, z) -- the z-plane index of the site + the plane count
    } else {
        const int cbp = yc * 64 + lane;               // index inside the z-plane
        const int y = fdiv(cbp, a.dXH), xh = cbp - y * a.XH;
        const int i = chunk * 64 + lane;
        int nf, nb;
        bool wf, wb;
        if constexpr (MU == 0) {
            const int q = (y + z + t + s.p) & 1;      // x = 2 xh + q
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
        s.ub = (unsigned)(nb >> 6) * LKC + MU * LKM + (unsigned)(nb & 63) * LKL;
        s.sf = wf ? a.sgn_f[MU] : real(1.0);
        s.sb = wb ? a.sgn_b[MU] : real(1.0);
        s.wf = wf; s.wb = wb;
        s.fidx = xh + a.XH * (z + a.L2 * t);          // y face: (x, z, t)
    }
    return s;
}

template <typename T>
__device__ inline const real2* boff64(const T* base, unsigned long long bytes) { return reinterpret_cast<const real2*>(reinterpret_cast<const char*>(base) + bytes); }
template <typename T>
__device__ inline const real2* boff(const T* base, unsigned bytes) { return reinterpret_cast<const real2*>(reinterpret_cast<const char*>(base) + bytes); }

// h *= sign, skipped when no lane of the wave has a sign other than 1 (x * 1 = x: same bits either way)
__device__ inline void pipe_sign(cd (&h0)[3], cd (&h1)[3], real sign) {
    if (__builtin_amdgcn_ballot_w64(sign != real(1.0)) != 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { h0[c] = sign * h0[c]; h1[c] = sign * h1[c]; }
    }
}

template <int MU, bool DAG, bool R12, bool NTB>
__device__ inline void pipe_wave(const PipeArgs& a, real2 (*part)[12][64], volatile int* nextvb, int vb, int q, int tries, int lane,
                                 real al_upd, double& nrm_acc) {
    constexpr int SF = DAG ? -1 : 1;
    constexpr int NS = MU == 3 ? 6 : 12;                       // t: only the two rows the projector keeps
    constexpr int FF = MU == 3 ? (SF > 0 ? 6 : 0) : 0;         // first component of the forward / backward hop's rows
    constexpr int FB = MU == 3 ? (SF > 0 ? 0 : 6) : 0;
    constexpr int NL = R12 ? 6 : 9;
    const int nper = a.nvirt >> 3;
    const size_t gpar = (size_t)a.nch * 4 * NL * 64;           // elements of one parity block of the gauge field
    PipeSite s = pipe_site<MU, NL>(a, vb, lane);
    cd sF[NS], uF[9];
    load_comps12<FF, NS, false>(sF, boff(s.p ? a.in[0] : a.in[1], s.nf));
    load_link_any<R12, false>(uF, boff(a.gauge + (s.p ? gpar : 0), s.uf), 64);
    for (;;) {
        cd acc[12], chi0[3], chi1[3], h0[3], h1[3];
#pragma unroll
        for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
        // forward hop: its operands were issued one stage ago (prologue, or in front of the previous chunk's second barrier)
        finish_link<R12>(uF);
        project_regs<MU, SF>(h0, h1, sF);
        pipe_sign(h0, h1, s.sf);
        su3_mv<false>(chi0, uF, h0);
        su3_mv<false>(chi1, uF, h1);
        reconstruct<MU, SF>(acc, chi0, chi1);
        __builtin_amdgcn_sched_barrier(0);
        cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)}, rv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
#if LQCD_PIPE_DEARLY
        if (a.upd_scal) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dst[1] : a.dst[0], s.own) + co12(3 * MU + cc));
        }
        if (a.a != real(0.0)) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) xv[cc] = ld(boff(s.p ? a.xin[1] : a.xin[0], s.own) + co12(3 * MU + cc));
        }
#endif
        // backward hop's operands, into the registers the forward operands occupied
        cd sB[NS], uB[9];
        load_comps12<FB, NS, false>(sB, boff(s.p ? a.in[0] : a.in[1], s.nb));
        load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
        // the NEXT chunk: one lane of the x wave pulls it from the queue now -- as late as its answer can still arrive for free (it returns
        // with the loads above).  Pulling earlier lets the order in which chunks are handed out drift away from the order in which their
        // loads are issued, and the L2 hits of the sweep live on that order (profiles/r03_pipe_lookahead.log).
        unsigned tick = 0;
        if constexpr (MU == 0) {
            if (a.ctr && lane == 0 && tries < 8) tick = __hip_atomic_fetch_add(a.ctr + 32 * q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_sched_barrier(0);
        finish_link<R12>(uB);
        project_regs<MU, -SF>(h0, h1, sB);
        pipe_sign(h0, h1, s.sb);
        su3_mv<true>(chi0, uB, h0);
        su3_mv<true>(chi1, uB, h1);
        reconstruct<MU, -SF>(acc, chi0, chi1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MU == 0) {
            if (lane == 0) {
                int nx = -1;
                if (!a.ctr) {                 // static walk of per_wg consecutive virtual blocks (hardware dispatch order between workgroups)
                    nx = (vb + 1) % a.per_wg != 0 ? vb + 1 : -1;
                } else if (tries < 8) {
                    if (tick < (unsigned)nper) nx = 8 * (int)tick + q;
                    else { q = (q + 1) & 7; tries++; nx = pipe_fetch(a.ctr, nper, q, tries); }      // queue exhausted: help the next XCD (tail only)
                }
                nextvb[2] = nx;
            }
        }
        // every wave has consumed the previous chunk's partial sums (its LDS reads fed its stores): the array may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < 12; j++) part[MU][j][lane] = mk2(acc[j].re, acc[j].im);
        // the next chunk's addressing (mostly scalar); the last chunk computes itself again and discards the loads
        const int vbn = __builtin_amdgcn_readfirstlane(nextvb[2]);    // written in front of the barrier above, next written in front of the next chunk's
        const bool more = vbn >= 0;
        const PipeSite sn = pipe_site<MU, NL>(a, more ? vbn : vb, lane);
        __builtin_amdgcn_sched_barrier(0);
        // this chunk's diagonal term / old r, then the NEXT chunk's forward operands: all in flight across the barrier and the LDS combine
#if !LQCD_PIPE_DEARLY
        if (a.upd_scal) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dst[1] : a.dst[0], s.own) + co12(3 * MU + cc));
        }
        if (a.a != real(0.0)) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) xv[cc] = ld(boff(s.p ? a.xin[1] : a.xin[0], s.own) + co12(3 * MU + cc));
        }
#endif
#if !LQCD_PIPE_NOPF
        load_comps12<FF, NS, false>(sF, boff(sn.p ? a.in[0] : a.in[1], sn.nf));
        load_link_any<R12, false>(uF, boff(a.gauge + (sn.p ? gpar : 0), sn.uf), 64);
#endif
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // LDS writes done; NO vmcnt wait: the loads above stay in flight
        __builtin_amdgcn_sched_barrier(0);
        {
            real nrm = 0.0;
            real2* dstp = const_cast<real2*>(boff(s.p ? a.dst[1] : a.dst[0], s.own));
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
                const int j = 3 * MU + cc;
                const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
                cd sm = mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
                cd v = a.b * sm;
                v = mk(fma(a.a, xv[cc].re, v.re), fma(a.a, xv[cc].im, v.im));
                if (a.upd_scal) {
                    cd r = rv[cc];
                    r.re = fma(-al_upd, v.re, r.re); r.im = fma(-al_upd, v.im, r.im);
                    nrm = fma(r.re, r.re, nrm); nrm = fma(r.im, r.im, nrm);
                    st(dstp + co12(j), r);
                } else {
                    nrm = fma(v.re, v.re, nrm); nrm = fma(v.im, v.im, nrm);
                    if (a.nt_store) st_nt(dstp + co12(j), v); else st(dstp + co12(j), v);
                }
            }
            nrm_acc += (double)nrm;
        }
        if (!more) break;
        vb = vbn;
        s = sn;
#if LQCD_PIPE_NOPF      // experiment: persistent only, the forward operands are issued at the top of the chunk like variant 1
        load_comps12<FF, NS, false>(sF, boff(s.p ? a.in[0] : a.in[1], s.nf));
        load_link_any<R12, false>(uF, boff(a.gauge + (s.p ? gpar : 0), s.uf), 64);
#endif
    }
}

#ifndef LQCD_PIPE_OCC
#ifdef LQCD_F32
#define LQCD_PIPE_OCC (R12 ? 5 : 4)     // 95 VGPRs with the 12-real links (what the mixed-precision solvers use); the 18-real instance would spill at 96
#else
#define LQCD_PIPE_OCC 3
#endif
#endif
#ifndef LQCD_SDIR_BOTH32
#define LQCD_SDIR_BOTH32 0      // fp32 build: 1 = both hops' operands of a direction in flight at once (110 VGPRs, 4 waves per SIMD)
#endif
#if defined(LQCD_F32) && LQCD_SDIR_BOTH32
#define LQCD_SDIR_BOTH 1
#else
#define LQCD_SDIR_BOTH 0
#endif
#ifdef LQCD_F32
#ifndef LQCD_SDIR_OCC32
#define LQCD_SDIR_OCC32 (R12 ? 5 : 4)
#endif
#define LQCD_DS_BOUNDS_S __launch_bounds__(256, LQCD_SDIR_OCC32)
#else
#define LQCD_DS_BOUNDS_S __launch_bounds__(256, 3)
#endif
#ifdef LQCD_F32
static constexpr bool kF32Build = true;
#else
static constexpr bool kF32Build = false;
#endif
template <bool DAG, bool R12, bool NTB>
__global__ __launch_bounds__(256, LQCD_PIPE_OCC) void wilson_dirsplit_pipe(PipeArgs a) {
    __shared__ real2 part[4][12][64];  // 48 KiB (fp32: 24)
    __shared__ double red[4];
    if ((a.upd_scal && a.upd_scal[S_DONE] != 0.0) || (a.skip && a.skip[S_DONE] != 0.0)) {
        // folded scalar steps: the update-mode launch of an overshooting iteration tells the x/p update behind it that the converging iterate
        // is complete (upd_done in stencil_common.h)
        if (a.scal_w && blockIdx.x == 0 && threadIdx.x == 0) a.scal_w[S_XDONE] = 1.0;
        return;
    }
    real al_upd = real(0);
    if (a.upd_scal) {
        if (a.scal_w) {      // folded scalar step (several ranks): see update_alpha
            const double rr = a.upd_scal[S_RR];
            const double al = rr / a.upd_scal[S_PQ];
            if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal_w[S_ALPHA] = al; a.scal_w[S_RROLD] = rr; }
            al_upd = (real)al;
        } else al_upd = (real)a.upd_scal[S_ALPHA];
    }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    __shared__ int nextvb[4];
    // the queue of the XCD this workgroup runs on (speed only: any value 0..7 gives the same results)
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    int q = (int)(xcc & 7u), tries = 0;
    if (threadIdx.x == 0) nextvb[0] = a.ctr ? pipe_fetch(a.ctr, a.nvirt >> 3, q, tries) : (int)blockIdx.x * a.per_wg;
    __syncthreads();
    const int vb0 = __builtin_amdgcn_readfirstlane(nextvb[0]);
    double nrm = 0.0;
    if (vb0 >= 0) {
        switch (w) {
        case 0: pipe_wave<0, DAG, R12, NTB>(a, part, nextvb, vb0, q, tries, lane, al_upd, nrm); break;
        case 1: pipe_wave<1, DAG, R12, NTB>(a, part, nextvb, vb0, q, tries, lane, al_upd, nrm); break;
        case 2: pipe_wave<2, DAG, R12, NTB>(a, part, nextvb, vb0, q, tries, lane, al_upd, nrm); break;
        default: pipe_wave<3, DAG, R12, NTB>(a, part, nextvb, vb0, q, tries, lane, al_upd, nrm); break;
        }
    }
    // this workgroup will not touch the queues again; the last one to say so zeroes them for the next launch (stream order)
    if (threadIdx.x == 0 && a.ctr) {
        const unsigned d = __hip_atomic_fetch_add(a.ctr + 256, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            for (int k = 0; k < 9; k++) __hip_atomic_store(a.ctr + 32 * k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (a.norm_partial) {
        nrm = wave_sum(nrm);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) a.norm_partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

#ifndef LQCD_SDIR_GLDS
#define LQCD_SDIR_GLDS 0        // 1 (fp64 build, experiment of round 4): the backward neighbour's spinor travels global -> LDS with gfx950's asynchronous copy while the
#endif                          // forward hop runs.  Bit-identical, and NO faster (0.3472 vs 0.3480 ms at 32^3x64, profiles/r04_glds_ab.log): with both hops' spinor
                                // operands in flight at unchanged occupancy the kernel time does not move -- the launch is bound by the fabric's miss traffic
                                // (5.4 TB/s of the ~6.3 TB/s achievable on 1.18x the compulsory bytes), not by a workgroup's dependent round trips
// ------------------------------------------------------------------------------------------ Wilson, direction-split, scalar addressing
// dslash_pipe = 2: variant 1's schedule (one workgroup per chunk, hardware dispatch order, the compiler's own hop-by-hop schedule) on the
// persistent kernel's addressing: compact argument struct, t / z / y-chunk wave-uniform, scalar base + 32-bit lane offset + immediate for every
// load, no branch around a load.  Per launch -20 % VALU and -42 % SALU instructions than variant 1 (profiles/r03_pmc_pipe_static.csv), i.e. a
// shorter way from dispatch to the first load.  Same operations in the same order per site, same |.|^2 partial per workgroup: bit-identical to
// variant 1 including the CG iterates.
// FOLD instances: the operands of a hop that leaves the rank.  The ghost of a face site holds the SPIN-PROJECTED half spinor (6 components, stride Fh) the pack
// kernels wrote: it is loaded into components 0..5 and the other six are zero, so the projection that follows reproduces it (h = ghost + i^k 0; t direction:
// the factor 2 of the projection is undone by the sign, which takes a factor 1/2 -- exact).  The backward ghost is U^+ P psi already: its link is the unit
// matrix (rows 0, 1; row 2 rebuilt or set), so su3_mv returns h.  Same instructions for face and bulk lanes, no exterior kernel, complete |.|^2 partials.
// z / t faces are whole chunks (face is wave-uniform: a scalar branch around the loads); the y face is one row of four inside a chunk (per-lane select).
template <int MU, int NS, int F0>
__device__ inline void fold_load_spinor(cd* sp, const real2* __restrict__ reg, const real2* __restrict__ ghost, int Fh, bool face) {
    if constexpr (MU >= 2) {
        if (face) {
#pragma unroll
            for (int j = 0; j < 6; j++) sp[j] = ld(ghost + (size_t)j * Fh);
#pragma unroll
            for (int j = 6; j < NS; j++) sp[j] = mk(0.0, 0.0);
        } else load_comps12<F0, NS, false>(sp, reg);
    } else {
        const real2* __restrict__ src = face ? ghost : reg + co12(F0);
        const size_t st = face ? (size_t)Fh : co12(1);
#pragma unroll
        for (int j = 0; j < 6; j++) sp[j] = ld(src + (size_t)j * st);
#pragma unroll
        for (int j = 6; j < NS; j++) {
            const cd v = ld(reg + co12(F0 + j));
            sp[j].re = face ? real(0.0) : v.re;      // (component by component: a select between two structs is lowered through scratch memory)
            sp[j].im = face ? real(0.0) : v.im;
        }
    }
}
template <int MU, bool R12>
__device__ inline void fold_unit_link(cd (&u)[9], bool face) {      // (called with the link loaded: MU == 1 selects per lane, MU >= 2 overwrites under the scalar branch)
    constexpr int N = R12 ? 6 : 9;
#pragma unroll
    for (int j = 0; j < N; j++) {
        const real dg = (j == 0 || j == 4 || j == 8) ? real(1.0) : real(0.0);
        u[j].re = face ? dg : u[j].re;
        u[j].im = face ? real(0.0) : u[j].im;
    }
}

template <int MU, bool DAG, bool R12, bool NTB, bool DOT = false, bool DELTA = false, bool DW5 = false, bool FOLD = false, bool CINV = false>
__device__ inline void sdir_wave(const PipeArgs& a_, real2 (*part)[12][64], int lane, real al_upd, real& nrm, real& dre, real& dim, int vb, real& dre2, real& dim2) {
    // DW5: this workgroup's slice s5 of the five-dimensional fields; block ids keep their XCD (b & 7) and the L5 slices of a chunk follow each other on it, so the
    // links of the chunk are fetched from the fabric once and hit the XCD's L2 for the other slices
    int vblock = vb, s5 = 0;      // (vb: blockIdx.x, or the entry of the bulk / boundary list of a folded launch)
    if constexpr (DW5) {
        const int g8 = (int)(blockIdx.x >> 3), gq = fdiv_nb(g8, a_.d_ls);
        s5 = g8 - gq * a_.ls;
        vblock = gq * 8 + (int)(blockIdx.x & 7);
    }
    PipeArgs a5;
    if constexpr (DW5) {
        a5 = a_;
        const unsigned long long off = (unsigned long long)s5 * a_.slice_bytes;
        a5.in[0] = boff64(a_.in[0], off); a5.in[1] = boff64(a_.in[1], off);
        a5.xin[0] = boff64(a_.xin[0], off); a5.xin[1] = boff64(a_.xin[1], off);
        a5.dst[0] = const_cast<real2*>(boff64(a_.dst[0], off)); a5.dst[1] = const_cast<real2*>(boff64(a_.dst[1], off));
    }
    const PipeArgs& a = DW5 ? a5 : a_;      // (every other instance reads the kernel arguments where they are)
    constexpr int SF = DAG ? -1 : 1;
    constexpr int NS = MU == 3 ? 6 : 12;
    constexpr int FF = MU == 3 ? (SF > 0 ? 6 : 0) : 0;
    constexpr int FB = MU == 3 ? (SF > 0 ? 0 : 6) : 0;
    constexpr int NL = DELTA ? 8 : (R12 ? 6 : 9);      // 16-byte words per link: 18 reals | rows 0, 1 | rows 0, 1 + the fp32 deviation of row 2 (add_delta_row2)
#ifdef LQCD_F32
    constexpr bool LATE_R = false;
#else
    constexpr bool LATE_R = DELTA || !R12;            // fp64: the instances that would spill at 3 waves per SIMD with the old r held across the hops
#endif
    const size_t gpar = (size_t)a.nch * 4 * NL * 64;
    const PipeSite s = pipe_site<MU, NL>(a, vblock, lane);
    cd xv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)}, rv[3] = {mk(0, 0), mk(0, 0), mk(0, 0)};
    const bool z_is_x = DOT && (s.p ? a.dotz[1] == a.xin[1] : a.dotz[0] == a.xin[0]) && a.a != real(0.0);
    const bool two = DOT && z_is_x && (s.p ? a.dotz2[1] : a.dotz2[0]) != nullptr;      // second inner product: z = xin leaves the registers of z to z2
    auto load_z = [&]() {
        if (!z_is_x) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dotz[1] : a.dotz[0], s.own) + co12(3 * MU + cc));
        } else if (two) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dotz2[1] : a.dotz2[0], s.own) + co12(3 * MU + cc));
        }
    };
    if constexpr (DOT) {        // dot mode: z takes the registers the old r has in update mode (requested ahead of the hops, scalar base + lane offset; CINV instances: behind them)
        if constexpr (!CINV) load_z();
    } else if (!LATE_R && a.upd_scal) {    // (18-real and 12 + delta links: their extra words take these registers during the hops; the old r is requested behind them)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dst[1] : a.dst[0], s.own) + co12(3 * MU + cc));
    }
    constexpr bool LATE_X = (LATE_R && !R12 && !DOT) || (CINV && DOT);      // all 18 reals, and the dot instances with the clover blocks in the epilogue: the diagonal term's load moves behind the hops as well
    if (!LATE_X && a.a != real(0.0)) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ld(boff(s.p ? a.xin[1] : a.xin[0], s.own) + co12(3 * MU + cc));
    }
    cd acc[12], chi0[3], chi1[3], h0[3], h1[3];
#pragma unroll
    for (int j = 0; j < 12; j++) acc[j] = mk(0.0, 0.0);
#if LQCD_SDIR_BOTH
    {   // both hops' operands in flight at once (fp32 build: half-size registers leave the room at 4 waves per SIMD; twice the bytes in flight per wave)
        cd sF[NS], uF[9], sB[NS], uB[9];
        load_comps12<FF, NS, false>(sF, boff(s.p ? a.in[0] : a.in[1], s.nf));
        load_link_any<R12, false>(uF, boff(a.gauge + (s.p ? gpar : 0), s.uf), 64);
        load_comps12<FB, NS, false>(sB, boff(s.p ? a.in[0] : a.in[1], s.nb));
        load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
        finish_link<R12>(uF);
        project_regs<MU, SF>(h0, h1, sF);
        pipe_sign(h0, h1, s.sf);
        su3_mv<false>(chi0, uF, h0);
        su3_mv<false>(chi1, uF, h1);
        reconstruct<MU, SF>(acc, chi0, chi1);
        finish_link<R12>(uB);
        project_regs<MU, -SF>(h0, h1, sB);
        pipe_sign(h0, h1, s.sb);
        su3_mv<true>(chi0, uB, h0);
        su3_mv<true>(chi1, uB, h1);
        reconstruct<MU, -SF>(acc, chi0, chi1);
    }
#else
    // FOLD: this lane's forward / backward hop leaves the rank -- its operand is the ghost (MU >= 2: wave-uniform)
    bool gF = false, gB = false;
    real sgF = s.sf, sgB = s.sb;
    if constexpr (FOLD && MU >= 1) {
        if ((a.fold >> MU) & 1) {
            constexpr real half = MU == 3 ? real(0.5) : real(1.0);
            gF = s.wf; gB = s.wb;
            sgF = gF ? half * a.gsf[MU] : s.sf;
            sgB = gB ? half * a.gsb[MU] : s.sb;
        }
    }
    const size_t gslot = FOLD ? (size_t)(a.both ? s.p : 0) * 6 * (size_t)a.Fh[MU] + (size_t)s.fidx : 0;
    {
        cd sF[NS], uF[9], dF[2];
        if constexpr (FOLD && MU >= 1) fold_load_spinor<MU, NS, FF>(sF, boff(s.p ? a.in[0] : a.in[1], s.nf), a.gh_f[MU] + gslot, a.Fh[MU], gF);
        else load_comps12<FF, NS, false>(sF, boff(s.p ? a.in[0] : a.in[1], s.nf));
        load_link_any<R12, false>(uF, boff(a.gauge + (s.p ? gpar : 0), s.uf), 64);
        if constexpr (DELTA) { dF[0] = ld(boff(a.gauge + (s.p ? gpar : 0), s.uf) + 6 * 64); dF[1] = ld(boff(a.gauge + (s.p ? gpar : 0), s.uf) + 7 * 64); }
#if LQCD_SDIR_GLDS
        // The backward neighbour's spinor goes global -> LDS with the asynchronous copy of gfx950 (global_load_lds_dwordx4: wave-uniform LDS base +
        // 16 B x lane, exactly the [component][lane] slab this wave owns in the partial-sum area, which nobody touches before the hops are done):
        // both hops' spinor operands are in flight from the start at NO register cost -- the second dependent memory round trip of a workgroup
        // shrinks to the backward link (the "both hops in flight" of variant 8 without its drop to 2 waves per SIMD).
        {
            const real2* gb = boff(s.p ? a.in[0] : a.in[1], s.nb);
#pragma unroll
            for (int j = 0; j < NS; j++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + co12(FB + j)),
                                                 (__attribute__((address_space(3))) void*)(&part[MU][j][0]), 16, 0, 0);
        }
#endif
        finish_link<R12>(uF);
#ifndef LQCD_F32
        if constexpr (DELTA) add_delta_row2(uF, dF);
#endif
        project_regs<MU, SF>(h0, h1, sF);
        pipe_sign(h0, h1, sgF);
        su3_mv<false>(chi0, uF, h0);
        su3_mv<false>(chi1, uF, h1);
        reconstruct<MU, SF>(acc, chi0, chi1);
    }
    __builtin_amdgcn_sched_barrier(0);      // the backward operands take the registers of the forward ones (3 waves per SIMD)
    {
        cd sB[NS], uB[9], dB[2];
#if LQCD_SDIR_GLDS
        load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the LDS copies (issued long ago) and the link
#pragma unroll
        for (int j = 0; j < NS; j++) sB[j] = ld(&part[MU][j][lane]);
#else
        if constexpr (FOLD && MU >= 2) {
            fold_load_spinor<MU, NS, FB>(sB, boff(s.p ? a.in[0] : a.in[1], s.nb), a.gh_b[MU] + gslot, a.Fh[MU], gB);
            if (gB) fold_unit_link<MU, R12>(uB, true);      // (scalar branch: the ghost is U^+ P psi already, no link is read)
            else load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
        } else if constexpr (FOLD && MU == 1) {
            fold_load_spinor<MU, NS, FB>(sB, boff(s.p ? a.in[0] : a.in[1], s.nb), a.gh_b[MU] + gslot, a.Fh[MU], gB);
            load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
            fold_unit_link<MU, R12>(uB, gB);
        } else {
            load_comps12<FB, NS, false>(sB, boff(s.p ? a.in[0] : a.in[1], s.nb));
            load_link_any<R12, NTB>(uB, boff(a.gauge + (s.p ? 0 : gpar), s.ub), 64);
        }
#endif
        if constexpr (DELTA) {
            const real2* db = boff(a.gauge + (s.p ? 0 : gpar), s.ub);
            dB[0] = NTB ? ld_nt(db + 6 * 64) : ld(db + 6 * 64);
            dB[1] = NTB ? ld_nt(db + 7 * 64) : ld(db + 7 * 64);
        }
        finish_link<R12>(uB);
#ifndef LQCD_F32
        if constexpr (DELTA) add_delta_row2(uB, dB);
#endif
        project_regs<MU, -SF>(h0, h1, sB);
        pipe_sign(h0, h1, sgB);
        su3_mv<true>(chi0, uB, h0);
        su3_mv<true>(chi1, uB, h1);
        reconstruct<MU, -SF>(acc, chi0, chi1);
    }
#endif
    if constexpr (LATE_X) if (a.a != real(0.0)) {
#pragma unroll
        for (int cc = 0; cc < 3; cc++) xv[cc] = ld(boff(s.p ? a.xin[1] : a.xin[0], s.own) + co12(3 * MU + cc));
    }
    if constexpr (LATE_R && !DOT) if (a.upd_scal) {       // the LDS exchange and the barrier cover this load
#pragma unroll
        for (int cc = 0; cc < 3; cc++) rv[cc] = ld(boff(s.p ? a.dst[1] : a.dst[0], s.own) + co12(3 * MU + cc));
    }
#pragma unroll
    for (int j = 0; j < 12; j++) part[MU][j][lane] = mk2(acc[j].re, acc[j].im);
    if constexpr (DOT && CINV) load_z();      // (the barrier and the block products cover the load)
    cd w5[DW5 ? 3 : 1];
    if constexpr (DW5) {      // fifth-direction hops of this wave's three components: -P_A psi(s+1) - P_B psi(s-1), the mass term at the walls; P_-+ psi = (psi -+ g5 psi)/2
                              // and (g5 psi)_spin = -psi_(spin xor 2): the partner component is six further on or back.  Issued behind the stores of the partial sums (round 5: the twelve accumulators are dead by now -- the instance no longer spills): the barrier and the LDS reads cover them
        const int su = s5 + 1 < a_.ls ? s5 + 1 : 0, sd = s5 >= 1 ? s5 - 1 : a_.ls - 1;
        const real cu = real(0.5) * (s5 + 1 < a_.ls ? real(-1.0) : a_.dw_mass), cd_ = real(0.5) * (s5 >= 1 ? real(-1.0) : a_.dw_mass);
        constexpr real sa = DAG ? real(-1.0) : real(1.0);
        const real2* pu = boff(boff64(s.p ? a_.xin[1] : a_.xin[0], (unsigned long long)su * a_.slice_bytes), s.own);
        const real2* pd = boff(boff64(s.p ? a_.xin[1] : a_.xin[0], (unsigned long long)sd * a_.slice_bytes), s.own);
#pragma unroll
        for (int cc = 0; cc < 3; cc++) {
            constexpr int dummy = 0; (void)dummy;
            const int j = 3 * MU + cc, jp = MU < 2 ? j + 6 : j - 6;
            const cd u0 = ld(pu + co12(j)), u1 = ld(pu + co12(jp)), d0 = ld(pd + co12(j)), d1 = ld(pd + co12(jp));
            w5[cc] = mk(cu * (u0.re + sa * u1.re) + cd_ * (d0.re - sa * d1.re), cu * (u0.im + sa * u1.im) + cd_ * (d0.im - sa * d1.im));
        }
    }
    __syncthreads();
    real2* dstp = const_cast<real2*>(boff(s.p ? a.dst[1] : a.dst[0], s.own));
    cd hs[CINV ? 3 : 1];
    if constexpr (CINV) {       // this wave's rows of C (H in): all 12 summed components, then the packed 6x6 blocks (the epilogue of wilson_dirsplit's CINV instances)
        // one chiral block at a time, its six chi components summed from LDS when the block needs them (12 summed components held at once would spill a few registers
        // at three workgroups per CU); the same additions in the same order as clover_rows on a full v12
        auto vsum = [&](int j) {
            const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
            return mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
        };
        const real2* __restrict__ ca = a.clover + (((size_t)s.p * a.nch + (size_t)s.chunk) * 36) * 64 + lane;
        constexpr int S1 = MU & 1;
        constexpr bool lower = MU >= 2;
#pragma unroll
        for (int cc = 0; cc < 3; cc++) hs[cc] = mk(0.0, 0.0);
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const real sg = b == 0 ? real(-1.0) : real(1.0);
            cd chi[6];
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const cd u = vsum(q), l = vsum(6 + q);
                chi[q] = mk(u.re + sg * l.re, u.im + sg * l.im);
                if (q & 1) __builtin_amdgcn_sched_barrier(0);      // (16 LDS reads in flight at a time: the scheduler otherwise batches all 48 of a block -- and the next block's -- and spills)
            }
            const real2* __restrict__ ab = ca + (size_t)(18 * b) * 64;
            const real wgt = real(0.5) * ((b == 0 && lower) ? real(-1.0) : real(1.0));
#pragma unroll
            for (int cc = 0; cc < 3; cc++) {
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
                hs[cc] = mk(hs[cc].re + wgt * y.re, hs[cc].im + wgt * y.im);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
        const int j = 3 * MU + cc;
        const real2 s0 = part[0][j][lane], s1 = part[1][j][lane], s2 = part[2][j][lane], s3 = part[3][j][lane];
        cd sm = CINV ? hs[cc] : mk((s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y));
        cd v = a.b * sm;
        v = mk(fma(a.a, xv[cc].re, v.re), fma(a.a, xv[cc].im, v.im));
        if constexpr (DW5) v = mk(v.re + w5[cc].re, v.im + w5[cc].im);
        if constexpr (DOT) {        // <z, v> = conj(z) v next to |v|^2 (same expressions as wilson_dirsplit's dot epilogue)
            const cd z = z_is_x ? xv[cc] : rv[cc];
            nrm = fma(v.re, v.re, nrm); nrm = fma(v.im, v.im, nrm);
            if (a.nt_store) st_nt(dstp + co12(j), v); else st(dstp + co12(j), v);
            dre = fma(z.re, v.re, dre); dre = fma(z.im, v.im, dre);
            dim = fma(z.re, v.im, dim); dim = fma(-z.im, v.re, dim);
            if (two) {
                const cd z2 = rv[cc];
                dre2 = fma(z2.re, v.re, dre2); dre2 = fma(z2.im, v.im, dre2);
                dim2 = fma(z2.re, v.im, dim2); dim2 = fma(-z2.im, v.re, dim2);
            }
        } else if (a.upd_scal) {
            cd r = rv[cc];
            r.re = fma(-al_upd, v.re, r.re); r.im = fma(-al_upd, v.im, r.im);
            nrm = fma(r.re, r.re, nrm); nrm = fma(r.im, r.im, nrm);
            st(dstp + co12(j), r);
        } else {
            nrm = fma(v.re, v.re, nrm); nrm = fma(v.im, v.im, nrm);
            if (a.nt_store) st_nt(dstp + co12(j), v); else st(dstp + co12(j), v);
        }
    }
}

template <bool DAG, bool R12, bool NTB, bool DOT = false, bool DELTA = false, bool DW5 = false, bool FOLD = false, bool CINV = false>
__global__ LQCD_DS_BOUNDS_S void wilson_dirsplit_s(PipeArgs a) {
    __shared__ real2 part[4][12][64];  // 48 KiB (fp32: 24)
    __shared__ double red[DOT ? 20 : 4];
    if ((a.upd_scal && a.upd_scal[S_DONE] != 0.0) || (a.skip && a.skip[S_DONE] != 0.0)) {
        if (a.scal_w && blockIdx.x == 0 && threadIdx.x == 0) a.scal_w[S_XDONE] = 1.0;
        return;
    }
    real al_upd = real(0);
    if (a.upd_scal) {
        if (a.scal_w) {      // folded scalar step (several ranks): see update_alpha
            const double rr = a.upd_scal[S_RR];
            const double al = rr / a.upd_scal[S_PQ];
            if (blockIdx.x == 0 && threadIdx.x == 0) { a.scal_w[S_ALPHA] = al; a.scal_w[S_RROLD] = rr; }
            al_upd = (real)al;
        } else al_upd = (real)a.upd_scal[S_ALPHA];
    }
    int vb = (int)blockIdx.x;
    if constexpr (FOLD) {
        if (a.vlist) vb = a.vlist[blockIdx.x];
        if (a.fsel) {      // bulk / boundary launch of an overlapping schedule: t, z and the y rows of a chunk are workgroup-uniform.  (The launch covers the chunks of its
                           // kind only -- vlist, built by fold_classify_s with this very arithmetic; the test stays as the authority)
            int p_, t, z, yc;
            pipe_map(a, vb, p_, t, z, yc);
            const int y0 = fdiv(yc * 64, a.dXH), y1 = fdiv(yc * 64 + 63, a.dXH);
            const bool bnd = (((a.fold >> 3) & 1) && (t == 0 || t == a.LT - 1)) || (((a.fold >> 2) & 1) && (z == 0 || z == a.L2 - 1)) ||
                             (((a.fold >> 1) & 1) && (y0 == 0 || y1 == a.L1 - 1));
            if ((a.fsel == 1) == bnd) return;
        }
    }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    real nrm = 0.0, dre = 0.0, dim = 0.0, dre2 = 0.0, dim2 = 0.0;
    switch (w) {
    case 0: sdir_wave<0, DAG, R12, NTB, DOT, DELTA, DW5, FOLD, CINV>(a, part, lane, al_upd, nrm, dre, dim, vb, dre2, dim2); break;
    case 1: sdir_wave<1, DAG, R12, NTB, DOT, DELTA, DW5, FOLD, CINV>(a, part, lane, al_upd, nrm, dre, dim, vb, dre2, dim2); break;
    case 2: sdir_wave<2, DAG, R12, NTB, DOT, DELTA, DW5, FOLD, CINV>(a, part, lane, al_upd, nrm, dre, dim, vb, dre2, dim2); break;
    default: sdir_wave<3, DAG, R12, NTB, DOT, DELTA, DW5, FOLD, CINV>(a, part, lane, al_upd, nrm, dre, dim, vb, dre2, dim2); break;
    }
    if constexpr (DOT) {                // three sums per workgroup, the order of wilson_dirsplit's dot epilogue; five with a second inner product (dot_z2)
        const bool five = a.dotz2[0] != nullptr || a.dotz2[1] != nullptr;
        double t3[5] = {(double)dre, (double)((a.dot_conj & 1) ? -dim : dim), (double)nrm, (double)dre2, (double)dim2};
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
        nrm = wave_sum(nrm);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) a.norm_partial[vb] = (red[0] + red[1]) + (red[2] + red[3]);
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
template <bool R12, bool BOTH, bool NTB, bool FOLD>
__device__ __forceinline__ void staggered_dirsplit_body(const KArgs& k, const HArgs* hf) {
    __shared__ real2 part[4][3][64];
    __shared__ double red[4];
    if (upd_done(k)) return;
    const real al_upd = update_alpha(k);
    int chunk, p;
    map_block(k, chunk, p);
    if constexpr (FOLD) { if (fold_chunk_skipped(k, chunk, p)) return; }
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
        if constexpr (FOLD) {      // folded halo schedule: this direction's hops that leave the rank, from the ghost buffers (the exterior kernel's arithmetic)
            const int slot = k.parity_mode == 2 ? p : 0;
            switch (w) {
            case 0: staggered_ext_add<0>(acc, *hf, c, slot, p, i); break;
            case 1: staggered_ext_add<1>(acc, *hf, c, slot, p, i); break;
            case 2: staggered_ext_add<2>(acc, *hf, c, slot, p, i); break;
            default: staggered_ext_add<3>(acc, *hf, c, slot, p, i); break;
            }
        }
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
        nrm = wave_sum(nrm);
        if (lane == 0) red[w] = nrm;
        __syncthreads();
        if (threadIdx.x == 0) k.norm_partial[vblock_of(k)] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

template <bool R12, bool BOTH = false, bool NTB = false>
__global__ __launch_bounds__(256) void staggered_dirsplit(KArgs k) { staggered_dirsplit_body<R12, BOTH, NTB, false>(k, nullptr); }
template <bool R12>
__global__ __launch_bounds__(256) void staggered_dirsplit_fold(KArgs k, HArgs h) { staggered_dirsplit_body<R12, false, false, true>(k, &h); }

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

#ifndef LQCD_F32
// Pack launch of the folded schedule with the final reduction of the application's |.|^2 partials riding along (fused CG: D p -> [pack D p for D^+ | sum |D p|^2] ->
// exchange -> D^+): grid (x, 9), row 0 = ONE reduction block (dispatched first), rows 1..8 = the pack blocks of wilson_pack with 256 threads each.  The reduction
// reproduces reduce_final (blas.hip) bit for bit -- small sums: the one-wave order; large sums: thread t owns the classes t, t + 256, t + 512, t + 768 of the
// 1024-thread kernel, one __shfl_down tree per class group, the 16 wave sums added in sequence -- so iterates do not depend on which launch did the sum.
__global__ __launch_bounds__(256) void wilson_pack_reduce(HArgs k, const double* __restrict__ partial, int nblocks, double* scal, int slot, int op, PeerRedArgs pr) {
    if (blockIdx.y == 0) {
        if (blockIdx.x != 0) return;
        __shared__ double red[FB / 64];
        __shared__ double tot;
        // pr.nranks > 0 (peer-mapped backend): wave 0 adds the ranks' sums right here (comm.hip peer_allreduce_wave), beside the pack blocks
        if (nblocks > 1024) {
            const int w = (int)threadIdx.x >> 6;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double t = shfl_tree_sum(sum_partials_class(partial, nblocks, 1, 0, (int)threadIdx.x + 256 * q));
                if ((threadIdx.x & 63) == 0) red[w + 4 * q] = t;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0;
                for (int j = 0; j < FB / 64; j++) t += red[j];
                tot = t;
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            double t = nblocks > 1024 ? tot : sum_partials_small_nv(partial, nblocks, 1, 0);
            if (pr.nranks) t = __shfl(peer_allreduce_wave(pr, t, 1), 0, 64);
            if (threadIdx.x == 0) { scal[slot] = t; if (op) cg_scalar_step(scal, op); }
        }
        return;
    }
    const int y = (int)blockIdx.y - 1, side = y & 1;
    switch (y >> 1) {
    case 0: wilson_pack_dir<0>(k, side); break;
    case 1: wilson_pack_dir<1>(k, side); break;
    case 2: wilson_pack_dir<2>(k, side); break;
    default: wilson_pack_dir<3>(k, side); break;
    }
}
#endif

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
    cd fin[12];
#pragma unroll
    for (int j = 0; j < 12; j++) {
        cd v = ld(o + co12(j));
        const real before = v.re * v.re + v.im * v.im;
        v.re = fma(coef, acc[j].re, v.re);
        v.im = fma(coef, acc[j].im, v.im);
        corr += (v.re * v.re + v.im * v.im) - before;
        st(o + co12(j), v);
        fin[j] = v;
    }
    if (k.pack_next >= 0) {
        // the finished site is an input site of the next application: its packed faces belong to the slot of the RECEIVING parity 1 - pout
        const int nslot = k.parity_mode == 2 ? 1 - pout : 0;
        wilson_pack_site<0>(k, fin, c, nslot, pout, i, k.pack_next);
        wilson_pack_site<1>(k, fin, c, nslot, pout, i, k.pack_next);
        wilson_pack_site<2>(k, fin, c, nslot, pout, i, k.pack_next);
        wilson_pack_site<3>(k, fin, c, nslot, pout, i, k.pack_next);
    }
    return corr;
}

// block-level sum of the per-thread norm corrections of an exterior kernel (128 threads); with red_out the last block to arrive sums every
// partial of the application (interior blocks + these corrections) in a fixed order -- the one-block reduce_final launch behind the
// exterior disappears from the critical path of a partitioned CG iteration.  Visibility: the producers store their 8-byte partial write-through
// (sc1), drain it and arrive on a device-scope counter; the last block reads every partial with sc1 loads (MI355X guide, inter-workgroup
// recipe R1: sc1 stores AND sc1 loads need no fences); the interior's partials come from an earlier launch.
__device__ inline void ext_partial(const HArgs& k, real corr) {
    if (!k.norm_partial) return;
    __shared__ double red[2];
    __shared__ unsigned last;
    corr = wave_sum(corr);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = corr;
    __syncthreads();
    if (threadIdx.x == 0) {
        double* slot = k.norm_partial + k.partial_offset + blockIdx.y * gridDim.x + blockIdx.x;
        if (k.red_out) {
            // write-through (sc1) store of the 8-byte partial, drained, then the arrival: no release fence -- a fence per block would write back
            // every dirty line of the XCD's L2 (the whole output of this application) a few hundred times (measured: +37 us per launch)
            __hip_atomic_store(slot, red[0] + red[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned t = __hip_atomic_fetch_add(k.red_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = t == gridDim.x * gridDim.y - 1 ? 1u : 0u;
        } else *slot = red[0] + red[1];
    }
    if (!k.red_out) return;
    __syncthreads();
    if (!last) return;
    // the interior's partials were written by an earlier launch: plain loads, four independent chains in flight per thread; only the
    // exterior's own corrections (a few hundred) need the L1-bypassing loads
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int j = threadIdx.x;
    for (; j + 384 < k.partial_offset; j += 512) {
        s0 += k.norm_partial[j]; s1 += k.norm_partial[j + 128]; s2 += k.norm_partial[j + 256]; s3 += k.norm_partial[j + 384];
    }
    for (; j < k.partial_offset; j += 128) s0 += k.norm_partial[j];
    for (j = k.partial_offset + threadIdx.x; j < k.red_n; j += 128) s1 += __hip_atomic_load(k.norm_partial + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    double s = (s0 + s1) + (s2 + s3);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        k.red_out[0] = red[0] + red[1];
        __hip_atomic_store(k.red_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    // the requested number of sub-domains per t-slice, or the largest smaller multiple of 8 that divides the chunks of a slice (the local
    // volume 48 x 24 x 24 x 48 of an 8-way partitioned 48^3 x 96 has 216 = 8 x 27 chunks per slice: 16 becomes 8 and the map stays on)
    if (slice % TB == 0)
        while (k.nsub > 8 && (slice / TB) % k.nsub != 0) k.nsub -= 8;
    k.cps = (slice % TB == 0 && (slice / TB) % k.nsub == 0) ? slice / TB : 0;
    const int plane = c->geom.XH * c->geom.L[1];
    k.cpp = (plane % TB == 0) ? plane / TB : 0;
    k.ysplit = 1;
#ifdef LQCD_ABLATE
    k.dbg = c->tun.dbg;
#endif
    k.nt = (c->tun.nt_gauge & 3) | (c->tun.nt_store ? 4 : 0) | ((c->tun.nt_gauge & 4) ? 8 : 0);
    // (y,z) tiling of the sub-domains: the requested split, or the largest smaller one the geometry admits (48^3 x 96: 18 chunks per z-plane,
    // so 4 becomes 2 -- measured 1.11 -> 1.06 ms for the staggered kernel there, profiles/r02_staggered_map_sweep_48x96.log)
    for (int ys = c->tun.xcd_ysplit; ys > 1; ys--)
        if (k.cps > 0 && k.cpp > 0 && k.cpp % ys == 0 && k.nsub % ys == 0 && c->geom.L[2] % (k.nsub / ys) == 0) { k.ysplit = ys; break; }
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
    k.dotz[0] = (const real2*)s.dot_z[0]; k.dotz[1] = (const real2*)s.dot_z[1]; k.dot_partial = s.dot_partial; k.dot_conj = s.dot_conj;
    k.dotz2[0] = (const real2*)s.dot_z2[0]; k.dotz2[1] = (const real2*)s.dot_z2[1];
    k.fsel = s.fold >= 2 ? s.fold - 1 : 0;
    k.vlist = nullptr;
    return k;
}

static bool use_dirsplit(lqcd_ctx_s* c, int kind, real r) {   // variants 1/2/3 work on 64-site chunks
    if (!(c->tun.dslash_variant >= 1 && c->tun.dslash_variant <= 8)) return false;
    // Wilson: the split kernels use the r = 1 projectors.  On a partitioned lattice a general-r application runs as two r = 1 calls
    // (apply.hip, split_general_r), so the launch geometry (number of |.|^2 partials) is the r = 1 one there for every r.
    return kind == LQCD_STAGGERED || r == 1.0 || c->geom.part[0] || c->geom.part[1] || c->geom.part[2] || c->geom.part[3];
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

static PipeArgs make_pipe_args(lqcd_ctx_s* c, const KArgs& k, const StencilCall& s) {
    PipeArgs a;
    a.gauge = k.gauge12 ? k.gauge12 : k.gauge;
    const bool upd = k.upd_scal != nullptr;
    for (int p = 0; p < 2; p++) { a.dst[p] = upd ? k.upd[p] : k.out[p]; a.in[p] = k.in[p]; a.xin[p] = k.xin[p]; a.dotz[p] = k.dotz[p]; a.dotz2[p] = k.dotz2[p]; }
    a.dot_partial = k.dot_partial; a.dot_conj = k.dot_conj;
    a.clover = k.clover;
    a.norm_partial = k.norm_partial; a.upd_scal = k.upd_scal; a.skip = k.skip; a.scal_w = k.scal_w;
    a.a = k.a; a.b = k.b;
    a.nt_store = (k.nt & 4) != 0;
    a.nvirt = k.nblocks; a.both = s.parity_mode == 2; a.pmode = s.parity_mode == 2 ? 0 : s.parity_mode;
    a.XH = k.g.XH; a.L1 = k.g.L[1]; a.L2 = k.g.L[2]; a.LT = k.g.L[3]; a.nch = k.g.nch; a.dXH = k.g.dXH;
    for (int mu = 0; mu < 4; mu++) {
        a.sgn_f[mu] = k.g.part[mu] ? real(0.0) : real(k.g.bc_fwd[mu]);
        a.sgn_b[mu] = k.g.part[mu] ? real(0.0) : real(k.g.bc_bwd[mu]);
    }
    a.cps = k.cps; a.cpp = k.cpp; a.cpr = k.cpr; a.per_pass = std::max(1, k.cpr * k.g.L[3]); a.ty = k.ty; a.tz = k.tz; a.ysplit = k.ysplit;
    a.ctr = c->pipe_ctr; a.per_wg = 1;
    a.d_perpass = k.d_perpass; a.d_cpr = k.d_cpr; a.d_ysplit = k.d_ysplit; a.d_ty = k.d_ty; a.d_cpp = make_fastdiv(std::max(1, k.cpp));
    a.ls = s.dw_ls; a.slice_bytes = (unsigned long long)s.dw_slice * sizeof(real2); a.d_ls = make_fastdiv(std::max(1, s.dw_ls)); a.dw_mass = (real)s.dw_mass;
    a.fold = 0;
    a.fsel = s.fold >= 2 ? s.fold - 1 : 0;
    a.vlist = nullptr;
    for (int mu = 0; mu < 4; mu++) {
        // ghost buffers of this call's message size: [recv_bwd | recv_fwd] back to back (make_hargs' rule)
        const size_t cnt = (size_t)(s.parity_mode == 2 ? 2 : 1) * 6 * face_half_sites(c->geom, mu);
        a.gh_b[mu] = (const real2*)halo_recv_base(c, mu); a.gh_f[mu] = (const real2*)halo_recv_base(c, mu) + cnt;
        a.Fh[mu] = face_half_sites(c->geom, mu);
        a.gsf[mu] = (c->coord[mu] == c->pe[mu] - 1) ? real(c->geom.bc_fwd[mu]) : real(1.0);
        a.gsb[mu] = (c->coord[mu] == 0) ? real(c->geom.bc_bwd[mu]) : real(1.0);
        if (s.fold && k.g.part[mu]) a.fold |= 1 << mu;
    }
    return a;
}

static HArgs make_hargs(lqcd_ctx_s* c, const StencilCall& s);

// ---- bulk / boundary lists of the folded launches.  Which chunks have a site on a partitioned face is decided ON THE DEVICE, by the arithmetic the kernels use
// themselves (pipe_map of the scalar-addressing kernel; map_block_v + the ballot of fold_chunk_skipped for the twins), once per (kernel family, parity mode, map);
// the host only compacts the flags in the order of the map and keeps the two lists in the context.
__global__ void fold_classify_s(PipeArgs a, int n, unsigned char* bnd) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= n) return;
    int p_, t, z, yc;
    pipe_map(a, b, p_, t, z, yc);
    const int y0 = fdiv(yc * 64, a.dXH), y1 = fdiv(yc * 64 + 63, a.dXH);
    bnd[b] = ((((a.fold >> 3) & 1) && (t == 0 || t == a.LT - 1)) || (((a.fold >> 2) & 1) && (z == 0 || z == a.L2 - 1)) ||
              (((a.fold >> 1) & 1) && (y0 == 0 || y1 == a.L1 - 1))) ? 1 : 0;
}
__global__ __launch_bounds__(64) void fold_classify_k(KArgs k, unsigned char* bnd) {
    int chunk, p;
    map_block_v(k, (int)blockIdx.x, chunk, p);
    KArgs q = k;
    q.fsel = 1;      // "skipped by the bulk launch" = boundary
    const bool b = fold_chunk_skipped(q, chunk, p);
    if (threadIdx.x == 0) bnd[blockIdx.x] = b ? 1 : 0;
}
static int fold_lists_get(lqcd_ctx_s* c, int family, const StencilCall& s, const KArgs& k, const PipeArgs* a, const int** list, int* n) {
    const int which = s.fold == 2 ? 0 : 1;
    int mask = 0;
    for (int mu = 0; mu < 4; mu++) mask |= (c->geom.part[mu] ? 1 : 0) << mu;
    for (const FoldLists& f : c->fold_lists)
        if (f.family == family && f.parity_mode == s.parity_mode && f.nvirt == k.nblocks && f.mask == mask && f.remap == k.remap && f.nsub == k.nsub && f.ysplit == k.ysplit) {
            *list = f.d_list[which]; *n = f.n[which];
            return LQCD_OK;
        }
    const int nv = k.nblocks;
    unsigned char* d_flag = nullptr;
    HIPCHK(hipMalloc((void**)&d_flag, (size_t)nv));
    if (family == 0) hipLaunchKernelGGL(fold_classify_s, dim3((nv + 255) / 256), dim3(256), 0, c->stream, *a, nv, d_flag);
    else { KArgs q = k; q.vlist = nullptr; hipLaunchKernelGGL(fold_classify_k, dim3(nv), dim3(64), 0, c->stream, q, d_flag); }
    std::vector<unsigned char> flag((size_t)nv);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(flag.data(), d_flag, (size_t)nv, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_flag);
    if (e != hipSuccess) return hip_fail(e, "fold_lists_get", __FILE__, __LINE__);
    FoldLists f;
    f.family = family; f.parity_mode = s.parity_mode; f.nvirt = nv; f.mask = mask; f.remap = k.remap; f.nsub = k.nsub; f.ysplit = k.ysplit;
    std::vector<int> l[2];
    for (int b = 0; b < nv; b++) l[flag[(size_t)b] ? 1 : 0].push_back(b);
    for (int w = 0; w < 2; w++) {
        f.n[w] = (int)l[w].size();
        f.d_list[w] = nullptr;
        if (f.n[w]) {
            HIPCHK(hipMalloc((void**)&f.d_list[w], l[w].size() * sizeof(int)));
            HIPCHK(hipMemcpy(f.d_list[w], l[w].data(), l[w].size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    c->fold_lists.push_back(f);
    *list = f.d_list[which]; *n = f.n[which];
    return LQCD_OK;
}

// folded launch of the direction-split kernels (every case the scalar-addressing FOLD instances do not take)
static int launch_dirsplit_fold(lqcd_ctx_s* c, const StencilCall& s, const KArgs& k_all) {
    if (k_all.dot_partial || k_all.alpha_partials || s.dw_ls > 1 || (s.kind == LQCD_WILSON && s.r != 1.0)) {
        set_error("stencil: the folded launch has no dot / small-lattice / five-dimensional / general-r form");
        return LQCD_ERR_UNSUPPORTED;
    }
    HArgs h = make_hargs(c, s);
    KArgs k = k_all;
    int nb = k.nblocks;
    if (s.fold >= 2) {      // bulk / boundary launch: only the chunks of its kind
        LQCHK(fold_lists_get(c, 1, s, k_all, nullptr, &k.vlist, &nb));
        if (nb == 0) return LQCD_OK;
    }
    const dim3 grid(nb), block(256);
    if (s.kind == LQCD_STAGGERED) {
        if (k.gauge12) hipLaunchKernelGGL((staggered_dirsplit_fold<true>), grid, block, 0, c->stream, k, h);
        else hipLaunchKernelGGL((staggered_dirsplit_fold<false>), grid, block, 0, c->stream, k, h);
    } else {
#define LQ_DF(R, CL, CI) do { if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit_fold<true, R, CL, CI>), grid, block, 0, c->stream, k, h); \
                              else hipLaunchKernelGGL((wilson_dirsplit_fold<false, R, CL, CI>), grid, block, 0, c->stream, k, h); } while (0)
        if (s.clover_on_hop) { if (!k.clover) { set_error("stencil: clover-on-hop needs the packed blocks"); return LQCD_ERR_UNSUPPORTED; }
                               if (k.gauge12) LQ_DF(true, false, true); else LQ_DF(false, false, true); }
        else if (k.clover) { if (k.gauge12) LQ_DF(true, true, false); else LQ_DF(false, true, false); }
        else { if (k.gauge12) LQ_DF(true, false, false); else LQ_DF(false, false, false); }
#undef LQ_DF
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int launch_stencil_interior(lqcd_ctx_s* c, const StencilCall& s) {
    if (s.dw_ls > 1 && !stencil_dw5_applies(c, s)) { set_error("stencil: the five-dimensional launch does not apply to this call (domainwall.hip checks before asking)"); return LQCD_ERR_UNSUPPORTED; }
    if (use_dirsplit(c, s.kind, s.r)) {
        KArgs k = make_kargs(c, s, 64);
        const size_t pad = (size_t)c->tun.lds_pad_kb * 1024;
        // "12 + delta" links are read by the scalar-addressing Wilson kernel alone: every other launch takes the 18 stored reals
        const bool delta = s.gauge12_delta && !kF32Build && s.kind == LQCD_WILSON && k.gauge12 && !k.alpha_partials && !k.dot_partial && !s.clover_on_hop && !k.clover && !s.fold &&
                           c->tun.dslash_pipe == 2 && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false);
        if (s.gauge12_delta && !delta) { k.gauge12 = nullptr; c->tun.recon_active = 0; }
        if (s.fold) {      // folded halo schedule: the scalar-addressing FOLD instances where they exist, the folded twins of the direction-split kernels otherwise
            const bool sdir = !kF32Build && s.kind == LQCD_WILSON && s.r == 1.0 && !k.clover && !s.clover_on_hop && !k.dot_partial && !k.alpha_partials && s.dw_ls <= 1 &&
                              !k.g.part[0] && c->tun.dslash_pipe == 2 && (k.gauge12 || c->tun.dslash_s18) && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false);
            if (!sdir) return launch_dirsplit_fold(c, s, k);
        }
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
        } else if (s.kind == LQCD_WILSON && k.dot_partial) {      // dot mode (fused even-odd BiCGStab): a direction-split kernel with the inner-product epilogue
            if ((k.clover && !s.clover_on_hop) || s.r != 1.0) { set_error("stencil: dot mode needs the Wilson r = 1 kernel without a diagonal clover term"); return LQCD_ERR_UNSUPPORTED; }
            dim3 grid(k.nblocks), block(256);
            bool launched = false;
#ifndef LQCD_F32
            if (s.clover_on_hop && k.gauge12 && c->tun.clover_hop_s && c->tun.dslash_pipe == 2 && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false)) {
                // round 6: the scalar-addressing kernel with the inverse blocks in its epilogue (CINV instance)
                PipeArgs a = make_pipe_args(c, k, s);
                const bool ntb = (k.nt & 1) != 0;
                if (s.dagger) { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<true, true, true, true, false, false, false, true>), grid, block, 0, c->stream, a);
                                else hipLaunchKernelGGL((wilson_dirsplit_s<true, true, false, true, false, false, false, true>), grid, block, 0, c->stream, a); }
                else { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<false, true, true, true, false, false, false, true>), grid, block, 0, c->stream, a);
                       else hipLaunchKernelGGL((wilson_dirsplit_s<false, true, false, true, false, false, false, true>), grid, block, 0, c->stream, a); }
                launched = true;
            } else
#endif
            if (s.clover_on_hop) {      // even-odd clover solver: inverse blocks on the hop sum + the inner-product epilogue (both builds)
                if (k.gauge12) { if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, true, false, true, true>), grid, block, pad, c->stream, k);
                                 else hipLaunchKernelGGL((wilson_dirsplit<false, true, false, true, true>), grid, block, pad, c->stream, k); }
                else { if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, false, false, true, true>), grid, block, pad, c->stream, k);
                       else hipLaunchKernelGGL((wilson_dirsplit<false, false, false, true, true>), grid, block, pad, c->stream, k); }
                launched = true;
            }
#ifndef LQCD_F32
            else if (k.gauge12 && c->tun.dslash_pipe == 2 && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false)) {      // the scalar-addressing form
                PipeArgs a = make_pipe_args(c, k, s);
                const bool ntb = (k.nt & 1) != 0;
                if (s.dagger) { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<true, true, true, true>), grid, block, 0, c->stream, a);
                                else hipLaunchKernelGGL((wilson_dirsplit_s<true, true, false, true>), grid, block, 0, c->stream, a); }
                else { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<false, true, true, true>), grid, block, 0, c->stream, a);
                       else hipLaunchKernelGGL((wilson_dirsplit_s<false, true, false, true>), grid, block, 0, c->stream, a); }
                launched = true;
            }
#endif
            if (!launched) {            // the plain direction-split kernel (fp32 build: the inner chain of the mixed-precision even-odd solver)
                if (k.gauge12) {
                    if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, true, false, true>), grid, block, pad, c->stream, k);
                    else hipLaunchKernelGGL((wilson_dirsplit<false, true, false, true>), grid, block, pad, c->stream, k);
                } else {
                    if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, false, false, true>), grid, block, pad, c->stream, k);
                    else hipLaunchKernelGGL((wilson_dirsplit<false, false, false, true>), grid, block, pad, c->stream, k);
                }
            }
        } else if (s.kind == LQCD_WILSON && s.clover_on_hop) {      // even-odd clover solver: out = a xin + b C (H in), C = the inverse clover blocks of the output parity
            if (!k.clover || s.r != 1.0) { set_error("stencil: clover-on-hop needs the packed blocks and r = 1"); return LQCD_ERR_UNSUPPORTED; }
            dim3 grid(k.nblocks), block(256);
#ifndef LQCD_F32
            if (k.gauge12 && c->tun.clover_hop_s && c->tun.dslash_pipe == 2 && !k.alpha_partials && !s.fold && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false)) {
                PipeArgs a = make_pipe_args(c, k, s);      // round 6: the scalar-addressing kernel, inverse blocks in its epilogue
                const bool ntb = (k.nt & 1) != 0;
                if (s.dagger) { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<true, true, true, false, false, false, false, true>), grid, block, 0, c->stream, a);
                                else hipLaunchKernelGGL((wilson_dirsplit_s<true, true, false, false, false, false, false, true>), grid, block, 0, c->stream, a); }
                else { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<false, true, true, false, false, false, false, true>), grid, block, 0, c->stream, a);
                       else hipLaunchKernelGGL((wilson_dirsplit_s<false, true, false, false, false, false, false, true>), grid, block, 0, c->stream, a); }
            } else
#endif
            if (k.gauge12) { if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, true, false, false, true>), grid, block, pad, c->stream, k);
                             else hipLaunchKernelGGL((wilson_dirsplit<false, true, false, false, true>), grid, block, pad, c->stream, k); }
            else { if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit<true, false, false, false, true>), grid, block, pad, c->stream, k);
                   else hipLaunchKernelGGL((wilson_dirsplit<false, false, false, false, true>), grid, block, pad, c->stream, k); }
        } else if (s.kind == LQCD_WILSON && !k.alpha_partials && wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, k.clover != nullptr) &&
                   (c->tun.dslash_pipe != 2 || ((k.gauge12 || c->tun.dslash_s18) && !kF32Build))) {      // the scalar-addressing kernel: fp64 only (its one-site-per-lane fp32
                                                                                  // instance measured 62.5 vs 57 ms of variant 1 in the mixed CG, profiles/r03_mixed_precision.log); since
                                                                                  // round 4 also for the 18 stored reals (no spill once the diagonal term / old r load behind the hops)
            PipeArgs a = make_pipe_args(c, k, s);
            const bool persist = c->tun.dslash_pipe == 1 || c->tun.dslash_pipe == 3;
            if (c->tun.dslash_pipe == 3) { a.ctr = nullptr; a.per_wg = wilson_pipe_per_wg(c, k.nblocks); }
            const dim3 pg(c->tun.dslash_pipe == 1 ? wilson_pipe_grid(c, k.nblocks, s.prec) : c->tun.dslash_pipe == 3 ? k.nblocks / a.per_wg : k.nblocks), pb(256);
            const bool ntb = (k.nt & 1) != 0;
#ifndef LQCD_F32
            if (s.dw_ls > 1) {      // Domainwall: the L5 slices in one launch, fifth-direction hops in the epilogue (plain loads for the backward link: it is re-used by the next slice)
                if (c->tun.dslash_pipe != 2 || delta) { set_error("stencil: the five-dimensional launch needs the scalar-addressing kernel in its plain mode"); return LQCD_ERR_UNSUPPORTED; }
                const dim3 g5((unsigned)k.nblocks * (unsigned)s.dw_ls);
                if (!k.gauge12) { set_error("stencil: the five-dimensional launch reads the 12-real links"); return LQCD_ERR_UNSUPPORTED; }      // (its 18-real instance spills 154 VGPRs)
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit_s<true, true, false, false, false, true>), g5, pb, 0, c->stream, a);
                else hipLaunchKernelGGL((wilson_dirsplit_s<false, true, false, false, false, true>), g5, pb, 0, c->stream, a);
            } else
            if (s.fold) {      // partitioned lattice, exchange complete (apply.hip, folded one-stream schedule): boundary hops from the ghost buffers in this launch
                if (persist || delta || s.dw_ls > 1 || k.g.part[0]) { set_error("stencil: the folded launch needs the scalar-addressing kernel and an unpartitioned x direction"); return LQCD_ERR_UNSUPPORTED; }
                dim3 fg = pg;
                if (s.fold >= 2) {      // bulk / boundary launch: only the chunks of its kind, in the order of the map
                    int nb = 0;
                    LQCHK(fold_lists_get(c, 0, s, k, &a, &a.vlist, &nb));
                    if (nb == 0) return LQCD_OK;
                    fg = dim3(nb);
                }
#define LQ_FOLD(D, R) do { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<D, R, true, false, false, false, true>), fg, pb, 0, c->stream, a); \
                           else hipLaunchKernelGGL((wilson_dirsplit_s<D, R, false, false, false, false, true>), fg, pb, 0, c->stream, a); } while (0)
                if (k.gauge12) { if (s.dagger) LQ_FOLD(true, true); else LQ_FOLD(false, true); }
                else { if (s.dagger) LQ_FOLD(true, false); else LQ_FOLD(false, false); }
#undef LQ_FOLD
            } else
            if (delta) {       // rows 0, 1 + fp32 deviation of row 2 (reference-format configurations); plain loads for the backward link as well: the
                               // non-temporal form measured 0.3962 against 0.3915 ms at 32^3x64 (profiles/r04_links_12_plus_delta.log)
                if (s.dagger) hipLaunchKernelGGL((wilson_dirsplit_s<true, true, false, false, true>), pg, pb, 0, c->stream, a);
                else hipLaunchKernelGGL((wilson_dirsplit_s<false, true, false, false, true>), pg, pb, 0, c->stream, a);
            } else
#endif
            {
#define LQ_PIPE(D, R) do { if (persist) { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_pipe<D, R, true>), pg, pb, 0, c->stream, a); \
                                          else hipLaunchKernelGGL((wilson_dirsplit_pipe<D, R, false>), pg, pb, 0, c->stream, a); } \
                           else { if (ntb) hipLaunchKernelGGL((wilson_dirsplit_s<D, R, true>), pg, pb, 0, c->stream, a); \
                                  else hipLaunchKernelGGL((wilson_dirsplit_s<D, R, false>), pg, pb, 0, c->stream, a); } } while (0)
            if (k.gauge12) { if (s.dagger) LQ_PIPE(true, true); else LQ_PIPE(false, true); }
            else { if (s.dagger) LQ_PIPE(true, false); else LQ_PIPE(false, false); }
#undef LQ_PIPE
            }
#if !defined(LQCD_F32) && defined(LQCD_VARIANTS)   // opt-in variants 2-8 (stencil_alt.hip, -DLQCD_VARIANTS builds): fp64 only -- the fp32 build (paired-component fields) has the direction-split and the
                   // site-per-lane kernels, and the mixed-precision solvers pin dslash_variant to 0/1 for the duration of a solve (mixed.hip)
        } else if (c->tun.dslash_variant >= 2 && launch_wilson_alt(c, s, k, pad)) {
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
        // (peer-mapped backend: the "send buffers" are the neighbours' ghost buffers of the next exchange, comm.hip halo_send_base)
        h.send_fwd[mu] = (real2*)halo_send_base(c, mu, 0); h.send_bwd[mu] = (real2*)halo_send_base(c, mu, 1) + cnt;
        h.recv_bwd[mu] = (const real2*)halo_recv_base(c, mu); h.recv_fwd[mu] = (const real2*)halo_recv_base(c, mu) + cnt;
        h.sign_fwd[mu] = (c->coord[mu] == c->pe[mu] - 1) ? c->geom.bc_fwd[mu] : 1.0;
        h.sign_bwd[mu] = (c->coord[mu] == 0) ? c->geom.bc_bwd[mu] : 1.0;
    }
    h.norm_partial = s.norm_partial;
    h.partial_offset = stencil_num_blocks(c, s.kind, s.r, s.parity_mode, s.prec, s.clover != nullptr);   // corrections are appended to the interior's partials
    h.upd_scal = s.upd_scal;
    h.upd[0] = (real2*)s.upd[0]; h.upd[1] = (real2*)s.upd[1];
    h.red_out = (s.norm_partial && s.red_slot >= 0) ? c->d_scal + s.red_slot : nullptr;
    h.red_ctr = c->pipe_ctr + PIPE_CTR_RED_WORD;      // a word of its own next to the persistent kernel's exit counter
    h.red_n = h.partial_offset + ((max_face_threads(c, s.parity_mode) + 127) / 128) * 8;
    h.pack_next = s.pack_next;
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
    c->halo_epoch++;
    dim3 grid((nt + 127) / 128, 8), block(128);
    if (s.kind == LQCD_WILSON) hipLaunchKernelGGL(wilson_pack, grid, block, 0, c->stream, h);
    else hipLaunchKernelGGL(staggered_pack, grid, block, 0, c->stream, h);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

#ifndef LQCD_F32
// the pack launch of `s` (Wilson) + the sum of nblocks |.|^2 partials into d_scal[slot] (+ a CG scalar step), one launch (wilson_pack_reduce)
int launch_pack_reduce(lqcd_ctx_s* c, const StencilCall& s, const double* partial, int nblocks, int slot, int op) {
    const int nt = max_face_threads(c, s.parity_mode);
    HArgs h = make_hargs(c, s);
    c->halo_epoch++;
    dim3 grid(std::max(1, (nt + 255) / 256), 9), block(256);
    PeerRedArgs pr;
    if (c->has_comm && c->peer.on) pr = comm_red_args(c); else memset(&pr, 0, sizeof pr);
    hipLaunchKernelGGL(wilson_pack_reduce, grid, block, 0, c->stream, h, partial, nblocks, c->d_scal, slot, op, pr);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
#endif

int launch_stencil_exterior(lqcd_ctx_s* c, const StencilCall& s) {
    const int nt = max_face_threads(c, s.parity_mode);
    if (nt == 0) return LQCD_OK;
    HArgs h = make_hargs(c, s);
    if (s.kind != LQCD_WILSON) h.pack_next = -1;      // only the Wilson exterior packs for a following application
    if (h.pack_next >= 0) c->halo_epoch++;
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
int stencil_num_blocks(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec, bool clover) {
    if (prec == 2) return pair32_num_blocks(c);       // fp32 site-pair kernel: one workgroup per 64 pairs
    const int TB = use_dirsplit(c, kind, r) ? 64 : c->tun.dslash_block;
    const int nvirt = ((c->geom.Vh + TB - 1) / TB) * (parity_mode == 2 ? 2 : 1);
#ifdef LQCD_VARIANTS
    if (use_dirsplit(c, kind, r) && kind == LQCD_WILSON && c->tun.dslash_variant == 3) return persist_grid(c, nvirt);
#endif
    if (c->tun.dslash_pipe == 1 && wilson_pipe_applies(c, kind, r, parity_mode, clover)) return wilson_pipe_grid(c, nvirt, prec);
    if (c->tun.dslash_pipe == 3 && wilson_pipe_applies(c, kind, r, parity_mode, clover)) return nvirt / wilson_pipe_per_wg(c, nvirt);
    return nvirt;
}
// dslash_pipe = 3: consecutive virtual blocks per workgroup (the largest divisor of the block count not above pipe_chunks_per_wg)
int wilson_pipe_per_wg(lqcd_ctx_s* c, int nvirt) {
    int n = std::max(1, c->tun.pipe_chunks_per_wg);
    while (n > 1 && nvirt % n != 0) n--;
    return n;
}
// variant 9: persistent workgroups, 3 per CU in the fp64 build (151..168 VGPRs, 48 KiB of LDS), 5 in the fp32 build; a multiple of 8 so that
// virtual block b + k * grid stays on the XCD of block b.  prec = 1: the fp32 build (its callers ask stencil_num_partials with prec = 1 too).
int wilson_pipe_grid(lqcd_ctx_s* c, int nvirt, int prec) {
    int g = c->tun.pipe_grid > 0 ? c->tun.pipe_grid : c->num_cu * (c->tun.pipe_per_cu > 0 ? c->tun.pipe_per_cu : (prec ? 5 : 3));
    g -= g % 8;
    if (g < 8) g = 8;
    return std::min(g, nvirt);
}
// The persistent form of variant 1 (tunable dslash_pipe): only where a persistent workgroup has at least pipe_min_chunks chunks to walk (below that the
// tail imbalance eats the gain) and only the r = 1 Wilson operator; a call that carries the packed clover blocks (fused A x epilogue) keeps
// the plain variant-1 kernel -- `clover` is that property of the call (StencilCall::clover != nullptr; solvers: op_fused_clover).
// the L5 slices of a Domainwall application as ONE launch of the scalar-addressing kernel: fp64, one GPU, full-lattice plain mode, 12-real links (fields on the group)
bool stencil_dw5_applies(lqcd_ctx_s* c, const StencilCall& s) {
    if (kF32Build || s.prec != 0 || s.kind != LQCD_WILSON || s.r != 1.0 || s.parity_mode != 2 || any_partitioned(c)) return false;
    if (s.alpha_partials || s.dot_partial || s.clover || s.clover_on_hop || s.gauge12_delta) return false;      // (|.|^2 partials -- one per workgroup = chunk x slice -- and the update mode ride along: dw_solve)
    if (c->tun.dslash_pipe != 2 || !s.gauge12) return false;
    return wilson_pipe_applies(c, s.kind, s.r, s.parity_mode, false);
}
bool wilson_pipe_applies(lqcd_ctx_s* c, int kind, double r, int parity_mode, bool clover) {
    if (c->tun.dslash_variant != 1 || !c->tun.dslash_pipe || clover || kind != LQCD_WILSON || !use_dirsplit(c, kind, r)) return false;
    if (r != 1.0) return false;
    // t-slices and z-planes of whole chunks (t, z, y-chunk of a workgroup's sites are wave-uniform), the XCD tile map, 32-bit byte offsets
    // inside a parity block of the largest field (18-real links: 2304 B per site)
    const Geom& g = c->geom;
    const int plane = g.XH * g.L[1];
    if (plane % 64 != 0 || c->tun.xcd_remap != 2 || (size_t)g.nch * 64 * 2304 >= ((size_t)1 << 32)) return false;
    if ((plane * g.L[2] / 64) % 8 != 0) return false;      // the map needs the chunks of a t-slice to split over the 8 XCDs
    const int nvirt = ((g.Vh + 63) / 64) * (parity_mode == 2 ? 2 : 1);
    if (c->tun.dslash_pipe == 2 || c->tun.dslash_pipe == 3) return true;      // hardware dispatch order: no minimum size
    return nvirt >= std::max(1, c->tun.pipe_min_chunks) * wilson_pipe_grid(c, nvirt, 0);
}
// The folded one-stream halo schedule (apply.hip stencil_apply; tunable halo_fold): pack -> exchange -> ONE stencil launch that reads the ghost buffers itself.
// Where: the RCCL path with schedule 3 chosen; Wilson (r = 1 calls: plain, clover epilogue, inverse clover blocks on the hop sum) and staggered, fp64 and the fp32
// build.  The scalar-addressing Wilson kernel has its own FOLD instances (fp64, x unpartitioned, no clover term); everything else takes the folded twins of the
// direction-split kernels (wilson_dirsplit_fold / staggered_dirsplit_fold: the exterior kernel's arithmetic inside the direction waves).
bool halo_fold_applies(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec, bool clover) {
    if (!c->tun.halo_fold) return false;      // (every schedule has a folded form since round 6)
    if (!any_partitioned(c) || !c->has_comm || !c->local_peers.empty()) return false;
    if ((kind != LQCD_WILSON && kind != LQCD_STAGGERED) || (prec != 0 && prec != 1)) return false;
    // the direction-split kernels with one workgroup per chunk (default and scalar-addressing form): every instance has a folded twin.  Not the persistent forms
    // (dslash_pipe 1 / 3) and not the experiment variants.
    if (c->tun.dslash_variant != 1 || (c->tun.dslash_pipe != 0 && c->tun.dslash_pipe != 2)) return false;
    (void)r; (void)parity_mode; (void)clover;      // on a partitioned lattice a general-r application is two r = 1 calls (apply.hip split_general_r)
    return true;
}
// interior block partials + (partitioned lattice, unless the launch is folded) the exterior kernel's correction partials
int stencil_num_partials(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec, bool clover) {
    const int nt = halo_fold_applies(c, kind, r, parity_mode, prec, clover) ? 0 : max_face_threads(c, parity_mode);
    return stencil_num_blocks(c, kind, r, parity_mode, prec, clover) + (nt > 0 ? ((nt + 127) / 128) * 8 : 0);
}
#endif

}  // namespace lqcd
