// lqcd_internal.h -- shared internals of liblqcd_hip.so (gfx950 only; no CUDA-compat paths).
//
// Device data layout (HBM), chosen for coalesced 16-B/lane loads (one complex fp64 per lane, 1 KiB per
// wave64 instruction):
//   * sites are checkerboarded: parity p = (x+y+z+t)&1, cb = (x>>1) + XH*(y + LY*(z + LZ*t)), XH = LX/2
//   * chunk-blocked (default): spinor [parity][chunk = cb/64][comp][cb%64], gauge [parity][chunk][mu][a*3+b][cb%64]
//     (comp = spin*3 + colour (Wilson, 12) or colour (staggered, 3); element = one complex number, row a, column b of U_mu(n))
//   * an EVEN/ODD spinor is one parity block.
//
// Precision: the stencil translation unit is compiled twice -- fp64 (namespace lqcd::p64, the default everything else sees)
// and, with -DLQCD_F32, fp32 (lqcd::p32; same layouts with float2 elements) for the inner solver of the mixed-precision CG.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lqcd_hip.h"

namespace lqcd {

#ifdef LQCD_F32
typedef float real;
typedef float2 real2;
#define LQCD_PNS p32
inline namespace p32 {}
namespace p64 {}
#else
typedef double real;
typedef double2 real2;
#define LQCD_PNS p64
inline namespace p64 {}
namespace p32 {}
#endif

// exact division of 0 <= n < 2^31 by a run-time constant d >= 1 as mulhi + shift:  q = (n * M) >> (31 + s),
// s = ceil(log2 d), M = floor(2^(31+s) / d) + 1 (fits 32 bits; exact because n * (M*d - 2^(31+s)) < 2^(31+s) for n < 2^31).
// Two VALU ops instead of the ~25 of a generic u32 division -- they sit on the critical path in front of the first load of
// every stencil workgroup.  Verified exhaustively on the host for d <= 70000 (edges + strided n).
struct FastDiv { unsigned m; int sh; int d; };
inline FastDiv make_fastdiv(int d) {
    FastDiv f; f.d = d; f.m = 0; f.sh = 0;
    if (d == 1) return f;
    int s = 0;
    while ((1u << s) < (unsigned)d) s++;
    f.m = (unsigned)((((unsigned long long)1 << (31 + s)) / (unsigned)d) + 1);
    f.sh = s - 1;
    return f;
}
__host__ __device__ inline int fdiv(int n, const FastDiv& f) {
    if (f.d == 1) return n;
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)(__umulhi((unsigned)n, f.m) >> f.sh);
#else
    return (int)((((unsigned long long)(unsigned)n * f.m) >> 32) >> f.sh);
#endif
}

struct Geom {           // local sub-lattice geometry, passed by value to kernels
    int L[4];           // local extents
    int XH;             // L[0]/2
    FastDiv dXH, dL1, dL2;  // magic numbers for the divisions of cb_to_coords
    int Vh;             // sites per parity
    int Vs;             // component stride in sites: Vh + padding.  With Vh a power of two every component array would be a
                        // multiple of 256 KiB apart and the 21 loads of a hop alias to one L2 set / memory channel.
    int nch;            // number of 64-site chunks per parity = ceil(Vh / 64)
    int part[4];        // 1 if direction is partitioned over ranks (neighbour is off-rank)
    double bc_fwd[4];   // sign applied when a forward hop wraps locally across the GLOBAL boundary (unpartitioned dirs)
    double bc_bwd[4];
    int origin[4];      // global coordinates of local site (0,0,0,0)
    int gL[4];          // global extents
};

// checkerboard index -> local coordinates
__host__ __device__ inline void cb_to_coords(const Geom& g, int parity, int cb, int c[4]) {
    const int q0 = fdiv(cb, g.dXH);
    const int xh = cb - q0 * g.XH;
    const int q1 = fdiv(q0, g.dL1);
    c[1] = q0 - q1 * g.L[1];
    const int q2 = fdiv(q1, g.dL2);
    c[2] = q1 - q2 * g.L[2];
    c[3] = q2;
    c[0] = 2 * xh + ((c[1] + c[2] + c[3] + parity) & 1);
}
__host__ __device__ inline int coords_to_cb(const Geom& g, const int c[4]) {
    return (c[0] >> 1) + g.XH * (c[1] + g.L[1] * (c[2] + g.L[2] * c[3]));
}

// ---------------------------------------------------------------- gauge-field addressing
// LQCD_GAUGE_AOSOA = 1 (default): chunk-blocked layout  [parity][chunk = cb/64][mu][a*3+b][cb%64] -- the 36 link components a
// workgroup needs for 64 sites are ONE contiguous 36 KiB block (long DRAM bursts instead of 36 pieces of 1 KiB that are
// 16 MB apart); every wave load is still 64 lanes x 16 B = 1 KiB contiguous.
// LQCD_GAUGE_AOSOA = 0: plain structure-of-arrays [parity][mu][a*3+b][cb] with component stride Vs.
#ifndef LQCD_GAUGE_AOSOA
#define LQCD_GAUGE_AOSOA 1
#endif
__host__ __device__ inline size_t glink_off(const Geom& g, int p, int mu, int i) {   // element (a*3+b) = 0 of U_mu at (p, i)
#if LQCD_GAUGE_AOSOA == 2   // parities interleaved per chunk: [chunk][parity][mu][9][64]
    return ((((size_t)(i >> 6) * 2 + p) * 4 + mu) * 9) * 64 + (i & 63);
#elif LQCD_GAUGE_AOSOA
    return ((((size_t)p * g.nch + (size_t)(i >> 6)) * 4 + mu) * 9) * 64 + (i & 63);
#else
    return ((size_t)(p * 4 + mu) * 9) * g.Vs + i;
#endif
}
__host__ __device__ inline size_t glink12_off(const Geom& g, int p, int mu, int i) {   // 12-real copy: [parity][chunk][mu][6][64]
    return ((((size_t)p * g.nch + (size_t)(i >> 6)) * 4 + mu) * 6) * 64 + (i & 63);
}
__host__ __device__ inline size_t gauge12_elems(const Geom& g) { return (size_t)2 * g.nch * 24 * 64; }
__host__ __device__ inline int glink_stride(const Geom& g) {   // distance between consecutive components of one link
#if LQCD_GAUGE_AOSOA
    return 64;
#else
    return g.Vs;
#endif
}
__host__ __device__ inline size_t gauge_elems(const Geom& g) {
#if LQCD_GAUGE_AOSOA
    return (size_t)2 * g.nch * 36 * 64;
#else
    return (size_t)2 * 4 * 9 * g.Vs;
#endif
}

// ---------------------------------------------------------------- fermion-field addressing
// LQCD_SPINOR_AOSOA = 1 (default): [parity][chunk = cb/64][comp][cb%64] (12 KiB contiguous per chunk of a Wilson field);
// 0: [parity][comp][cb] with component stride Vs.  A parity block holds ncomp*Vs elements in both layouts, BLAS-1 is flat.
#ifndef LQCD_SPINOR_AOSOA
#define LQCD_SPINOR_AOSOA 1
#endif
__host__ __device__ inline size_t sp_off(int ncomp, int i) {   // offset of component 0 of site i inside a parity block
#if LQCD_SPINOR_AOSOA
    return (size_t)(i >> 6) * (size_t)(ncomp * 64) + (i & 63);
#else
    (void)ncomp;
    return (size_t)i;
#endif
}
__host__ __device__ inline int sp_stride(const Geom& g) {      // distance between consecutive components of one site
#if LQCD_SPINOR_AOSOA
    return 64;
#else
    return g.Vs;
#endif
}

// ---------------------------------------------------------------- faces (halo geometry)
// A face of direction mu holds the sites with x_mu fixed; per parity it has Fh(mu) = Vh / L[mu] sites.
// Face index f enumerates the remaining three coordinates in increasing direction order, the first of them halved
// (checkerboarded) -- the same rule as the bulk cb index.
__host__ __device__ inline int face_half_sites(const Geom& g, int mu) { return g.Vh / g.L[mu]; }
__host__ __device__ inline void face_to_coords(const Geom& g, int mu, int fixed, int parity, int f, int c[4]) {
    int d0 = (mu == 0) ? 1 : 0;
    int d1 = (mu <= 1) ? 2 : 1;
    int d2 = (mu <= 2) ? 3 : 2;
    int h0 = g.L[d0] / 2;
    int a0 = f % h0;
    int q = f / h0;
    c[d1] = q % g.L[d1];
    c[d2] = q / g.L[d1];
    c[mu] = fixed;
    c[d0] = 2 * a0 + ((c[d1] + c[d2] + fixed + parity) & 1);
}
__host__ __device__ inline int coords_to_face(const Geom& g, int mu, const int c[4]) {
    int d0 = (mu == 0) ? 1 : 0;
    int d1 = (mu <= 1) ? 2 : 1;
    int d2 = (mu <= 2) ? 3 : 2;
    return (c[d0] >> 1) + (g.L[d0] / 2) * (c[d1] + g.L[d1] * c[d2]);
}

// ---------------------------------------------------------------- complex helpers (precision of the translation unit)
inline namespace LQCD_PNS {
struct cd {
    real re, im;
};
__host__ __device__ inline cd mk(real a, real b) { cd r = {a, b}; return r; }
__host__ __device__ inline real2 mk2(real a, real b) { real2 t; t.x = a; t.y = b; return t; }
__device__ inline cd ld(const real2* p) { real2 v = *p; return mk(v.x, v.y); }
__device__ inline void st(real2* p, cd v) { real2 t; t.x = v.re; t.y = v.im; *p = t; }
__device__ inline cd operator+(cd a, cd b) { return mk(a.re + b.re, a.im + b.im); }
__device__ inline cd operator-(cd a, cd b) { return mk(a.re - b.re, a.im - b.im); }
__device__ inline cd operator*(real s, cd a) { return mk(s * a.re, s * a.im); }
__device__ inline cd cmul(cd a, cd b) { return mk(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
// acc += a*b
__device__ inline void cfma(cd& acc, cd a, cd b) {
    acc.re = fma(a.re, b.re, acc.re); acc.re = fma(-a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im); acc.im = fma(a.im, b.re, acc.im);
}
// acc += conj(a)*b
__device__ inline void cfma_conj(cd& acc, cd a, cd b) {
    acc.re = fma(a.re, b.re, acc.re); acc.re = fma(a.im, b.im, acc.re);
    acc.im = fma(a.re, b.im, acc.im); acc.im = fma(-a.im, b.re, acc.im);
}
// multiply by i^K
template <int K> __device__ inline cd mul_ipow(cd a) {
    constexpr int k = ((K % 4) + 4) % 4;
    if constexpr (k == 0) return a;
    else if constexpr (k == 1) return mk(-a.im, a.re);
    else if constexpr (k == 2) return mk(-a.re, -a.im);
    else return mk(a.im, -a.re);
}
}  // inline namespace

// gamma_mu (mu = 0,1,2) has one entry per row: row a -> column PERM[mu][a], value i^GK[mu][a]  (SURVEY.md Appendix A);
// gamma_4 = diag(1,1,-1,-1)
constexpr int PERM[3][4] = {{3, 2, 1, 0}, {3, 2, 1, 0}, {2, 3, 0, 1}};
constexpr int GK[3][4] = {{3, 3, 1, 1}, {2, 0, 0, 2}, {3, 1, 1, 3}};

// Sum over the 64 lanes of a wave, the same value in every lane, without the LDS crossbar: four DPP exchanges inside the 16-lane rows (quad_perm
// [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four row sums through v_readlane.  (A __shfl butterfly on doubles is a chain of
// dependent ds_bpermute round trips, ~0.3 us.)  All 64 lanes must be active.
template <int CTRL>
__device__ inline double dpp_get(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ inline double lane_get(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ inline double wave_sum(double v) {
    v += dpp_get<0xB1>(v);
    v += dpp_get<0x4E>(v);
    v += dpp_get<0x141>(v);
    v += dpp_get<0x140>(v);
    return (lane_get(v, 0) + lane_get(v, 16)) + (lane_get(v, 32) + lane_get(v, 48));
}

// Sum of n <= 1024 block partials that hold nvals interleaved values per block ([block][nvals]), value v: THE summation order of small reductions.
// Lane l adds partials l, l + 64, l + 128, ... in sequence (at most 16 independent loads in flight), one DPP wave sum at the end.  reduce_final
// (blas.hip) uses this function for n <= 1024, the consumers that fold a reduction into their prologue (cg_small, the fused BiCGStab chain) call it
// in every wave: the same bits everywhere.  All 64 lanes of the calling wave must be active.
__device__ inline double sum_partials_small_nv(const double* __restrict__ partial, int n, int nvals, int v, bool soa = false) {      // soa: [value][block] partials (coalesced)
    const int lane = threadIdx.x & 63;
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {      // all loads first (one memory round trip, not sixteen), then the additions in sequence; a missing partial is +0.0
        const int i = lane + 64 * k;
        t[k] = i < n ? partial[soa ? (size_t)v * n + i : (size_t)i * nvals + v] : 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += t[k];
    return wave_sum(s);
}
__device__ inline double sum_partials_small(const double* __restrict__ partial, int n) { return sum_partials_small_nv(partial, n, 1, 0); }

// ---------------------------------------------------------------- BiCGStab scalar steps (shared by the one-thread scalar kernels of blas.hip and the
// prologues of the fused chain in solvers.hip: the same expressions, so the two forms produce the same bits)
struct c2 { double re, im; };
__device__ inline c2 bicg_alpha(c2 rho, c2 r0v) {            // alpha = rho / <r0, v>
    const double d = r0v.re * r0v.re + r0v.im * r0v.im;
    c2 a;
    a.re = (rho.re * r0v.re + rho.im * r0v.im) / d;
    a.im = (rho.im * r0v.re - rho.re * r0v.im) / d;
    return a;
}
__device__ inline c2 bicg_omega(c2 ts, double tt, bool half) {   // omega = <t, s> / |t|^2; a half step (|s|^2 < eps) takes omega = 0: x += alpha p only, r = s
    c2 w;
    w.re = half ? 0.0 : ts.re / tt;
    w.im = half ? 0.0 : ts.im / tt;
    return w;
}
__device__ inline c2 bicg_beta(c2 rho1, c2 rho, c2 alpha, c2 omega) {   // beta = (rho' / rho) (alpha / omega)
    const double d0 = rho.re * rho.re + rho.im * rho.im;
    const double qr = (rho1.re * rho.re + rho1.im * rho.im) / d0, qi = (rho1.im * rho.re - rho1.re * rho.im) / d0;
    const double dw = omega.re * omega.re + omega.im * omega.im;
    const double er = (alpha.re * omega.re + alpha.im * omega.im) / dw, ei = (alpha.im * omega.re - alpha.re * omega.im) / dw;
    c2 b;
    b.re = qr * er - qi * ei;
    b.im = qr * ei + qi * er;
    return b;
}

// ---------------------------------------------------------------- the summation order of LARGE reductions (more than 1024 block partials)
// reduce_final (blas.hip, one block of FB = 1024 threads): thread c owns the class of partials c, c + FB, c + 2 FB, ... and sums it with four interleaved
// accumulators; the 64 class sums of a wave go through a __shfl_down tree, the 16 wave sums are added in sequence by thread 0.  The pieces live here because a
// second kernel reproduces that order bit for bit with 256 threads (stencil.hip wilson_pack_reduce: thread t owns classes t, t + 256, t + 512, t + 768).
constexpr int FB = 1024;
__device__ inline double sum_partials_class(const double* __restrict__ partial, int nblocks, int nvals, int v, int cls, bool soa = false) {
    // element (block i, value v): partial[i * nvals + v], or partial[v * nblocks + i] in the [value][workgroup] layout (same additions, coalesced reads)
    const size_t si = soa ? 1 : (size_t)nvals;
    partial += soa ? (size_t)v * nblocks : (size_t)v;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = cls;
    // sixteen, then eight loads in flight per thread before the first addition (one memory round trip for 16 K / 8 K partials instead of four / two); the additions
    // follow in the order of the plain loop below -- the same bits
    for (; i + 15 * FB < nblocks; i += 16 * FB) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; k++) t[k] = partial[(size_t)(i + k * FB) * si];
#pragma unroll
        for (int q = 0; q < 4; q++) { s0 += t[4 * q]; s1 += t[4 * q + 1]; s2 += t[4 * q + 2]; s3 += t[4 * q + 3]; }
    }
    for (; i + 7 * FB < nblocks; i += 8 * FB) {
        double t[8];
#pragma unroll
        for (int k = 0; k < 8; k++) t[k] = partial[(size_t)(i + k * FB) * si];
#pragma unroll
        for (int q = 0; q < 2; q++) { s0 += t[4 * q]; s1 += t[4 * q + 1]; s2 += t[4 * q + 2]; s3 += t[4 * q + 3]; }
    }
    for (; i + 3 * FB < nblocks; i += 4 * FB) {
        s0 += partial[(size_t)i * si];
        s1 += partial[(size_t)(i + FB) * si];
        s2 += partial[(size_t)(i + 2 * FB) * si];
        s3 += partial[(size_t)(i + 3 * FB) * si];
    }
    for (; i < nblocks; i += FB) s0 += partial[(size_t)i * si];
    return (s0 + s1) + (s2 + s3);
}
__device__ inline double shfl_tree_sum(double s) {      // lane 0 holds the sum of the wave's 64 values (the order every reduction of this library uses)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    return s;
}

// ---------------------------------------------------------------- the CG / BiCGStab scalar steps behind a reduction (one thread)
// slots of the device scalar block d_scal used by the solvers
enum { S_RED0 = 0, S_RR = 8, S_PQ = 9, S_ALPHA = 10, S_BETA = 11, S_DONE = 12, S_ITERS = 13, S_EPS = 14, S_RRNEW = 15, S_XDONE = 16, S_RROLD = 17, S_APREV = 18,
       S_AHIST = 64 };      // S_AHIST .. +7: the step lengths of the search directions whose x update is still pending (cg_defer_x >= 3: ring of p buffers)
// BiCGStab block (complex scalars are two consecutive doubles; B_TS..B_TT and B_RR..B_RHO1 are filled by one 3-value reduction)
enum { B_RHO = 24, B_R0V = 26, B_VV = 28, B_ALPHA = 29, B_SS = 31, B_TS = 32, B_TT = 34, B_OMEGA = 35, B_RR = 37, B_RHO1 = 38, B_BETA = 40,
       B_DONE = 42, B_ITERS = 43, B_EPS = 44, B_HALF = 45, B_RES = 46, B_RHOB = 47, B_TS5 = 49, B_UNSURE = 54, B_END = 55 };   // B_TS5: <t, s>, |t|^2, <r0, t> of the merged chain (bicg_fused = 4), one 5-value reduction;      // B_R0V..B_VV, B_TS..B_TT and B_RR..B_RHO1 are filled by one
                                                                                                     // 3-value reduction each; B_RHOB: second rho slot of the fused chain
__device__ inline void cg_scalar_step(double* s, int op) {
    if (op == 1) {
        if (s[S_DONE] != 0.0) { s[S_XDONE] = 1.0; return; }
        s[S_ALPHA] = s[S_RR] / s[S_PQ];
    } else if (op == 2) {
        if (s[S_DONE] != 0.0) return;
        const double rrn = s[S_RRNEW];
        s[S_BETA] = rrn / s[S_RR];
        s[S_RR] = rrn;
        s[S_ITERS] += 1.0;
        if (rrn < s[S_EPS]) s[S_DONE] = 1.0;
    } else if (op >= 3 && op <= 6) {
        // BiCGStab (ops.hip bicgstab_core): 3 alpha = rho/<r0,v> ; 4 half-step test on |s|^2 ; 5 omega = <t,s>/|t|^2 ;
        // 6 iters++, convergence / breakdown, beta = (rho'/rho)(alpha/omega), rho = rho'
        if (s[B_DONE] != 0.0) return;
        if (op == 3) {
            c2 rho = {s[B_RHO], s[B_RHO + 1]}, r0v = {s[B_R0V], s[B_R0V + 1]};
            const c2 a = bicg_alpha(rho, r0v);
            s[B_ALPHA] = a.re;
            s[B_ALPHA + 1] = a.im;
        } else if (op == 4) {
            s[B_HALF] = (s[B_SS] < s[B_EPS]) ? 1.0 : 0.0;
        } else if (op == 5) {
            c2 ts = {s[B_TS], s[B_TS + 1]};
            const c2 w = bicg_omega(ts, s[B_TT], s[B_HALF] != 0.0);     // half step: x += alpha p only, r = s
            s[B_OMEGA] = w.re;
            s[B_OMEGA + 1] = w.im;
        } else {
            s[B_ITERS] += 1.0;
            const double rr = (s[B_HALF] != 0.0) ? s[B_SS] : s[B_RR];
            s[B_RES] = rr;
            if (s[B_HALF] != 0.0 || rr < s[B_EPS]) { s[B_DONE] = 1.0; return; }
            if (!(fabs(rr) <= 1.79e308)) { s[B_DONE] = 2.0; return; }     // NaN / inf: breakdown
            c2 rho1 = {s[B_RHO1], s[B_RHO1 + 1]}, rho = {s[B_RHO], s[B_RHO + 1]}, al = {s[B_ALPHA], s[B_ALPHA + 1]}, om = {s[B_OMEGA], s[B_OMEGA + 1]};
            const c2 b = bicg_beta(rho1, rho, al, om);
            s[B_BETA] = b.re;
            s[B_BETA + 1] = b.im;
            s[B_RHO] = rho1.re; s[B_RHO + 1] = rho1.im;
        }
    }
}

// ---------------------------------------------------------------- counter-based RNG (identical bits on every rank / decomposition)
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline uint64_t rng_key(uint64_t seed, uint64_t site, uint64_t a, uint64_t b) {
    return splitmix64(splitmix64(splitmix64(seed ^ 0xA5A5A5A5DEADBEEFull) + site) + (a << 8) + b);
}
__host__ __device__ inline double u01(uint64_t k) { return ((k >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// ---------------------------------------------------------------- workgroup -> lattice map shared by the sweeps outside stencil.hip
// The XCD-aware tile sweep of the Dslash kernels (stencil.hip map_block_v, same arithmetic): block b runs on XCD b % 8 (observed, not
// contractual: the map only changes speed); XCD k owns (y,z) tiles k, k + 8, ... of every t-slice and sweeps t, the even and the odd half
// of a 64-site chunk are dispatched back to back.  Falls back to the plain order when the slice does not tile.
struct BlockMap {
    int nblocks, remap, cps, cpp, ysplit, cpr, ty, tz, LT;
    FastDiv d_perpass, d_cpr, d_ysplit, d_ty;
};
inline BlockMap make_block_map(const Geom& g, int remap, int nsub_in, int ysplit_in) {
    BlockMap m;
    const int TB = 64;
    const int chunks = (g.Vh + TB - 1) / TB;
    m.nblocks = 2 * chunks;
    m.remap = remap;
    m.LT = g.L[3];
    const int slice = g.XH * g.L[1] * g.L[2];
    int nsub = (nsub_in >= 8 && nsub_in % 8 == 0) ? nsub_in : 8;
    if (slice % TB == 0)
        while (nsub > 8 && (slice / TB) % nsub != 0) nsub -= 8;      // largest admissible multiple of 8 (stencil.hip make_kargs)
    m.cps = (slice % TB == 0 && (slice / TB) % nsub == 0) ? slice / TB : 0;
    const int plane = g.XH * g.L[1];
    m.cpp = (plane % TB == 0) ? plane / TB : 0;
    m.ysplit = 1;
    for (int ys = ysplit_in; ys > 1; ys--)      // the requested (y,z) split or the largest smaller one the geometry admits (stencil.hip make_kargs)
        if (m.cps > 0 && m.cpp > 0 && m.cpp % ys == 0 && nsub % ys == 0 && g.L[2] % (nsub / ys) == 0) { m.ysplit = ys; break; }
    m.cpr = m.cps > 0 ? m.cps / nsub : 1;
    m.ty = m.ysplit > 1 ? m.cpp / m.ysplit : 1;
    m.tz = m.cpr / m.ty;
    m.d_perpass = make_fastdiv(m.cpr * g.L[3] > 1 ? m.cpr * g.L[3] : 1);
    m.d_cpr = make_fastdiv(m.cpr > 1 ? m.cpr : 1);
    m.d_ysplit = make_fastdiv(m.ysplit > 1 ? m.ysplit : 1);
    m.d_ty = make_fastdiv(m.ty > 1 ? m.ty : 1);
    return m;
}
__host__ __device__ inline void block_map(const BlockMap& m, int b, int& chunk, int& p) {
    if (m.remap == 2 && m.cps > 0) {
        const int xcd = b & 7;
        int j = b >> 3;
        p = j & 1; j >>= 1;
        const int per_pass = m.cpr * m.LT;
        const int pass = fdiv(j, m.d_perpass);
        j -= pass * per_pass;
        const int t = fdiv(j, m.d_cpr), mm = j - t * m.cpr, sd = xcd + 8 * pass;
        int s;
        if (m.ysplit > 1) {
            const int sz = fdiv(sd, m.d_ysplit), sy = sd - sz * m.ysplit;
            const int zz = fdiv(mm, m.d_ty), yy = mm - zz * m.ty;
            s = (sz * m.tz + zz) * m.cpp + sy * m.ty + yy;
        } else {
            s = sd * m.cpr + mm;
        }
        chunk = t * m.cps + s;
        return;
    }
    int lb = b;
    if (m.remap && !(m.nblocks & 7)) lb = (b & 7) * (m.nblocks >> 3) + (b >> 3);
    chunk = lb >> 1; p = lb & 1;
}

// ---------------------------------------------------------------- runtime objects
struct Ctx;

struct Tunables {
    int dslash_block = 128;   // threads per workgroup of the stencil kernels
    int xcd_remap = 2;        // workgroup -> lattice map (stencil.hip map_block): 2 = per-XCD (y,z) tile swept through t
    int dslash_variant = 1;   // Wilson r=1 kernel: 0 site-per-lane, 1 dirsplit (4 waves/64 sites), 2 hopsplit (8 waves/64 sites)
    int nt_gauge = 1;         // split kernels: bit 0 = the BACKWARD use of a link (its second and last) is a non-temporal load, bit 1 = the forward use
                              // too.  Measured at 32^3x64 (profiles/r02_nt_sweep.log): bit 0 -1.5..2.5 %, bit 1 +40 % (the second use then misses
                              // the Infinity Cache as well)
    int nt_store = 1;         // output spinor stored non-temporally (it is not read again by this kernel: keeps its lines out of the L2) -1..3 %
    int cg_fused = 2;         // 0: reference form (c1 = p.q), 1: |Dp|^2 from the stencil, 2: + r-update fused into D^+, x/p updates merged
    int graph = 0;            // capture solver iterations in a hipGraph
    int persist_per_cu = 2;   // variant 3: resident workgroups per CU
    int dslash_pipe = 2;      // Wilson r = 1, variant 1, lattices whose z-planes are whole chunks: 1 = the persistent, software-pipelined form of the
                              // direction-split kernel (stencil.hip wilson_dirsplit_pipe; |.|^2 partials per persistent workgroup), 2 = variant 1's
                              // schedule on the persistent kernel's scalar addressing (wilson_dirsplit_s, default: 12-real links only, CG +1.6 % at
                              // 32^3 x 64, profiles/r03_pipe_ab.log); both bit-identical to variant 1; 0 = variant 1 itself
    int pipe_per_cu = 0;      // persistent kernel: resident workgroups per CU (0: what the
                              // kernel's registers / LDS admit -- 3 in the fp64 build, 5 in the fp32 build)
    int pipe_chunks_per_wg = 2;   // dslash_pipe = 3: consecutive virtual blocks one (non-persistent) workgroup walks with the pipelined loop
    int pipe_grid = 0;        // persistent kernel: explicit number of persistent workgroups (rounded down to a multiple of 8; tests), 0 = CUs x pipe_per_cu
    int pipe_min_chunks = 8;  // the persistent kernel runs only where every persistent workgroup has at least this many chunks to walk
#ifdef LQCD_ABLATE
    int dbg = 0;              // timing ablations (results are wrong when non-zero); -DLQCD_ABLATE builds only
#endif
    int xcd_ysplit = 4;       // remap 2: tile the sub-domains in (y,z) instead of plain z-slabs
    int xcd_nsub = 16;       // remap 2: sub-domains per t-slice (multiple of 8)
    int lds_pad_kb = 0;       // dynamic LDS added to the site-per-lane stencil launch (occupancy limiter, experiments)
    int halo_merge = 1;       // PE extent 2 in a direction: both faces travel to the same rank as ONE message each way
    int clover_fused = 1;     // Wilson-clover: apply A inside the direction-split kernel's epilogue (0: separate A x pass)
    int halo_stream_mode = -1; // partitioned stencil: 0 = exchange on the communication stream, interior on the compute stream;
                               // 1 = pack -> exchange -> exterior in order on the compute stream, interior on the second stream
                               // (no queue hop on the message path; pays when the exchange is the longer leg);
                               // 2 = interior enqueued first on the compute stream and in order with the exterior, pack -> exchange on the
                               // second stream (pays when the interior is the longer leg); 3 = everything in order on the compute stream (no overlap, no
                               // ~13 us cross-queue join: pays when the exchange is short); 4 (round 6) = one stream, the bulk of the folded stencil launch
                               // in FRONT of the exchange step and its boundary chunks behind it (peer-mapped backend: the faces fly while the bulk runs);
                               // -1 = time all five once, collectively
    int halo_tuned_us[5] = {0, 0, 0, 0, 0};   // per-application times the auto choice was made from (0: not tuned yet)
    int halo_inject_us = 0;    // TEST AID: every face exchange completes this many microseconds late (a timed one-wave spin behind the exchange on its stream) -- stands in for
                               // the flight time of real links on the one-GPU proxy, so that the schedules' crossover can be measured (profiles/r06_schedule_latency_table.log)
    int staggered_parity_solve = 1;  // lqcd_fermi_action / lqcd_calc_UdSfdU: a staggered pseudofermion with a zero odd half is solved with
                                     // the half-lattice CG of lqcd_solve_cg_DdagD_parity (0: always the full-lattice CG)
    int halo_fold = 1;        // the stencil launch reads the ghost buffers itself -- no exterior launch, no norm corrections; scalar-addressing Wilson kernel: its FOLD
                              // instances, everything else: the folded twins of the direction-split kernels (stencil.hip halo_fold_applies).  Schedule 3: one launch behind
                              // the exchange; schedules 0-2, 4 (round 6): a bulk launch beside / in front of the exchange + a boundary launch behind it.  0: interior + exterior kernels
    int halo_fold_active = 0; // read-only: the last partitioned stencil application ran folded (no exterior launch)
    int halo_fuse = 2;        // partitioned fused CG (Wilson, fp64): bit 0 = the exterior's last block does the final reduction (no reduce_final launch),
                              // bit 1 = the exterior of D p packs the faces D^+ needs and the x/p update packs the new p (no pack launches).
                              // One-GPU proxy at the N = 8 local volume (profiles/r03_halo_fuse_proxy.log): bit 1 +1.2 %, bit 0 -7 % (the last
                              // block's serial tail costs more than the 4 us launch it replaces) -> default 2
    int cg_fold_scalars = 1;  // several ranks: the scalar steps behind the two all-reduces of a CG iteration run in the consumers' prologues (no one-thread kernels)
    int nt_blas = 1;          // deferred-x CG update kernels stream their fields with non-temporal loads / stores: 842 -> 868 iter/s at 32^3x64
    int md_reunitarize = 1;   // lqcd_gauge_exp_update (U_update!) projects every updated link that is unitary to 1e-13 back onto SU(3) in the same pass
                              // (a field that was never on the group to that precision, e.g. a text-file configuration, is not touched): rounding alone carries
                              // max |row2 - conj(row0 x row1)| past the 12-real gate (1e-14) within ~280 link updates; 0 = the reference's literal update
    int staple_recon = 1;     // staple sweep on a field whose links are known to be on the group: rows 0, 1 are loaded, row 2 is rebuilt (2/3 of the L2 -> L1 bytes)
    int staple_tile = 1;      // staple sweep (single GPU, links on the group, chunks of whole x-rows): the links of both parities of a chunk sit in LDS and every neighbour link
                              // inside that (x, y) tile is read from there (md.hip gauge_force_kernel_tile); 0: every neighbour link through the L1 / L2 path
    int md_remap = 1;         // staple sweep: workgroups follow the Dslash kernels' XCD-aware tile sweep (0: plain chunk order)
    int cg_defer_x = 1;       // fused CG: 2 = x is updated every SECOND iteration with both search directions (x += a_k p_k + a_{k+1} p_{k+1}, p ping-pongs
                              // between two buffers): 9 instead of 10 spinor passes per iteration on average, identical iterates; K = 3..8 (round 5; 288 GB of HBM make
                              // the buffers free): a ring of K search-direction buffers, x += sum of K terms every K-th iteration -- (4 K + 1) / K update passes per
                              // iteration instead of 4.5 (K = 4: 4.25), the same iterates; 1 (default): 2 on an unpartitioned lattice, 8 on a partitioned one (measured:
                              // solvers.hip cg_ring_wanted); 0: x every iteration
    int cg_persist = 1;       // staggered CG on an unpartitioned lattice of <= 256 chunks: the whole solve is ONE launch (cg_persist.hip), two grid-wide
                              // synchronisations per iteration instead of three dependent launches; 0: the cg_small launch chain
    int cg_small = 1;         // fused CG on an unpartitioned lattice with <= 1024 stencil workgroups: the two reduction launches of an iteration are folded
                              // into the prologues of the kernels that consume them (3 dependent launches per iteration instead of 5)
    int cg_skip_done = 1;     // fused CG: the first Dslash of an iteration checks the convergence flag as well (0: only the second does)
    int clover_transport = 0; // 1: build the clover sums by the plaquette-transport passes also on an unpartitioned lattice (tests)
    int stag_both = 0;            // 1: staggered split kernel issues the loads of both hops of a direction back to back (unpartitioned lattices)
    int mixed_defer_x = 1;    // fp32 CG of the mixed-precision solver: x updated every second iteration with both search directions (solvers.hip cg_update_even/odd in fp32):
                              // 3 + 6 float streams per pair of iterations instead of 5 + 5, identical iterates
    int mixed_xfuse = 0;      // 1: fp32 site-pair solver forms x += alpha p in the epilogue of the update-mode D^+ (the update kernel then forms p only).  Bit-identical,
                              // measured SLOWER (50.4 vs 48.4 ms at 32^3x64: the six extra loads sit behind the barrier of a kernel that is at its register limit): off
    int mixed_pair32 = 1;         // mixed-precision solvers, plain Wilson r = 1 on an unpartitioned lattice with 12-real links: the fp32 inner operator is the
                                  // site-pair kernel (stencil_pair32.hip: two sites per lane, packed fp32 arithmetic); 0 = the one-site-per-lane fp32 build
    int mixed_action_solver = 0;  // 1: lqcd_fermi_action / lqcd_calc_UdSfdU solve with the mixed-precision CG (true-residual stopping rule);
                                  // 2: the same, and the rational actions solve all poles with the mixed-precision multi-shift CG
    int gauge_recon = 12;     // 12 (default): the direction-split kernels read 2 rows per link and rebuild the third -- only while every
                              // link of the field is unitary to 1e-14 (checked per gauge version), otherwise the 18 stored reals are
                              // read; bytes/site 960 -> 768 (Wilson), 672 -> 480 (staggered).  18: always read all 18 reals.
#ifdef LQCD_VARIANTS
    int variants_built = 1;   // read-only: the library carries the opt-in Wilson kernel variants 2-8 (stencil_alt.hip, -DLQCD_VARIANTS)
#else
    int variants_built = 0;   // read-only: dslash_variant >= 2 runs variant 1 (built without -DLQCD_VARIANTS)
#endif
    int bicg_fused = 4;       // even-odd BiCGStab, plain Wilson r = 1 on an unpartitioned lattice: 4 [default, round 6] = 2 + the x / r update and the p update as ONE launch WITHOUT a
                              // barrier: rho' and |r'|^2 from inner products that exist before r' does (solvers.hip bicgf_xrp_rec; <r0, t> from a second inner product in the dot
                              // epilogue of every dot-mode kernel, Wilson-clover included) -- 6 launches, 12 vector passes instead of 14: 112.4 -> 106.0 us per iteration at 16^3x32, 20.9 -> 19.9 ms per
                              // solve at 32^3x64; equal to form 2 up to the rounding of the two recurrences (no drift: rho - alpha <r0, v> = <r0, s> = 0 in every iteration); Wilson-clover 16^3x32: 186 -> 174 us per iteration.  3 [opt-in, round 6] = 2 + the x / r update and the p update as ONE launch
                              // with a grid-wide barrier between them (6 launches per iteration; all <= 1024 workgroups resident) -- bit-identical and SLOWER (128.6 vs
                              // 112.4 us per iteration at 16^3x32, profiles/r06_bicgstab_eo_chain.log: a barrier of 1024 workgroups costs more than the launch boundary it replaces); 1 = the inner products come from the epilogues of the Schur
                              // operator's second hop (no dot-product passes), reductions and scalar steps as separate one-block launches; 2 = on lattices of
                              // <= 1024 chunks per parity the reductions and scalar steps also move into the prologues of the consumers (7 dependent launches per
                              // iteration instead of 17, identical iterates); 0 = the generic chain (what the clover / full-lattice solvers run)
    int clover_hop_s = 1;     // even-odd Wilson-clover solver: the hops with the inverse clover blocks on the hop sum run the scalar-addressing kernel (CINV instance of
                              // wilson_dirsplit_s) where it applies; 0: the plain direction-split kernel
    int bicg_xrp_active = 0;  // read-only: the last even-odd BiCGStab solve ran the fused x / r / p launch -- 1: bicg_fused = 3 (grid barrier, every workgroup resident), 2: bicg_fused = 4 (recurrences)
    int action_eo_solver = 1; // lqcd_fermi_action / lqcd_calc_UdSfdU, Wilson(-clover): X = (D^+D)^-1 eta through two even-odd BiCGStab solves (Y = D^-+ eta, X = D^-1 Y)
                              // instead of the CG on the normal equations (0: the reference's form); same stopping rule for the same residual (actions.hip)
    int lazy_links = 0;       // 1: the per-direction call triples of the reference's U_update! / P_update! (lqcd_link_exp -> lqcd_link_mul -> lqcd_link_copy,
                              // lqcd_link_staple -> lqcd_link_mul -> lqcd_link_add_ta) are recorded and run as ONE fused launch each, four completed triples of one
                              // update as one four-direction launch (md.hip "lazy link triples"); 0: every call launches its own kernel.  The temporaries of a
                              // completed fused triple are never written, so the plain C ABI is EAGER by default (every field holds what the call says it does);
                              // the Julia and Python bindings, whose callers hand the temporaries back to their pool unread, switch it on when they create a
                              // context
    int dw_batched = 1;       // Domainwall operator: the L5 slices of an application as one launch of the scalar-addressing Wilson kernel with the fifth-direction hops in its
                              // epilogue (where that kernel applies: fp64, one GPU, z-planes of whole chunks); 0: L5 Wilson launches + one fifth-direction pass
    int dw_fused_cg = 1;      // Domainwall solves: 1 = the fused CG iteration of the four-dimensional operators on the five-dimensional launch (|D p|^2 partials in D's epilogue,
                              // r -= alpha D^+ t in D^+'s, x / p update in one pass: 5 vector passes and 5 launches per iteration instead of 11 and 11); 0: cg_generic
    int dw_active = 0;        // read-only: the last Domainwall application ran as one five-dimensional launch
    int lazy_merge = 2;       // (with lazy_links) a complete link update U <- exp(a P) U waits; the next one of the same U, P with nothing in between that reads U or
                              // writes P adds its step: exp(b P) exp(a P) = exp((a + b) P), one pass instead of two (the back-to-back half steps of
                              // runMD_QPQ_sw!, standardMD.jl:146-166: 11 link passes per MD step instead of 20); lqcd_gauge_exp_update takes part.  2 (default; one GPU): a complete
                              // momentum update P_update! waits as well, and runs with the link update that follows it as ONE sweep (staple_force_expu: the new
                              // links go to a second buffer that changes places with the field's).  0: every complete update is launched at once
    int mixed_lean_residual = 0;  // test aid: the fused true residual of the mixed even-odd BiCGStab never writes r0 / p / x (what it does behind a step that was expected to be the last)
    int bicg_dot_soa = 1;         // dot partials of the even-odd chains as [value][workgroup]: 1 = where the reductions are separate launches (more than 1024 workgroups: reduce_final
                                  // reads them coalesced, 14.9 -> 8.4 us), 2 = also where the consumers' prologues sum them, 0 = never
    int bicg_rec_guard = 6;       // bicg_fused = 4: digits of cancellation the recurrence |r'|^2 = |s|^2 - |<t,s>|^2 / |t|^2 may show before the stopping test waits for the summed |r'|^2
                                  // (one kernel later); 0 makes every test wait (tests)
    int bicg_reliable = 0;        // mixed-precision even-odd BiCGStab, 1: behind a correction step the fp32 chain goes on with its search direction, r0 and scalars (the true
                                  // residual replaces the recursive one: a reliable update) instead of starting again from p = r.  Measured at 32^3 x 64 on a hot configuration
                                  // (profiles/r06_links16_reliable.log): no fewer iterations -- kappa 0.141: 14 / 13-14, 0.19: 28 / 28, 0.22: 57 / 62 -- restarted BiCGStab loses
                                  // nothing on these systems, so the default stays 0
    int mixed_links16 = 1;        // site-pair inner operator of the mixed-precision solvers: links as int16 fixed point, value = n / 32767 (8 bytes per pair element instead of 16;
                                  // stencil_pair32.hip ldx16).  The inner operator then differs from the outer one by 1.2e-5 (measured, relative to the result), so a correction
                                  // step gains four digits instead of six.  1: in the even-odd BiCGStab (the action / force solves), where the step count stays three at
                                  // eps = 1e-16 and a solve takes 11.4 instead of 13.0 ms at 32^3 x 64; 2: in every mixed-precision solver (the CG on D^+D loses: 49.5 vs
                                  // 46.5 ms, a fourth restart); 0: fp32 links
    int pair32_active = 0;    // read-only: the last mixed-precision solve / lqcd_op_apply_f32 ran the fp32 site-pair kernel
    int recon_active = 0;     // read-only: 1 if the last operator application used the 12-real links, 2: rows 0, 1 + the fp32 deviation of row 2 ("12 + delta")
    int bicg_mixed = 0;       // 1: the even-odd BiCGStab of the plain Wilson operator (lqcd_solve_bicgstab_eo, and through it the action / force solves) runs the fp32 chain
                              // inside an fp64 defect correction; the stopping rule holds for the true fp64 residual.  mixed_action_solver = 1 switches it on for the action solves
    int dslash_s18 = 1;       // dslash_pipe = 2: the scalar-addressing kernel also for the 18 stored reals (round 4: its instance no longer spills at 3 waves per SIMD --
                              // the diagonal term and the old r of the update mode are requested behind the hops); 0: the plain direction-split kernel there
    int gauge_delta = 1;      // fields that fail the 12-real gate but lie within 1e-9 of the group (reference-format configurations) take the "12 + delta" links in the
                              // scalar-addressing Wilson kernel: 896 B/site moved instead of 960, results equal to the 18-real kernel's to fp64 rounding; 0: always 18 reals
};

// Recorded single-direction link operations of one context (md.hip "lazy link triples").  A record holds raw handles: every entry point that
// reads, writes or destroys a gauge-shaped field runs the records first (links_flush), so none outlives a field it names.
struct LinkRef {
    lqcd_gauge_s* g = nullptr;
    int mu = 0;
    bool is(const lqcd_gauge_s* h, int m) const { return g == h && mu == m; }
};
struct LazyLinks {
    int kind = 0;               // the open triple: 0 none, 1 exp recorded, 2 exp + mul, 3 staple recorded, 4 staple + mul
    LinkRef E, P, W, U, S, T;   // expU, p[mu], W, U[mu] of the U_update! triple; dSdUmu, temp1 of the P_update! triple
    lqcd_gauge_s* Ug = nullptr; // the link field of the staple
    int mu = 0;
    double t = 0.0, beta = 0.0;
    struct Done {               // a completed triple, deferred once more: kind 1 F[slot] <- exp(a G[slot]) F[slot]; kind 2 F[slot] += a TA(G[slot] (b/2) staples)
        int kind;
        lqcd_gauge_s* F;
        int slot;
        double a;
        lqcd_gauge_s* G;
        double b;
    };
    std::vector<Done> done;
    // a complete four-direction link update F <- exp(a G) F that has not been launched yet: the next one of the same fields, with nothing that reads the
    // links or writes the momenta in between, adds its step to it (runMD_QPQ_sw!'s back-to-back half steps, standardMD.jl:146-166)
    bool has_pend = false;
    Done pend = {0, nullptr, 0, 0.0, nullptr, 0.0};
    // ... and a complete momentum update F += a TA(-(b/6) G staples) in front of it: the two run as one sweep (lazy_merge = 2, md.hip staple_force_expu)
    bool has_pp = false;
    Done pp = {0, nullptr, 0, 0.0, nullptr, 0.0};
    bool busy() const { return kind != 0 || has_pend || has_pp || !done.empty(); }
};
constexpr int PIPE_CTR_WORDS = 10 * 32;          // the context's counter block (pipe_ctr): 8 per-XCD queue heads + 1 exit counter of the persistent stencil
constexpr int PIPE_CTR_RED_WORD = 8 * 32 + 16;   // kernel, 128 B apart; two more words of the exit counter's line: the exterior's reduction ticket (stencil.hip) ...
constexpr int PIPE_CTR_NOTPROJ_WORD = 8 * 32 + 24;   // ... and the "some link was not projected" flag of the link updates (md.hip)




// ---------------------------------------------------------------- peer-mapped communication backend (comm.hip, round 6)
// Beside RCCL: every rank owns ONE device allocation (its "window": ghost buffers of the stencil, mailboxes of the rarer face exchanges, flag words, slots of the
// scalar reductions), exports it with hipIpcGetMemHandle and maps the windows of the other ranks.  Producers store faces STRAIGHT into the neighbour's ghost buffer;
// ordering is by 64-bit sequence numbers in flag words, raised and awaited by one-wave kernels in stream order (no large kernel ever spins, so several ranks can
// share one device -- the world-size-2 tests on one GPU -- without deadlock); interprocess HIP events cannot be waited on across processes on this stack
// (profiles/r06_ipc_probe.log: hipStreamWaitEvent -> invalid argument).  The layout is a pure function of the local geometry, identical on every rank.
constexpr int PEER_MAX_RANKS = 8;       // one node of MI355X
constexpr int PEER_RED_RING = 4;        // reductions in flight before a slot is reused (ranks are never more than one reduction apart: 2 would do)
constexpr int PEER_RED_VALS = 8;        // values per reduction
constexpr size_t PEER_FLAG_STRIDE = 128;    // bytes between flag words (each on a line of its own)
struct PeerLayout {
    size_t halo_flag = 0;   // [4 mu][2 side] u64, side 0: written by my -mu neighbour (its forward face has landed), side 1: by my +mu neighbour
    size_t aux_flag = 0;    // [4][2] u64: number of the last mailbox message that has landed in box (mu, side)
    size_t aux_ack = 0;     // [4][2] u64: number of the last message I sent in direction (mu, dirn) that its receiver has copied out of the box
    size_t red_flag = 0;    // [ring][PEER_MAX_RANKS] u64
    size_t red_val = 0;     // [ring][PEER_MAX_RANKS][PEER_RED_VALS] double
    size_t ghost[4] = {};   // partitioned directions: 2 buffers (exchange number & 1) of [from -mu | from +mu]
    size_t ghost_bytes[4] = {};     // bytes of ONE buffer
    size_t aux_box[4] = {}; // partitioned directions: [side 0 | side 1], aux_cap bytes each
    size_t aux_cap[4] = {};
    size_t total = 0;
};
struct PeerComm {
    bool on = false;                // this context communicates through mapped windows (has_comm is set as well)
    bool exported = false;
    int finegrained = 1;            // the window is fine-grained device memory (remote stores bypass the L2: required between devices; 0 only for one-device experiments)
    char* win[PEER_MAX_RANKS] = {}; // mapped base of every rank's window (own rank: the allocation itself)
    bool opened[PEER_MAX_RANKS] = {};   // mapped by hipIpcOpenMemHandle (closed at teardown)
    PeerLayout lay;
    uint64_t xchg_seq = 0;          // stencil halo exchanges issued so far: producers write buffer xchg_seq & 1, consumers read buffer (xchg_seq - 1) & 1
    uint64_t red_seq = 0;           // reductions issued so far
    uint64_t aux_sent[4][2] = {}, aux_rcvd[4][2] = {};
    unsigned* status = nullptr;     // pinned host words, written by a wait that gave up: [0] what (1 halo, 2 reduction, 3 mailbox data, 4 mailbox ack), [1] index, [2..3] the number awaited
    int timeout_ms = 60000;
};
// bulk / boundary lists of the folded stencil launches (stencil.hip fold_lists_get): virtual blocks in the order of the workgroup map, kept per context
struct FoldLists {
    int family, parity_mode, nvirt, mask, remap, nsub, ysplit;      // what the lists were made for (family 0: scalar-addressing kernel, 1: the folded twins)
    int* d_list[2];                                                 // 0: bulk (no site on a partitioned face), 1: boundary
    int n[2];
};
// argument block of an exchange step: flag words to raise / to wait for (peer_signal_wait_wave below); nsig == nwt == 0: nothing to do
struct PeerSyncArgs {
    unsigned long long* sig[PEER_MAX_RANKS];     // flag words in the neighbours' windows that this rank raises
    unsigned long long* wt[PEER_MAX_RANKS];      // flag words in this rank's window it waits for
    int nsig, nwt;
    unsigned long long sig_seq[PEER_MAX_RANKS], wt_seq[PEER_MAX_RANKS];
    unsigned long long limit;
    unsigned* status;
    unsigned what;
    int fence;      // 1: system-scope release / acquire fences around the flag operations (the stand-alone exchange step: the L2 is clean behind the producer's kernel end,
                    // so the write-back costs nothing there); 0: only completion of the wave's own stores (the mailbox acknowledgements, which order no data)
};
// argument block of a reduction over the ranks inside a kernel (peer_allreduce_wave below); nranks == 0: not a peer context / nothing to do
struct PeerRedArgs {
    double* val[PEER_MAX_RANKS];                // rank r's slot block of this reduction: [source rank][PEER_RED_VALS]
    unsigned long long* flag[PEER_MAX_RANKS];   // rank r's flag words of this reduction: [source rank]
    int nranks, rank;
    unsigned long long seq, limit;              // limit: ticks of the 100 MHz clock a wait may take
    unsigned* status;
    int fence;                                  // as in PeerSyncArgs
};

#ifdef __HIPCC__
// entry j of an 8-pointer table held in kernel arguments, without a dynamic index (which would send the whole argument struct through scratch)
template <typename T>
__device__ __forceinline__ T* pick8(T* const (&t)[PEER_MAX_RANKS], int j) {
    T* p = t[0];
    p = j == 1 ? t[1] : p; p = j == 2 ? t[2] : p; p = j == 3 ? t[3] : p; p = j == 4 ? t[4] : p;
    p = j == 5 ? t[5] : p; p = j == 6 ? t[6] : p; p = j == 7 ? t[7] : p;
    return p;
}
__device__ __forceinline__ void peer_give_up(unsigned* status, unsigned what, unsigned idx, unsigned long long awaited) {
    if (status && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        status[1] = idx; status[2] = (unsigned)awaited; status[3] = (unsigned)(awaited >> 32);
        __hip_atomic_store(status, what, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__device__ __forceinline__ unsigned long long pick8v(const unsigned long long (&t)[PEER_MAX_RANKS], int j) {
    unsigned long long p = t[0];
    p = j == 1 ? t[1] : p; p = j == 2 ? t[2] : p; p = j == 3 ? t[3] : p; p = j == 4 ? t[4] : p;
    p = j == 5 ? t[5] : p; p = j == 6 ? t[6] : p; p = j == 7 ? t[7] : p;
    return p;
}
// Ordering of earlier stores before a flag store that follows.  fence = 1: a system-scope release fence -- the stand-alone exchange step runs behind the kernel end of
// the producer, where the L2 holds nothing dirty, so the write-back it implies is cheap.  INSIDE a producer launch it is not: an in-kernel exchange step (a ticket among
// the pack blocks, then fence + flag) measured 60 us per launch (the whole L2 is written back), and without the fence the faces stored through an IPC mapping were still
// in the writer's L2 when the other process read them (profiles/r06_peer_insync_ab.log) -- the step stays a launch of its own.  fence = 0: completion of this wave's own
// stores (vmcnt counts stores on gfx9-family parts), enough where every word involved is written by system-scope atomics, which bypass the caches.
__device__ __forceinline__ void peer_release(int fence) {
    if (fence) __threadfence_system();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// The exchange step, by one full wave.  Lane j < nsig: everything this stream has stored so far is visible before flag
// sig[j] rises to sig_seq[j]; lane j < nwt: wait until flag wt[j] has reached wt_seq[j] -- the launches behind this one find the faces the neighbours stored
// before THEY raised the flags (their kernel-start acquire; fence = 1 adds an acquire fence here).  A wait that outlives a.limit records itself in a.status and
// returns (comm_check turns it into LQCD_ERR_COMM).  Flag words are only ever touched by system-scope atomics, which bypass the caches.
__device__ inline void peer_signal_wait_wave(const PeerSyncArgs& a) {
    const int j = threadIdx.x & 63;
    peer_release(a.fence);
    if (j < a.nsig) __hip_atomic_store(pick8(a.sig, j), pick8v(a.sig_seq, j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (j < a.nwt) {
        const unsigned long long* w = pick8(a.wt, j);
        const unsigned long long want = pick8v(a.wt_seq, j);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > a.limit) { peer_give_up(a.status, a.what, (unsigned)j, want); break; }
        }
    }
    if (a.fence) __threadfence_system();
}
// Sum over the ranks of n <= PEER_RED_VALS doubles, by ONE full wave (all 64 lanes active, wave-uniform n): lane r + 8 k passes this rank's value k in `mine`
// (the same in all r) and gets the sum of value k over the ranks back.  Every rank stores its values into slot [its rank] of every rank's window, raises the
// slot's flag to the reduction's number, waits for the nranks flags of its own window and adds the nranks slots IN RANK ORDER -- every rank forms the same
// bits, so convergence decisions taken from the sum agree.  A wait that outlives a.limit ticks records itself in a.status (host: comm_check) and returns.
__device__ inline double peer_allreduce_wave(const PeerRedArgs& a, double mine, int n) {
    const int l = threadIdx.x & 63, r = l & 7, k = l >> 3;
    const bool act = r < a.nranks && k < n;
    if (act) __hip_atomic_store(pick8(a.val, r) + a.rank * PEER_RED_VALS + k, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    peer_release(a.fence);       // the values (of every lane of this wave) before the flags
    if (act && k == 0) __hip_atomic_store(pick8(a.flag, r) + a.rank, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* myflag = pick8(a.flag, a.rank);
    const double* myval = pick8(a.val, a.rank);
    bool ok = !(r < a.nranks && k == 0);
    const unsigned long long t0 = wall_clock64();
    while (!__all(ok)) {
        if (!ok) ok = __hip_atomic_load(myflag + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= a.seq;
        if (!ok) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > a.limit) { peer_give_up(a.status, 2u, (unsigned)r, a.seq); ok = true; }
        }
    }
    if (a.fence) __threadfence_system();
    else asm volatile("" ::: "memory");      // (the value loads below are issued after the flag loads have returned: the loop inspected them)
    const double x = act ? __hip_atomic_load(myval + r * PEER_RED_VALS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
    double s = __shfl(x, 8 * k, 64);
    for (int q = 1; q < a.nranks; q++) s += __shfl(x, 8 * k + q, 64);
    return s;
}
#endif

}  // namespace lqcd

struct lqcd_ctx_s {
    int device = 0;
    int gL[4], pe[4], rank = 0, nranks = 1, coord[4];
    int nbr_fwd[4], nbr_bwd[4];
    lqcd::Geom geom;
    hipStream_t stream = nullptr, comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_comm = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    hipEvent_t ev_tune0 = nullptr, ev_tune1 = nullptr;   // the halo-schedule tuner's own timing events (ev_t0 / ev_t1 belong to the bench entry points)
    // reductions
    bool zero_guess_hint = false; // set by a caller that has just cleared the solution field (actions.hip): the even-odd BiCGStab skips M x0 and the |x0|^2 test
    double* d_partial = nullptr;  // [MAX_PARTIAL_BLOCKS * 4]
    double* d_scal = nullptr;     // small device scalar block (solver state)
    double* h_scal = nullptr;     // pinned mirror
    uint64_t halo_epoch = 0;      // bumped by everything that writes the halo send buffers: a producer's pre-packed faces are valid only while it is unchanged
    unsigned* cgp_ctr = nullptr;  // one-launch CG (cg_persist.hip): 8 barrier counters 128 B apart, monotonic across launches
    unsigned cgp_epoch = 0;       // barriers those counters have counted; cgp_nwg: for which grid (-1: unknown -> zero them first)
    int cgp_nwg = -1;
    unsigned* pipe_ctr = nullptr; // persistent stencil kernel: 8 per-XCD queue heads + 1 exit counter, 128 B apart; all zero between launches
    // halo buffers (sized for Wilson full-lattice: 2 parities * 6 comps * Fh)
    double2* send_fwd[4] = {}, *send_bwd[4] = {}, *recv_fwd[4] = {}, *recv_bwd[4] = {};
    size_t halo_elems[4] = {};
    // fermion-force halos (force.hip): full X and Y spinors of the lower face, allocated on first use
    double2* force_send[4] = {}, *force_recv[4] = {};
    int force_ncomp = 0;
    // staple-force halos (md.hip): forward ghost links (+ their send buffer) and the lower-staple faces, allocated on first use
    double2* gf_ghost[4] = {}, *gf_gsend[4] = {}, *gf_wsend[4] = {}, *gf_wrecv[4] = {};
    lqcd_gauge_s* stout_tmp[2] = {};    // W = U A and the N matrices of the stout back-propagation (md.hip), created on first use
    double2* gauge_spare = nullptr;     // second link buffer of the fused momentum + link update (md.hip staple_force_expu), allocated on first use
    double2* clover_q[2] = {};          // clover sums / transport ping-pong, six 3x3 matrices per site (clover.hip)
    double2* clover_ext = nullptr;      // halo-extended links + Lambda matrices of the partitioned clover force, and its face buffers
    size_t clover_ext_bytes = 0;
    double2* clover_ext_buf[2] = {};
    size_t clover_ext_buf_bytes[2] = {};
    // fp32 work space of the mixed-precision solver (mixed.hip): links + 4 spinors, allocated on first use
    const void* mix_gauge_of = nullptr;  // gauge handle / version the fp32 link copies in mix_buf[0], mix_buf[5] were made from
    uint64_t mix_gauge_version = 0;
    bool mix_gauge18_valid = false;      // the 18-real fp32 copy (mix_buf[0]) was made for that version (the site-pair kernel does not read it: skipped there)
    bool mix_gauge12_valid = false;      // the 12-real fp32 copy (mix_buf[5]) was made for that version
    int mix_gauge12_layout = 0;          // ... in which layout: 1 component pairs (stencil.hip fp32 build), 2 site pairs (stencil_pair32.hip)
    void* mix_buf[10] = {};    // 9: int16 link copy (mixed_links16)
    bool mix_gauge16_valid = false;
        // 7: fp32 x_j / p_j pool of the mixed-precision multi-shift solver; 8: second search-direction buffer of the fp32 CG
    size_t mix_bytes[9] = {};
    ncclComm_t comm = nullptr;      // halo send/recv (communication stream)
    ncclComm_t comm_red = nullptr;  // reductions and other collectives issued on the compute stream
    bool has_comm = false;
    lqcd::PeerComm peer;            // the peer-mapped backend (comm.hip); peer.on: it, not RCCL, carries this context's exchanges
    std::vector<lqcd::FoldLists> fold_lists;      // bulk / boundary block lists of the folded launches (stencil.hip)
    std::vector<lqcd_ctx_s*> local_peers;  // in-process emulation of the PE grid
    // scratch spinors owned by the context (Temporalfields analogue)
    std::vector<lqcd_spinor_s*> scratch;
    lqcd::Tunables tun;
    void* cg_session = nullptr;   // open timing session of lqcd_cg_session_* (ops.hip), if any
    int num_cu = 256;
    lqcd::LazyLinks lazy;         // recorded single-direction link operations (md.hip)
    bool has_waiting_pack = false;   // folded halo schedule: the pack launch for the next application waits for the reduction it shares a launch with (apply.hip, blas.hip)
    void* waiting_pack = nullptr;    // its StencilCall (owned)
    // destroy calls from another thread than the context's own (a garbage collector's finalizer thread): a gauge-shaped field that recorded link
    // operations may name is not flushed and freed there -- it is parked (under capi.hip's g_live_mu) and the context's thread frees it at its next field
    // creation / lqcd_ctx_sync / lqcd_ctx_destroy.  home_thread: the creating thread, or the one that last set the parameter "adopt_thread".
    std::thread::id home_thread;
    std::vector<lqcd_gauge_s*> parked_gauges;
};

struct lqcd_gauge_s {
    lqcd_ctx_s* ctx;
    double2* data;  // [parity][chunk][mu][9][64]
    size_t elems;
    uint64_t version = 1;        // bumped by every entry point that writes the field
    // optional 12-real copy (rows 0 and 1 of every link, [parity][chunk][mu][6][64]) for the opt-in compressed Dslash
    double2* data12 = nullptr;
    uint64_t version12 = 0;      // version of `data` the copy was made from
    bool recon_ok = false;       // row 2 == conj(row 0 x row 1) to 1e-14 on every link of that version
    double recon_dev = 0.0;      // max |row 2 - conj(row 0 x row 1)| measured when the copy was made
    // "12 + delta" copy for fields that are close to the group but not on it (the reference's text / ILDG configurations: 8.8e-11): rows 0, 1 in fp64 and
    // delta = row 2 - conj(row 0 x row 1) in fp32, [parity][chunk][mu][8][64]; built on demand when the 12-real gate fails, used by the scalar-addressing
    // Wilson kernel only while max |delta| <= 1e-9 (its fp32 rounding is then below 1e-16)
    double2* data12d = nullptr;
    uint64_t version12d = 0;
    bool delta_ok = false;
    uint64_t unitary_version = 0;  // version of `data` whose links are all known to be on the group to rounding (generated on it, measured by the
                                   // 12-real pass, or projected by the link update): kernels may then rebuild row 2 instead of loading it
};

struct lqcd_spinor_s {
    lqcd_ctx_s* ctx;
    int kind;    // LQCD_WILSON | LQCD_STAGGERED
    int subset;  // LQCD_FULL | LQCD_EVEN | LQCD_ODD
    int ncomp;   // 12 | 3
    double2* data;
    size_t elems;  // ncomp * Vh * (1|2) [* ls]
    bool in_use = false;  // scratch-pool flag
    int ls = 1;           // LQCD_DOMAINWALL: extent of the fifth direction, ls Wilson fields in one allocation (domainwall.hip)
    bool owner = true;    // false: a slice view of a five-dimensional field (lqcd_spinor_slice) -- destroy frees the handle only
    lqcd_spinor_s* view_of = nullptr;   // a slice view: its five-dimensional field ...
    int nviews = 0;                     // ... which counts its live views and, destroyed while some exist (finalizers run in any order), keeps its storage until the last is gone
    bool zombie = false;
};

struct lqcd_op_s {
    lqcd_ctx_s* ctx;
    int kind;
    lqcd_gauge_s* gauge;
    double km;  // kappa or mass
    double r;
    int bc[4];
    // clover term (clover.hip): csw != 0 turns D into D_sw = D + (A - 1); A is rebuilt lazily when the links change
    double csw = 0.0;
    double2* clover = nullptr;          // packed chiral blocks, [parity][chunk][36][64]
    uint64_t clover_version = 0;        // gauge version A was built from
    lqcd_spinor_s* clover_tmp = nullptr;   // A x, the diagonal input of the stencil
    double2* clover_inv = nullptr;      // A^-1 in the same packed format (even-odd solver), built on first use
    uint64_t clover_inv_version = 0;
    double2* clover_lambda = nullptr;   // six Hermitian 3x3 matrices per site: scratch of the clover force
    int bicg32_hint[2][4] = {};         // mixed-precision chain: iterations the last solve's correction steps took, per step, for D and D^+ apart (the action solves alternate) (mixed.hip)
    int bicg_hint = 0;                  // iterations the last even-odd BiCGStab solve with this operator took (polling schedule of the next one)
    // LQCD_DOMAINWALL (domainwall.hip): km = the fermion mass m; the 4-D Wilson operator of the slices (hop coefficient 1/2), five-dimensional work fields
    int L5 = 0;
    double dw_M = 0.0;
    lqcd_op_s* dw_wilson = nullptr;
    lqcd_spinor_s* dw_work[8] = {};
    double* dw_partial = nullptr;       // |.|^2 partials of the five-dimensional launch: one per workgroup = chunk x slice (domainwall.hip dw_solve)
    size_t dw_partial_n = 0;
};

namespace lqcd {

constexpr int MAX_PARTIAL_BLOCKS = 4096;
constexpr int SCAL_DOUBLES = 96;

void set_error(const std::string& msg);
int hip_fail(hipError_t e, const char* what, const char* file, int line);
int nccl_fail(ncclResult_t e, const char* what, const char* file, int line);

#define HIPCHK(x)                                                              \
    do {                                                                       \
        hipError_t _e = (x);                                                   \
        if (_e != hipSuccess) return lqcd::hip_fail(_e, #x, __FILE__, __LINE__); \
    } while (0)
#define NCCLCHK(x)                                                              \
    do {                                                                        \
        ncclResult_t _e = (x);                                                  \
        if (_e != ncclSuccess) return lqcd::nccl_fail(_e, #x, __FILE__, __LINE__); \
    } while (0)
#define LQCHK(x)                       \
    do {                               \
        int _s = (x);                  \
        if (_s != LQCD_OK) return _s;  \
    } while (0)
#define ARGCHK(cond, msg)                     \
    do {                                      \
        if (!(cond)) {                        \
            lqcd::set_error(msg);             \
            return LQCD_ERR_ARG;              \
        }                                     \
    } while (0)


// comm.hip: the two communication backends behind one set of calls.  RCCL (lqcd_ctx_comm_init): grouped ncclSend / ncclRecv, ncclAllReduce.  Peer-mapped windows
// (lqcd_ctx_peer_export / lqcd_ctx_peer_init): direct stores + flag words, see PeerComm above.
struct CommXfer {           // one leg of a face exchange: send `bytes` to the neighbour in direction (mu, dirn) and receive as much from the opposite neighbour
    const void* send;
    void* recv;
    size_t bytes;
    int mu;
    int dirn;               // 0: the message travels forward (to nbr_fwd[mu], from nbr_bwd[mu]); 1: backward
};
int comm_sendrecv(lqcd_ctx_s* c, const CommXfer* x, int n, hipStream_t stream, bool halo_comm);     // halo_comm: RCCL uses the halo communicator (comm stream) instead of comm_red
int comm_allreduce(lqcd_ctx_s* c, double* d_inout, int n, int cg_op = 0);        // in place on device doubles, on the compute stream; cg_op: the CG scalar step behind it (peer: same launch)
int comm_halo_exchange(lqcd_ctx_s* c, int kind, int parity_mode, int prec, int where);     // the stencil's face exchange (where: see apply.hip)
PeerRedArgs comm_red_args(lqcd_ctx_s* c);       // peer backend: the argument block of the NEXT reduction (counts it); otherwise nranks = 0
int comm_inject_stamp(lqcd_ctx_s* c, hipStream_t s);      // test aid (tunable halo_inject_us): the faces are sent HERE in stream order ...
int comm_inject_delay(lqcd_ctx_s* c, hipStream_t s);      // ... and arrive halo_inject_us later
int comm_check(lqcd_ctx_s* c);                  // peer backend: LQCD_ERR_COMM if a wait gave up since the last check (call behind a stream synchronisation)
void comm_teardown(lqcd_ctx_s* c);              // peer backend: unmap / free the windows (lqcd_ctx_destroy)
// where this context's producers store the faces of the next exchange / its consumers find the ghosts of the last one (bases of [fwd | bwd], [from bwd | from fwd])
double2* halo_send_base(lqcd_ctx_s* c, int mu, int toward_bwd);
const double2* halo_recv_base(lqcd_ctx_s* c, int mu);

// ---- kernel launchers implemented across the .hip files
// stencil: out = a * xin + b * Hop(in), for the parities selected by `parity_mode` (0, 1, or 2 = both).
struct StencilCall {
    int kind;
    const double2* gauge;         // [2][4][9][Vh]
    double2* out[2];              // per parity block (nullptr if not written)
    const double2* in[2];         // per parity block of the INPUT field (hop reads in[1-p])
    const double2* xin[2];        // per parity block of the diagonal term (may be nullptr when a == 0)
    double a, b;
    double r;                     // Wilson parameter
    int dagger;
    int parity_mode;              // 0 even out, 1 odd out, 2 both
    double* norm_partial = nullptr;  // if non-null: per-block partial sums of |out|^2 (or |r|^2 in update mode) are written here
    // update mode (CG): instead of storing v = a*xin + b*Hop(in), do  r <- r - alpha*v  with alpha read from the device
    // scalar block (upd_scal[S_ALPHA]); the kernel is a no-op once upd_scal[S_DONE] is set.  q = D^+ D p is never written.
    const double* upd_scal = nullptr;
    double2* upd[2] = {nullptr, nullptr};
    const double* skip_flag = nullptr;  // device scalar block: the interior launch is a no-op once skip_flag[S_DONE] is set (iterations
                                        // enqueued behind the converging one in a burst)
    // small lattices (cg_small): no separate reduction launches.  The update-mode kernel sums the <= 1024 block partials of the previous
    // kernel itself (every wave, in reduce_final's order: identical iterates) and forms alpha = rr / pq in its prologue.
    const double* alpha_partials = nullptr;
    int alpha_n = 0;
    double* scal_w = nullptr;     // the device scalar block, writable: block 0 records pq, alpha and the rr this iteration started from
    const double2* gauge12 = nullptr;  // compressed links (fp64 build, Wilson r = 1 split kernel) or nullptr
    const void* gauge16 = nullptr;     // prec == 2 only: the int16 fixed-point pair copy of the 12-real links (stencil_pair32.hip ldx16); the kernel then reads it instead of gauge12
    int gauge12_delta = 0;        // 1: gauge12 is the 8-word "12 + delta" copy (lqcd_gauge_s::data12d); only the scalar-addressing Wilson kernel reads it, every
                                  // other launch of the call falls back to the 18 stored reals
    const double2* clover = nullptr;    // packed clover blocks: the Wilson split kernel (variant 1) applies A to xin in its epilogue
    int prec = 0;                 // 0: fp64 fields, 1: fp32 fields (pointers are float2 data, see p32)
    // partitioned lattices, fused tails of the exterior launch (stencil.hip ext_partial / wilson_pack_site):
    int red_slot = -1;            // >= 0: the exterior's last block sums all |.|^2 partials of this application into d_scal[red_slot]
    int pack_next = -1;           // 0 / 1: the exterior also packs the faces of `out` for a following application with this dagger flag
    double2* xacc[2] = {nullptr, nullptr};      // update mode, fp32 site-pair kernel only: x += alpha p in the same epilogue (x = xacc, p = pacc; parity blocks)
    const double2* pacc[2] = {nullptr, nullptr};
    int prepacked = 0;            // 1: the send buffers already hold this application's faces (packed by its producer): no pack launch
    // dot mode (fused BiCGStab chain, Wilson r = 1 direction-split kernel, fp64, unpartitioned, no clover): the epilogue also forms, per workgroup,
    // dot_partial[3 b + (0,1,2)] = Re <z, out>, Im <z, out>, |out|^2 (<a, b> = sum conj(a) b) with z = dot_z (parity blocks like out)
    const double2* dot_z[2] = {nullptr, nullptr};
    double* dot_partial = nullptr;
    int dot_conj = 0;             // bit 0: <out, z> (the imaginary part changes sign); bit 1: the partials go out as [value][workgroup] instead of [workgroup][value] -- what the
                                  // one-block reduction of MORE than 1024 workgroups reads coalesced (reduce_to_slot, soa = true); the folded prologues read [workgroup][value]
    const double2* dot_z2[2] = {nullptr, nullptr};      // every dot-mode kernel, z = xin only: a second inner product <z2, out> (never conjugated) -> FIVE values per workgroup,
                                                          // dot_partial[5 b + (0..4)] = Re / Im <z, out>, |out|^2, Re / Im <z2, out>  (merged BiCGStab chain: <r0, t> beside <t, s>)
    // Domainwall (domainwall.hip): the L5 slices of a five-dimensional field in ONE launch of the scalar-addressing kernel -- in / xin / out point at slice 0, slice s5 is
    // dw_slice elements further on -- with the fifth-direction hops -P_A psi(s+1) - P_B psi(s-1) (mass term at the walls) added in the epilogue
    int dw_ls = 0;
    size_t dw_slice = 0;
    double dw_mass = 0.0;
    int clover_on_hop = 0;        // 1 (fp64 direction-split kernel, r = 1): `clover` holds packed blocks that are applied to the HOP SUM, out = a xin + b C (H in) -- the
                                  // inverse clover blocks of the even-odd Wilson-clover solver; the diagonal term stays plain
    int defer_pack = 0;           // folded schedule, pack_next >= 0: the caller sums this application's |.|^2 partials next (reduce_pack_to_slot) -- the pack launch for the
                                  // following application waits in the context and runs as ONE launch with that reduction
    int fold = 0;                 // set by stencil_apply (folded halo schedules): 1 = the exchange is complete when the launch starts and it takes the boundary hops from the
                                  // ghost buffers itself -- no exterior launch, complete |.|^2 partials; 2 / 3 (round 6, overlapping schedules) = the same kernel on the
                                  // chunks WITHOUT / WITH a site on a partitioned face: "bulk" runs beside the exchange, "boundary" after arrival
};
// stencil.hip, once per precision (p64 is the inline namespace everywhere except in the fp32 build of stencil.hip).
// With prec = 1 the field pointers of a StencilCall address float2 data (cast), scalars stay double.
#ifdef LQCD_F32      // reopen each namespace the way it was declared at the top of this header
#define LQCD_REOPEN_P64 namespace p64
#define LQCD_REOPEN_P32 inline namespace p32
#else
#define LQCD_REOPEN_P64 inline namespace p64
#define LQCD_REOPEN_P32 namespace p32
#endif
LQCD_REOPEN_P64 {
int launch_stencil_interior(lqcd_ctx_s* c, const StencilCall& s);
int launch_stencil_pack(lqcd_ctx_s* c, const StencilCall& s);
int launch_stencil_exterior(lqcd_ctx_s* c, const StencilCall& s);
}
LQCD_REOPEN_P32 {
int launch_stencil_interior(lqcd_ctx_s* c, const StencilCall& s);
int launch_stencil_pack(lqcd_ctx_s* c, const StencilCall& s);
int launch_stencil_exterior(lqcd_ctx_s* c, const StencilCall& s);
}
// force.hip: pack the lower-face X, Y spinors of every partitioned direction / exchange them / the outer-product sweep
int launch_force_pack(lqcd_ctx_s* c, int kind, lqcd_spinor_s* X, lqcd_spinor_s* Y);
int force_halo_exchange_rccl(lqcd_ctx_s* c, int kind);
int force_halo_exchange_local_all(lqcd_ctx_s** ctxs, int n, int kind);
int launch_fermion_force(lqcd_ctx_s* c, int kind, const lqcd_gauge_s* U, lqcd_gauge_s* out, lqcd_spinor_s* X, lqcd_spinor_s* Y, double km,
                         double r, double scale = 1.0, int accumulate = 0);
// stencil_pair32.hip: fp32 Wilson Dslash on site pairs (StencilCall::prec == 2) and the conversions of its field layout
bool pair32_geometry_ok(lqcd_ctx_s* c);
int pair32_num_blocks(lqcd_ctx_s* c);
int pair32_cvt_spinor(lqcd_ctx_s* c, float2* dst, const double2* src, double scale, int npar = 2, float2* dst2 = nullptr, float2* dst3 = nullptr, float2* xzero = nullptr);      // npar = 1: one parity block
int pair32_axpy_to_f64(lqcd_ctx_s* c, double2* y, const float2* x, double a, int npar = 2);
int pair32_cvt_gauge12(lqcd_ctx_s* c, float2* dst, const double2* src12);
int pair32_residual(lqcd_ctx_s* c, float2* dst, const double2* rhs, const double2* q, double scale, float2* dst2, float2* dst3, float2* xzero, int* nb);
int pair32_cvt_gauge16(lqcd_ctx_s* c, void* dst, const double2* src12);
int launch_pair32_interior(lqcd_ctx_s* c, const StencilCall& s);
int stencil_apply(lqcd_ctx_s* c, const StencilCall& s);  // full sequence incl. halo exchange (RCCL path); s.prec selects the build
int halo_exchange_rccl(lqcd_ctx_s* c, int kind, int parity_mode, int prec, int where);
int make_full_call(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, StencilCall& s);
int flush_waiting_pack(lqcd_ctx_s* c);       // apply.hip
int halo_schedule_settle(lqcd_op_s* op);   // partitioned context with halo_stream_mode = -1: run the one-off schedule timing now (apply.hip)
int op_refresh_clover(lqcd_op_s* op);   // rebuilds A when the links moved; clover_version follows only a successful build
void apply_bc(lqcd_ctx_s* c, const int bc[4]);
int op_apply_async(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, double* norm_partial, const double* skip_flag = nullptr);
int cg_run(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double eps, int maxiter, bool fixed, int* iters, double* final_rr);
bool any_partitioned(lqcd_ctx_s* c);
bool halo_fold_applies(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec, bool clover);   // the folded one-stream schedule runs for such a call (stencil.hip)
int halo_exchange_local_all(lqcd_ctx_s** ctxs, int n, int kind, int parity_mode);
int stencil_num_blocks(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec = 0, bool clover = false);
bool stencil_dw5_applies(lqcd_ctx_s* c, const StencilCall& s);      // a StencilCall with dw_ls > 1 can run (stencil.hip)
bool wilson_pipe_applies(lqcd_ctx_s* c, int kind, double r, int parity_mode, bool clover);   // the persistent kernel runs for this call (large lattices only)
// the operator's full-lattice applications carry the packed clover blocks into the stencil (make_full_call's rule)
inline bool op_fused_clover(const lqcd_op_s* op) {
    return op->csw != 0.0 && op->clover && op->clover_tmp && op->r == 1.0 && op->ctx->tun.dslash_variant == 1 && op->ctx->tun.clover_fused;
}
int wilson_pipe_grid(lqcd_ctx_s* c, int nvirt, int prec);
int wilson_pipe_per_wg(lqcd_ctx_s* c, int nvirt);
int stencil_num_partials(lqcd_ctx_s* c, int kind, double r, int parity_mode, int prec = 0, bool clover = false);

// BLAS-1 / reductions (blas.hip)
int blas_dot(lqcd_ctx_s* c, const double2* a, const double2* b, size_t n, double* re, double* im, bool allreduce);
int blas_norm2(lqcd_ctx_s* c, const double2* a, size_t n, double* n2, bool allreduce);
int blas_axpy(lqcd_ctx_s* c, double ar, double ai, const double2* x, double2* y, size_t n);
int blas_axpby(lqcd_ctx_s* c, double ar, double ai, const double2* x, double br, double bi, double2* y, size_t n);
int blas_scale(lqcd_ctx_s* c, double ar, double ai, double2* x, size_t n);
int allreduce_host(lqcd_ctx_s* c, double* vals, int n);
int reduce_to_slot(lqcd_ctx_s* c, int nblocks, int nvals, int slot, bool allreduce, int cg_op = 0, const double* partial = nullptr, bool soa = false);   // partial: default the context's d_partial
int reduce_tail(lqcd_ctx_s* c, int nvals, int slot, int cg_op);
int reduce_pack_to_slot(lqcd_ctx_s* c, int nblocks, int slot, int cg_op);      // reduce_to_slot(nvals = 1, all-reduce) + the pack launch the folded schedule left waiting (StencilCall::defer_pack)
LQCD_REOPEN_P64 { int launch_pack_reduce(lqcd_ctx_s* c, const StencilCall& s, const double* partial, int nblocks, int slot, int op); }
int stream_grid(lqcd_ctx_s* c, size_t n);

// clover.hip
size_t clover_elems(const Geom& g);
int clover_build(lqcd_ctx_s* c, const lqcd_gauge_s* U, double2* clov, double kappa, double csw);
int clover_apply(lqcd_ctx_s* c, const double2* clov, lqcd_spinor_s* out, lqcd_spinor_s* in);
int clover_apply_parity(lqcd_ctx_s* c, const double2* clov, int parity, double2* out, const double2* in, double sa, const double2* z, double sz);
int clover_invert(lqcd_ctx_s* c, const double2* clov, double2* inv);
size_t clover_lambda_elems(const Geom& g);
int stout_gather_ext(lqcd_ctx_s* c, const lqcd_gauge_s* U, const double2* lamN, lqcd_gauge_s* G, double rho);      // partitioned stout back-propagation (clover.hip: halo-extended block)
double2* stout_lambda_buffer(lqcd_ctx_s* c);
int clover_force(lqcd_ctx_s* c, const lqcd_gauge_s* U, lqcd_gauge_s* out, lqcd_spinor_s* X, lqcd_spinor_s* Y, double2* lam, double kappa,
                 double csw, double scale, int accumulate);

// fields.hip
double2* spinor_block(lqcd_spinor_s* s, int p);
int gauge_ensure_recon12(lqcd_gauge_s* g);   // (re)builds the 12-real copy if the field changed; sets g->recon_ok
int gauge_ensure_recon12d(lqcd_gauge_s* g);  // (re)builds the "12 + delta" copy; sets g->delta_ok
int plaquette_local_sum(lqcd_gauge_s* g, const double2* const ghost[4], double* sum);
int gauge_pack_face(lqcd_gauge_s* g, int mu, double2* dst);

// md.hip: run every recorded / deferred single-direction link operation of the context now (no-op when there is none)
int links_flush(lqcd_ctx_s* c);
bool ctx_park_gauge(lqcd_gauge_s* g);      // capi.hip: true = called from a foreign thread while the context records link operations: parked, the caller must not touch it
int ctx_drain_parked(lqcd_ctx_s* c);       // frees what foreign threads parked (called on the context's thread)
int gauge_destroy_now(lqcd_gauge_s* g);     // fields.hip
std::mutex& view_mutex();                  // fields.hip: guards nviews / zombie of five-dimensional fields (slice views are created and destroyed under it)
bool ctx_is_live(const lqcd_ctx_s* c);     // capi.hip: finalizers run in any order -- a field may outlive its context
inline int links_flush_of(lqcd_ctx_s* c) { return (c && c->lazy.busy()) ? links_flush(c) : LQCD_OK; }
inline int links_flush_of(lqcd_gauge_s* g) { return g ? links_flush_of(g->ctx) : LQCD_OK; }
inline int links_flush_of(lqcd_op_s* o) { return o ? links_flush_of(o->ctx) : LQCD_OK; }

// scratch spinors
lqcd_spinor_s* scratch_get(lqcd_ctx_s* c, int kind, int subset);
void scratch_put(lqcd_spinor_s* s);

}  // namespace lqcd
