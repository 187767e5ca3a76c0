// mixed.hip -- mixed-precision CG on D^+D: fp32 inner solver, fp64 outer defect correction (SURVEY.md 8(f) rank 3 and 8(b)
// "later solve_mixed_cg"; BASELINE.json configs[4] "mixed-precision fp32 inner / fp64 outer CG").  The reference has no such
// solver (its solve_DinvX! is fp64 throughout); the contract kept here is the reference's stopping rule on the TRUE fp64
// residual: on return |b - D^+D x|^2 < eps, recomputed in fp64.
//
//   outer (fp64):  r = b - A x ;  while |r|^2 >= eps:   e ~ A^-1 (r/|r|) in fp32 ;  x += |r| e ;  r = b - A x
//   inner (fp32):  the same fused CG iteration as the fp64 solver (ops.hip) -- D p with |.|^2 partials, D^+ in update mode,
//                  fused x/p update, device-resident scalars -- on float2 copies of the links and of four spinors, through the
//                  fp32 build of the stencil (lqcd::p32, stencil.hip compiled with -DLQCD_F32).  Half the bytes per iteration.
// The residual is normalised before it is rounded to fp32, so the inner solver never sees the absolute scale.  If an outer
// step fails to reduce the true residual by 2x (fp32 accuracy exhausted) the solve is finished by the fp64 CG from the
// current iterate.  Partitioned lattices: the fp32 halos go through the same pack / RCCL / exterior sequence (ncclFloat).
#include "ops_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace lqcd {

constexpr int MB = 256;

__global__ __launch_bounds__(MB) void cvt_to_f32(float2* __restrict__ dst, const double2* __restrict__ src, size_t n, double scale) {
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n; i += (size_t)gridDim.x * MB) {
        const double2 v = src[i];
        dst[i] = make_float2((float)(v.x * scale), (float)(v.y * scale));
    }
}
// Wilson spinors: the fp32 fields of the inner solver keep two consecutive components of a site in one 16-byte word,
// [chunk][component pair 0..5][lane][2] (stencil.hip, sp12_off / co12: the 8-byte accesses of a plain float2 copy of the fp64
// arrangement run at 0.54-0.70x the 16-byte rate).  One thread converts one pair: two coalesced 16-byte loads, one 16-byte store.
// n2 = number of pairs = elements / 2 (a parity block is a whole number of 64-site chunks, so chunks run across both blocks).
__global__ __launch_bounds__(MB) void cvt_wilson_to_f32(float4* __restrict__ dst, const double2* __restrict__ src, size_t n2, double scale) {
    for (size_t t = (size_t)blockIdx.x * MB + threadIdx.x; t < n2; t += (size_t)gridDim.x * MB) {
        const size_t lane = t & 63, q = (t >> 6) % 6, chunk = t / 384;
        const double2 a = src[chunk * 768 + (2 * q) * 64 + lane], b = src[chunk * 768 + (2 * q + 1) * 64 + lane];
        dst[t] = make_float4((float)(a.x * scale), (float)(a.y * scale), (float)(b.x * scale), (float)(b.y * scale));
    }
}
// y (fp64, [chunk][12][64]) += a * x (fp32, paired)
__global__ __launch_bounds__(MB) void axpy_from_wilson_f32(double2* __restrict__ y, const float4* __restrict__ x, double a, size_t n2) {
    for (size_t t = (size_t)blockIdx.x * MB + threadIdx.x; t < n2; t += (size_t)gridDim.x * MB) {
        const size_t lane = t & 63, q = (t >> 6) % 6, chunk = t / 384;
        const float4 xv = x[t];
        double2* y0 = y + chunk * 768 + (2 * q) * 64 + lane;
        double2* y1 = y0 + 64;
        double2 u = *y0, v = *y1;
        u.x = fma(a, (double)xv.x, u.x); u.y = fma(a, (double)xv.y, u.y);
        v.x = fma(a, (double)xv.z, v.x); v.y = fma(a, (double)xv.w, v.y);
        *y0 = u; *y1 = v;
    }
}
// fp32 12-real copy of the links (rows 0,1) for the compressed split kernels of the inner solver, two elements per 16-byte word:
// [parity][chunk][mu][pair 0..2][lane][2] (stencil.hip, gl12_off).
__global__ __launch_bounds__(MB) void cvt_gauge12_f32(Geom g, const double2* __restrict__ src, float2* __restrict__ dst) {
    const int t = blockIdx.x * MB + threadIdx.x;
    if (t >= 2 * g.Vh * 4) return;
    const int mu = t & 3, s = t >> 2, p = s / g.Vh, i = s % g.Vh;
    const size_t so = glink_off(g, p, mu, i);
    const size_t d_o = ((((size_t)p * g.nch + (size_t)(i >> 6)) * 4 + mu) * 6) * 64 + (size_t)(i & 63) * 2;
    const int Gs = glink_stride(g);
    float4* d4 = reinterpret_cast<float4*>(dst + d_o);
    for (int q = 0; q < 3; q++) {
        const double2 a = src[so + (size_t)(2 * q) * Gs], b = src[so + (size_t)(2 * q + 1) * Gs];
        d4[(size_t)q * 64] = make_float4((float)a.x, (float)a.y, (float)b.x, (float)b.y);
    }
}
// y (fp64) += a * x (fp32)
__global__ __launch_bounds__(MB) void axpy_from_f32(double2* __restrict__ y, const float2* __restrict__ x, double a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n; i += (size_t)gridDim.x * MB) {
        const float2 xv = x[i];
        double2 yv = y[i];
        yv.x = fma(a, (double)xv.x, yv.x);
        yv.y = fma(a, (double)xv.y, yv.y);
        y[i] = yv;
    }
}
// |a|^2 and |b|^2 block partials, two per block
__global__ __launch_bounds__(MB) void norm2_two_kernel(const double2* __restrict__ a, const double2* __restrict__ b, size_t n, double* partial) {
    __shared__ double red[2][MB / 64];
    double sa = 0, sb = 0;
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n; i += (size_t)gridDim.x * MB) {
        const double2 u = a[i], v = b[i];
        sa = fma(u.x, u.x, sa); sa = fma(u.y, u.y, sa);
        sb = fma(v.x, v.x, sb); sb = fma(v.y, v.y, sb);
    }
    sa = wave_sum(sa); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sa; red[1][threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < MB / 64; w++) t += red[threadIdx.x][w];
        partial[2 * blockIdx.x + threadIdx.x] = t;
    }
}
// r = b - q - sigma x  (fp64) with |r|^2 block partials  (sigma = 0: x is not read)
__global__ __launch_bounds__(MB) void residual_kernel(double2* __restrict__ r, const double2* __restrict__ b, const double2* __restrict__ q,
                                                       const double2* __restrict__ x, double sigma, size_t n, double* partial) {
    __shared__ double red[MB / 64];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n; i += (size_t)gridDim.x * MB) {
        const double2 bv = b[i], qv = q[i];
        double2 rv = make_double2(bv.x - qv.x, bv.y - qv.y);
        if (sigma != 0.0) { const double2 xv = x[i]; rv.x = fma(-sigma, xv.x, rv.x); rv.y = fma(-sigma, xv.y, rv.y); }
        r[i] = rv;
        acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < MB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// fp32 tail of a CG iteration: x += alpha p ; p = r + beta p   (same flag protocol as cg_update_xp in ops.hip)
__global__ __launch_bounds__(MB) void cg32_update_xp(const double* __restrict__ s, float4* __restrict__ x, float4* __restrict__ p,
                                                      const float4* __restrict__ r, size_t n4) {
    if (s[S_XDONE] != 0.0) return;
    const float al = (float)s[S_ALPHA], be = (float)s[S_BETA];
    const bool cont = s[S_DONE] == 0.0;
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
        float4 pv = p[i], xv = x[i];
        xv.x = fmaf(al, pv.x, xv.x); xv.y = fmaf(al, pv.y, xv.y); xv.z = fmaf(al, pv.z, xv.z); xv.w = fmaf(al, pv.w, xv.w);
        x[i] = xv;
        if (cont) {
            const float4 rv = r[i];
            pv.x = fmaf(be, pv.x, rv.x); pv.y = fmaf(be, pv.y, rv.y); pv.z = fmaf(be, pv.z, rv.z); pv.w = fmaf(be, pv.w, rv.w);
            p[i] = pv;
        }
    }
}

// Deferred x update in fp32 (solvers.hip cg_update_even / cg_update_odd, same protocol): iteration k even: p_{k+1} = r + beta p_k into the
// OTHER buffer, x untouched, alpha_k kept in S_APREV (the iteration that converges completes x itself); k odd: x += alpha_{k-1} p_{k-1} +
// alpha_k p_k in the order of two single updates (identical bits), p_{k+1} = r + beta p_k over the dead p_{k-1}.
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ inline float4 ld4(const float4* p) {
    if constexpr (NT) { const v4f32 v = __builtin_nontemporal_load(reinterpret_cast<const v4f32*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    else return *p;
}
template <bool NT> __device__ inline void st4(float4* p, float4 v) {
    if constexpr (NT) { const v4f32 t = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(t, reinterpret_cast<v4f32*>(p)); }
    else *p = v;
}
// NT (tunable nt_blas, as in the fp64 update kernels): streaming loads / stores -- none of these fields is read again before it has left every cache
template <bool NT>
__global__ __launch_bounds__(MB) void cg32_update_even(double* __restrict__ s, float4* __restrict__ x, const float4* __restrict__ pk,
                                                        float4* __restrict__ pnext, const float4* __restrict__ r, size_t n4) {
    if (s[S_XDONE] != 0.0) return;
    const float al = (float)s[S_ALPHA], be = (float)s[S_BETA];
    if (s[S_DONE] == 0.0) {
        for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
            const float4 pv = ld4<NT>(pk + i), rv = ld4<NT>(r + i);
            float4 o;
            o.x = fmaf(be, pv.x, rv.x); o.y = fmaf(be, pv.y, rv.y); o.z = fmaf(be, pv.z, rv.z); o.w = fmaf(be, pv.w, rv.w);
            st4<NT>(pnext + i, o);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) s[S_APREV] = s[S_ALPHA];
    } else {
        for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
            const float4 pv = pk[i];
            float4 xv = x[i];
            xv.x = fmaf(al, pv.x, xv.x); xv.y = fmaf(al, pv.y, xv.y); xv.z = fmaf(al, pv.z, xv.z); xv.w = fmaf(al, pv.w, xv.w);
            x[i] = xv;
        }
    }
}
template <bool NT>
__global__ __launch_bounds__(MB) void cg32_update_odd(const double* __restrict__ s, float4* __restrict__ x, float4* __restrict__ pprev,
                                                       const float4* __restrict__ pk, const float4* __restrict__ r, size_t n4) {
    if (s[S_XDONE] != 0.0) return;
    const float ap = (float)s[S_APREV], al = (float)s[S_ALPHA], be = (float)s[S_BETA];
    const bool cont = s[S_DONE] == 0.0;
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
        const float4 pp = ld4<NT>(pprev + i), pv = ld4<NT>(pk + i), rv = ld4<NT>(r + i);
        float4 xv = ld4<NT>(x + i);
        xv.x = fmaf(ap, pp.x, xv.x); xv.y = fmaf(ap, pp.y, xv.y); xv.z = fmaf(ap, pp.z, xv.z); xv.w = fmaf(ap, pp.w, xv.w);
        xv.x = fmaf(al, pv.x, xv.x); xv.y = fmaf(al, pv.y, xv.y); xv.z = fmaf(al, pv.z, xv.z); xv.w = fmaf(al, pv.w, xv.w);
        st4<NT>(x + i, xv);
        if (cont) {
            float4 o;
            o.x = fmaf(be, pv.x, rv.x); o.y = fmaf(be, pv.y, rv.y); o.z = fmaf(be, pv.z, rv.z); o.w = fmaf(be, pv.w, rv.w);
            st4<NT>(pprev + i, o);
        }
    }
}
// x += alpha_prev p_prev for a solve that stopped, unconverged, behind an even iteration
__global__ __launch_bounds__(MB) void cg32_flush_x(const double* __restrict__ s, float4* __restrict__ x, const float4* __restrict__ pk, size_t n4) {
    if (s[S_DONE] != 0.0) return;
    const float ap = (float)s[S_APREV];
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
        const float4 pv = pk[i];
        float4 xv = x[i];
        xv.x = fmaf(ap, pv.x, xv.x); xv.y = fmaf(ap, pv.y, xv.y); xv.z = fmaf(ap, pv.z, xv.z); xv.w = fmaf(ap, pv.w, xv.w);
        x[i] = xv;
    }
}

// the same tail when x += alpha p has already been done in the epilogue of the update-mode D^+ (site-pair kernel): p = r + beta p only
__global__ __launch_bounds__(MB) void cg32_update_p(const double* __restrict__ s, float4* __restrict__ p, const float4* __restrict__ r, size_t n4) {
    if (s[S_DONE] != 0.0) return;
    const float be = (float)s[S_BETA];
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
        float4 pv = p[i];
        const float4 rv = r[i];
        pv.x = fmaf(be, pv.x, rv.x); pv.y = fmaf(be, pv.y, rv.y); pv.z = fmaf(be, pv.z, rv.z); pv.w = fmaf(be, pv.w, rv.w);
        p[i] = pv;
    }
}

// fp32 multi-shift update (solvers.hip ms_update_all on float4 = two elements): base system x += alpha p (optional), p = r + beta p, and
// every active shift x_j += a_j p_j ; p_j = b_j p_j + z_j r in one pass; frozen shifts cost nothing
__global__ __launch_bounds__(MB) void ms32_update_all(const double* __restrict__ sc, const double* __restrict__ ms, float4* const* __restrict__ ptr,
                                                       float4* __restrict__ x0, float4* __restrict__ p0, const float4* __restrict__ r, size_t n4, int ns) {
    if (sc[S_XDONE] != 0.0) return;
    const float al = (float)sc[S_ALPHA], be = (float)sc[S_BETA];
    for (size_t i = (size_t)blockIdx.x * MB + threadIdx.x; i < n4; i += (size_t)gridDim.x * MB) {
        const float4 rv = r[i];
        {
            float4 pv = p0[i];
            if (x0) {
                float4 xv = x0[i];
                xv.x = fmaf(al, pv.x, xv.x); xv.y = fmaf(al, pv.y, xv.y); xv.z = fmaf(al, pv.z, xv.z); xv.w = fmaf(al, pv.w, xv.w);
                x0[i] = xv;
            }
            pv.x = fmaf(be, pv.x, rv.x); pv.y = fmaf(be, pv.y, rv.y); pv.z = fmaf(be, pv.z, rv.z); pv.w = fmaf(be, pv.w, rv.w);
            p0[i] = pv;
        }
        for (int j = 0; j < ns; j++) {
            const double ad = ms[3 * ns + j], bd = ms[4 * ns + j], zd = ms[5 * ns + j];
            if (ad == 0.0 && bd == 0.0 && zd == 0.0) continue;       // frozen shift
            const float a = (float)ad, bb = (float)bd, z = (float)zd;
            float4* __restrict__ x = ptr[j];
            float4* __restrict__ p = ptr[ns + j];
            float4 pv = p[i], xv = x[i];
            xv.x = fmaf(a, pv.x, xv.x); xv.y = fmaf(a, pv.y, xv.y); xv.z = fmaf(a, pv.z, xv.z); xv.w = fmaf(a, pv.w, xv.w);
            pv.x = fmaf(bb, pv.x, z * rv.x); pv.y = fmaf(bb, pv.y, z * rv.y); pv.z = fmaf(bb, pv.z, z * rv.z); pv.w = fmaf(bb, pv.w, z * rv.w);
            x[i] = xv; p[i] = pv;
        }
    }
}

static int mix_alloc(lqcd_ctx_s* c, int slot, size_t bytes) {
    if (c->mix_bytes[slot] >= bytes) return LQCD_OK;
    if (c->mix_buf[slot]) HIPCHK(hipFree(c->mix_buf[slot]));
    c->mix_buf[slot] = nullptr; c->mix_bytes[slot] = 0;
    HIPCHK(hipMalloc(&c->mix_buf[slot], bytes));
    c->mix_bytes[slot] = bytes;
    return LQCD_OK;
}

struct Mix32 {
    void* gauge16 = nullptr;    // int16 links of the site-pair kernel (mixed_links16), or nullptr
    float2 *gauge, *gauge12, *clover, *x, *r, *p, *t, *p2;      // p2: second search-direction buffer (deferred x update), or nullptr
    size_t blk;   // elements per parity block
    int layout = 0;   // fp32 spinor layout: 0 plain (staggered), 1 Wilson component pairs (stencil.hip, LQCD_F32), 2 Wilson site pairs (stencil_pair32.hip)
};

// a StencilCall on fp32 fields (pointers travel as double2*, stencil_apply dispatches on prec)
static StencilCall call32(lqcd_op_s* op, const Mix32& m, float2* out, float2* in, int dagger) {
    StencilCall s;
    s.kind = op->kind;
    s.gauge = (const double2*)m.gauge;
    s.gauge12 = (const double2*)m.gauge12;
    s.gauge16 = m.gauge16;
    s.clover = (const double2*)m.clover;
    for (int p = 0; p < 2; p++) {
        s.out[p] = (double2*)(out + p * m.blk);
        s.in[p] = (const double2*)(in + p * m.blk);
        s.xin[p] = (const double2*)(in + p * m.blk);
    }
    if (op->kind == LQCD_WILSON) { s.a = 1.0; s.b = -op->km; }
    else { s.a = op->km; s.b = dagger ? -0.5 : 0.5; }
    s.r = op->r;
    s.dagger = dagger;
    s.parity_mode = 2;
    s.prec = m.layout == 2 ? 2 : 1;
    return s;
}

// fp32 CG on A e = rhs (|rhs|^2 = 1, e starts at 0) until the recursive residual drops below eps2 or maxiter; returns iterations
static int inner_cg32(lqcd_op_s* op, const Mix32& m, size_t n, double eps2, int maxiter, int* iters, double* rr_out) {
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipMemsetAsync(m.x, 0, n * sizeof(float2), c->stream));
    HIPCHK(hipMemcpyAsync(m.p, m.r, n * sizeof(float2), hipMemcpyDeviceToDevice, c->stream));
    double init[9] = {1.0, 0, 0, 0, 0, 0, eps2, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    const int nbs = stencil_num_partials(c, op->kind, op->r, 2, m.layout == 2 ? 2 : 1, op->csw != 0.0 && op->clover != nullptr), nbu = stream_grid(c, n / 2), check_every = 8;
    int it = 0, kq = 0;      // kq: iterations enqueued (the device executes them until it converges)
    double rr = 1.0;
    bool done = false;
    const bool defer = m.p2 != nullptr && !(m.layout == 2 && c->tun.mixed_xfuse);
    while (!done && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++, kq++) {
            float2* pk = (defer && (kq & 1)) ? m.p2 : m.p;       // p_k of an even k lives in m.p
            float2* po = (defer && (kq & 1)) ? m.p : m.p2;
            apply_bc(c, op->bc);
            StencilCall s1 = call32(op, m, m.t, pk, 0);
            s1.norm_partial = c->d_partial;
            s1.skip_flag = c->d_scal;                        // a no-op once the inner solve has converged inside a burst
            LQCHK(stencil_apply(c, s1));
            LQCHK(reduce_to_slot(c, nbs, 1, S_PQ, true, 1));
            StencilCall s2 = call32(op, m, m.t, m.t, 1);      // update mode: nothing is written to out
            s2.norm_partial = c->d_partial;
            s2.upd_scal = c->d_scal;
            s2.upd[0] = (double2*)m.r;
            s2.upd[1] = (double2*)(m.r + m.blk);
            const bool xfused = m.layout == 2 && c->tun.mixed_xfuse;      // site-pair kernel: x += alpha p in the epilogue of this launch
            if (xfused)
                for (int p = 0; p < 2; p++) { s2.xacc[p] = (double2*)(m.x + p * m.blk); s2.pacc[p] = (const double2*)(m.p + p * m.blk); }
            LQCHK(stencil_apply(c, s2));
            LQCHK(reduce_to_slot(c, nbs, 1, S_RRNEW, true, 2));
            if (defer && !(kq & 1)) {
                if (c->tun.nt_blas) hipLaunchKernelGGL(cg32_update_even<true>, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (const float4*)pk, (float4*)po, (const float4*)m.r, n / 2);
                else hipLaunchKernelGGL(cg32_update_even<false>, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (const float4*)pk, (float4*)po, (const float4*)m.r, n / 2);
            } else if (defer) {
                if (c->tun.nt_blas) hipLaunchKernelGGL(cg32_update_odd<true>, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (float4*)po, (const float4*)pk, (const float4*)m.r, n / 2);
                else hipLaunchKernelGGL(cg32_update_odd<false>, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (float4*)po, (const float4*)pk, (const float4*)m.r, n / 2);
            }
            else if (xfused) hipLaunchKernelGGL(cg32_update_p, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.p, (const float4*)m.r, n / 2);
            else hipLaunchKernelGGL(cg32_update_xp, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (float4*)m.p, (const float4*)m.r, n / 2);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        it = (int)c->h_scal[S_ITERS - S_RR];
        done = c->h_scal[S_DONE - S_RR] != 0.0;
        if (!std::isfinite(rr)) break;
    }
    if (defer && !done && (kq & 1)) {      // stopped, unconverged, behind an even iteration: its alpha p is still owed to x (p_k of an even k lives in m.p)
        hipLaunchKernelGGL(cg32_flush_x, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, (float4*)m.x, (const float4*)m.p, n / 2);
        HIPCHK(hipGetLastError());
    }
    *iters = it;
    *rr_out = rr;
    return LQCD_OK;
}

// the fp32 build of the stencil has the site-per-lane and the direction-split kernels (variants 0, 1): the variant is pinned for the duration of a
// mixed-precision solve (the fp64 applications of the outer loop are bit-identical across the split variants)
struct VariantPin {
    lqcd_ctx_s* c; int saved;
    explicit VariantPin(lqcd_ctx_s* c_) : c(c_), saved(c_->tun.dslash_variant) { if (saved >= 2) c->tun.dslash_variant = 1; }
    ~VariantPin() { c->tun.dslash_variant = saved; }
};

// fp32 copies of the operator's links (18 reals, and 12 reals under the fp64 path's rule) and clover blocks, and the four fp32 work
// vectors, shared by the mixed-precision CG and the mixed-precision multi-shift CG.  The link copies follow the field
// (handle, version): a sequence of solves on the same links converts once.
static int mix_prepare(lqcd_op_s* op, size_t n, Mix32& m, bool eo_chain = false) {
    lqcd_ctx_s* c = op->ctx;
    LQCHK(halo_schedule_settle(op));      // the inner solvers count |.|^2 partials: the halo schedule (folded or not, also for the fp32 build) is fixed from here on
    const bool clov = op->csw != 0.0 && op->clover != nullptr;
    if (clov && !(op->r == 1.0 && c->tun.dslash_variant == 1)) {
        set_error("mixed-precision solvers: the Wilson-clover operator needs the direction-split kernel (r = 1, dslash_variant = 1) in the fp32 inner solver");
        return LQCD_ERR_UNSUPPORTED;
    }
    const size_t ng = op->gauge->elems;
    LQCHK(mix_alloc(c, 0, ng * sizeof(float2)));
    for (int k = 1; k <= 4; k++) LQCHK(mix_alloc(c, k, n * sizeof(float2)));
    LQCHK(mix_alloc(c, 5, gauge12_elems(c->geom) * sizeof(float2)));
    m.gauge = (float2*)c->mix_buf[0];
    // 12-real fp32 links only under the rule of the fp64 path: tunable gauge_recon = 12 AND every link of this field version unitary to
    // 1e-14 (gauge_ensure_recon12).  For anything else (non-unitary test fields, smeared links) the inner operator reads the 18-real
    // fp32 copy -- a rebuilt third row would differ from the outer operator by O(1) and the defect correction would stall.
    const bool use12 = c->tun.gauge_recon == 12 && gauge_ensure_recon12(op->gauge) == LQCD_OK && op->gauge->recon_ok;
    m.gauge12 = use12 ? (float2*)c->mix_buf[5] : nullptr;
    m.clover = nullptr;
    if (clov) {
        LQCHK(mix_alloc(c, 6, clover_elems(c->geom) * sizeof(float2)));
        m.clover = (float2*)c->mix_buf[6];
    }
    m.x = (float2*)c->mix_buf[1]; m.r = (float2*)c->mix_buf[2]; m.p = (float2*)c->mix_buf[3]; m.t = (float2*)c->mix_buf[4];
    m.p2 = nullptr;
    if (c->tun.mixed_defer_x) { LQCHK(mix_alloc(c, 8, n * sizeof(float2))); m.p2 = (float2*)c->mix_buf[8]; }
    m.blk = n / 2;
    // site-pair fp32 kernel (stencil_pair32.hip; tunable mixed_pair32): plain Wilson r = 1 with 12-real links on an unpartitioned lattice whose
    // geometry admits it; the fp32 fields of the solve (links in mix_buf[5], the four vectors) then live in the pair layout
    const bool pair = op->kind == LQCD_WILSON && op->r == 1.0 && !clov && use12 && c->tun.mixed_pair32 && c->tun.dslash_variant == 1 &&
                      pair32_geometry_ok(c) && n == (size_t)2 * 12 * c->geom.Vs;
    m.layout = pair ? 2 : (op->kind == LQCD_WILSON ? 1 : 0);
    c->tun.pair32_active = pair ? 1 : 0;
    const int glayout = pair ? 2 : 1;
    const bool same_links = c->mix_gauge_of == (const void*)op->gauge && c->mix_gauge_version == op->gauge->version;
    if (!same_links) { c->mix_gauge18_valid = false; c->mix_gauge12_valid = false; c->mix_gauge16_valid = false; }
    const bool need18 = !pair;      // the site-pair kernel reads the 12-real pair copy alone: no 18-real fp32 copy per link update (0.35 ms at 32^3x64)
    if (need18 && !c->mix_gauge18_valid) {
        hipLaunchKernelGGL(cvt_to_f32, dim3(stream_grid(c, ng)), dim3(MB), 0, c->stream, m.gauge, op->gauge->data, ng, 1.0);
        c->mix_gauge18_valid = true;
    }
    if (clov) {     // fp32 copy of the packed clover blocks (same layout); A follows the links first
        if (op->clover_version != op->gauge->version) {
            LQCHK(clover_build(c, op->gauge, op->clover, op->km, op->csw));
            op->clover_version = op->gauge->version;
        }
        const size_t nc = clover_elems(c->geom);
        hipLaunchKernelGGL(cvt_to_f32, dim3(stream_grid(c, nc)), dim3(MB), 0, c->stream, m.clover, op->clover, nc, 1.0);
    }
    const bool only16 = pair && (c->tun.mixed_links16 >= 2 || (c->tun.mixed_links16 == 1 && eo_chain));      // the site-pair kernel will read the int16 copy alone: no fp32 pair copy for this solve
    if (use12 && !only16 && !(c->mix_gauge12_valid && c->mix_gauge12_layout == glayout)) {
        if (pair) LQCHK(pair32_cvt_gauge12(c, m.gauge12, op->gauge->data12));
        else hipLaunchKernelGGL(cvt_gauge12_f32, dim3((2 * c->geom.Vh * 4 + MB - 1) / MB), dim3(MB), 0, c->stream, c->geom, op->gauge->data, m.gauge12);
        c->mix_gauge12_valid = true;
        c->mix_gauge12_layout = glayout;
    }
    m.gauge16 = nullptr;
    if (pair && (c->tun.mixed_links16 >= 2 || (c->tun.mixed_links16 == 1 && eo_chain))) {
        LQCHK(mix_alloc(c, 9, gauge12_elems(c->geom) * sizeof(float2) / 2));
        if (!c->mix_gauge16_valid) LQCHK(pair32_cvt_gauge16(c, c->mix_buf[9], op->gauge->data12));
        c->mix_gauge16_valid = true;
        m.gauge16 = c->mix_buf[9];
    }
    c->mix_gauge_of = (const void*)op->gauge;
    c->mix_gauge_version = op->gauge->version;
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
// fp64 -> fp32 (scaled) and y (fp64) += a * x (fp32) in the layout of the field kind (Wilson: component pairs)
static int to_f32(lqcd_ctx_s* c, int layout, float2* dst, const double2* src, size_t n, double scale) {
    if (layout == 2) return pair32_cvt_spinor(c, dst, src, scale);
    if (layout == 1) hipLaunchKernelGGL(cvt_wilson_to_f32, dim3(stream_grid(c, n / 2)), dim3(MB), 0, c->stream, (float4*)dst, src, n / 2, scale);
    else hipLaunchKernelGGL(cvt_to_f32, dim3(stream_grid(c, n)), dim3(MB), 0, c->stream, dst, src, n, scale);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
static int add_from_f32(lqcd_ctx_s* c, int layout, double2* y, const float2* x, double a, size_t n) {
    if (layout == 2) return pair32_axpy_to_f64(c, y, x, a);
    if (layout == 1) hipLaunchKernelGGL(axpy_from_wilson_f32, dim3(stream_grid(c, n / 2)), dim3(MB), 0, c->stream, y, (const float4*)x, a, n / 2);
    else hipLaunchKernelGGL(axpy_from_f32, dim3(stream_grid(c, n)), dim3(MB), 0, c->stream, y, x, a, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- mixed-precision even-odd BiCGStab (plain Wilson)
// M x = rhs on the even sites, M = 1 - k^2 H_eo H_oe:  outer fp64 defect correction around the fused chain of solvers.hip run on fp32 copies --
//     r = rhs - M x (fp64) ;  while |r|^2 >= eps:  e ~ M^-1 (r / |r|) by the fp32 chain to a relative residual tol ;  x += |r| e ;  r = rhs - M x
// The inner chain is the fp64 one (bicgstab_eo_wilson) in structure: the Schur operator's second hop forms the next inner product in its epilogue
// (fp32 build of the direction-split kernel, double partials), reductions and scalar steps (double) run in the prologues of the three streaming
// kernels on lattices of <= 1024 chunks per parity.  Vectors are float4 words (two complex components: the component-pair layout is flat to them).
// The stopping rule is enforced on the TRUE fp64 residual of the Schur system, so lqcd_solve_bicgstab_eo keeps its contract; a correction step that
// fails to gain a factor 4 hands the rest of the solve to the fp64 chain.
__device__ inline void cfma32(float& xr, float& xi, float ar, float ai, float br, float bi) {      // x += a b (complex)
    xr = fmaf(ar, br, xr); xr = fmaf(-ai, bi, xr);
    xi = fmaf(ar, bi, xi); xi = fmaf(ai, br, xi);
}
// the two complex numbers of a 16-byte word: component pairs (re0, im0, re1, im1) of the one-site-per-lane fp32 build, or the site pairs
// (reA, reB, imA, imB) of stencil_pair32.hip
template <bool PAIR>
__device__ inline void cfma4(float4& x, float ar, float ai, const float4 b) {      // x += a b on both
    if constexpr (PAIR) { cfma32(x.x, x.z, ar, ai, b.x, b.z); cfma32(x.y, x.w, ar, ai, b.y, b.w); }
    else { cfma32(x.x, x.y, ar, ai, b.x, b.y); cfma32(x.z, x.w, ar, ai, b.z, b.w); }
}
__device__ inline double norm4(const float4 v) { return (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w); }
template <bool PAIR>
__device__ inline void cdot4(double& re, double& im, const float4 z, const float4 r) {      // += conj(z) r on both
    if constexpr (PAIR) {
        re += (double)(z.x * r.x + z.z * r.z) + (double)(z.y * r.y + z.w * r.w);
        im += (double)(z.x * r.z - z.z * r.x) + (double)(z.y * r.w - z.w * r.y);
    } else {
        re += (double)(z.x * r.x + z.y * r.y) + (double)(z.z * r.z + z.w * r.w);
        im += (double)(z.x * r.y - z.y * r.x) + (double)(z.z * r.w - z.w * r.z);
    }
}
// s = r - alpha v ; partial |s|^2
template <bool PAIR>
__global__ __launch_bounds__(UB) void bicgf32_s(BicgF a, float4* __restrict__ s, const float4* __restrict__ r, const float4* __restrict__ v, size_t n4) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    float4 pr[2], pv[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) { pr[e] = r[i]; pv[e] = v[i]; } }
    c2 r0v, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]};
    if (a.fold) { double t3[3]; block_sum_partials<3>(a.pin, a.pin_n, t3, a.pin_soa != 0); r0v.re = t3[0]; r0v.im = t3[1]; }
    else { r0v.re = a.sc[B_R0V]; r0v.im = a.sc[B_R0V + 1]; }
    const c2 al = bicg_alpha(rho, r0v);
    if (a.pin3 && a.sc[B_UNSURE] != 0.0) {      // merged chain: the last update launch left the stopping test to the |r'|^2 it summed (solvers.hip bicgf_s)
        double t1[1];
        block_sum_partials<1>(a.pin3, a.pin3_n, t1);
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.sc[B_RES] = t1[0]; a.sc[B_RR] = t1[0]; }
        if (t1[0] < a.sc[B_EPS]) { if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[B_DONE] = 1.0; return; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.sc[B_R0V] = r0v.re; a.sc[B_R0V + 1] = r0v.im; a.sc[B_ALPHA] = al.re; a.sc[B_ALPHA + 1] = al.im; }
    const float ar = -(float)al.re, ai = -(float)al.im;
    double acc[1] = {0};
    auto one = [&](size_t i, float4 sv, const float4 vv) {
        cfma4<PAIR>(sv, ar, ai, vv);
        s[i] = sv;
        acc[0] += norm4(sv);
    };
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) one(i, pr[e], pv[e]); }
    for (size_t i = i0 + 2 * stride; i < n4; i += stride) one(i, r[i], v[i]);
    block_reduce_nv<1>(acc, a.pout);
}
// x += alpha p + omega s ; r = s - omega t ; partials |r|^2, <r0, r>
template <bool PAIR>
__global__ __launch_bounds__(UB) void bicgf32_xr(BicgF a, float4* __restrict__ x, float4* __restrict__ r, const float4* __restrict__ p, const float4* __restrict__ s,
                                                  const float4* __restrict__ t, const float4* __restrict__ r0, size_t n4) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    float4 pp[2], ps[2], pt[2], pz[2], px[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) { pp[e] = p[i]; ps[e] = s[i]; pt[e] = t[i]; pz[e] = r0[i]; px[e] = x[i]; } }
    const float ar = (float)a.sc[B_ALPHA], ai = (float)a.sc[B_ALPHA + 1];
    double ss, tt;
    c2 ts;
    if (a.fold) {
        double t1[1], t3[3];
        block_sum_partials<1>(a.pin2, a.pin2_n, t1);
        block_sum_partials<3>(a.pin, a.pin_n, t3, a.pin_soa != 0);
        ss = t1[0]; ts.re = t3[0]; ts.im = t3[1]; tt = t3[2];
    } else { ss = a.sc[B_SS]; ts.re = a.sc[B_TS]; ts.im = a.sc[B_TS + 1]; tt = a.sc[B_TT]; }
    const bool half = ss < a.sc[B_EPS];
    const c2 om = bicg_omega(ts, tt, half);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.sc[B_SS] = ss; a.sc[B_HALF] = half ? 1.0 : 0.0; a.sc[B_TS] = ts.re; a.sc[B_TS + 1] = ts.im; a.sc[B_TT] = tt;
        a.sc[B_OMEGA] = om.re; a.sc[B_OMEGA + 1] = om.im;
    }
    const float wr = (float)om.re, wi = (float)om.im;
    double acc[3] = {0, 0, 0};
    auto one = [&](size_t i, const float4 pv, const float4 sv, const float4 tv, const float4 zv, float4 xv) {
        float4 rv = sv;
        cfma4<PAIR>(xv, ar, ai, pv);
        cfma4<PAIR>(xv, wr, wi, sv);
        cfma4<PAIR>(rv, -wr, -wi, tv);
        x[i] = xv; r[i] = rv;
        acc[0] += norm4(rv);
        cdot4<PAIR>(acc[1], acc[2], zv, rv);        // <r0, r> = conj(r0) r
    };
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) one(i, pp[e], ps[e], pt[e], pz[e], px[e]); }
    for (size_t i = i0 + 2 * stride; i < n4; i += stride) one(i, p[i], s[i], t[i], r0[i], x[i]);
    block_reduce_nv<3>(acc, a.pout);
}
// p = r + beta (p - omega v)
template <bool PAIR>
__global__ __launch_bounds__(UB) void bicgf32_p(BicgF a, float4* __restrict__ p, const float4* __restrict__ r, const float4* __restrict__ v, size_t n4) {
    if (!a.cont && a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    float4 pv_[2], pr[2], pp[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) { pv_[e] = v[i]; pr[e] = r[i]; pp[e] = p[i]; } }
    const double wrd = a.sc[B_OMEGA], wid = a.sc[B_OMEGA + 1];
    const bool half = a.sc[B_HALF] != 0.0;
    double rrn;
    c2 rho1, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]}, al = {a.sc[B_ALPHA], a.sc[B_ALPHA + 1]}, om = {wrd, wid};
    if (a.fold) { double t3[3]; block_sum_partials<3>(a.pin, a.pin_n, t3); rrn = t3[0]; rho1.re = t3[1]; rho1.im = t3[2]; }
    else { rrn = a.sc[B_RR]; rho1.re = a.sc[B_RHO1]; rho1.im = a.sc[B_RHO1 + 1]; }
    const double rr = half ? a.sc[B_SS] : rrn;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (a.cont) {      // r is the true residual of a reliable update, rho1 = <r0, r> with it: no test, the chain is armed for the next stretch
        if (lead) { a.sc[B_RES] = rrn; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im; a.sc[B_EPS] = a.cont_eps; a.sc[B_DONE] = 0.0; }
    } else {
        if (lead) { a.sc[B_ITERS] += 1.0; a.sc[B_RES] = rr; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im; }
        if (half || rr < a.sc[B_EPS]) { if (lead) a.sc[B_DONE] = 1.0; return; }
        if (!(fabs(rr) <= 1.79e308)) { if (lead) a.sc[B_DONE] = 2.0; return; }
    }
    const c2 be = bicg_beta(rho1, rho, al, om);
    if (lead) { a.sc[B_BETA] = be.re; a.sc[B_BETA + 1] = be.im; a.sc[a.rho_out] = rho1.re; a.sc[a.rho_out + 1] = rho1.im; }
    const float br = (float)be.re, bi = (float)be.im, wr = -(float)wrd, wi = -(float)wid;
    auto one = [&](size_t i, const float4 vv, const float4 rv, float4 pv) {
        cfma4<PAIR>(pv, wr, wi, vv);      // p - omega v
        float4 o = rv;
        cfma4<PAIR>(o, br, bi, pv);
        p[i] = o;
    };
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) one(i, pv_[e], pr[e], pp[e]); }
    for (size_t i = i0 + 2 * stride; i < n4; i += stride) one(i, v[i], r[i], p[i]);
}
// the merged update launch of solvers.hip bicgf_xrp_rec on fp32 vectors: x += alpha p + omega s ; r = s - omega t ; p = r + beta (p - omega v) with
// rho' = rho - alpha <r0, v> - omega <r0, t> and |r'|^2 = |s|^2 - |<t, s>|^2 / |t|^2 (double scalars from double sums of fp32 products)
template <bool PAIR>
__global__ __launch_bounds__(UB) void bicgf32_xrp_rec(BicgF a, float4* __restrict__ x, float4* __restrict__ r, float4* __restrict__ p, const float4* __restrict__ s,
                                                       const float4* __restrict__ t, const float4* __restrict__ v, size_t n4) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    float4 pp[2], ps[2], pt[2], px[2], pv_[2];
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) { pp[e] = p[i]; ps[e] = s[i]; pt[e] = t[i]; px[e] = x[i]; pv_[e] = v[i]; } }
    const c2 al = {a.sc[B_ALPHA], a.sc[B_ALPHA + 1]}, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]}, r0v = {a.sc[B_R0V], a.sc[B_R0V + 1]};
    double ss, tt;
    c2 ts, r0t;
    if (a.fold == 1) {
        double t1[1], t5[5];
        block_sum_partials<1>(a.pin2, a.pin2_n, t1);
        block_sum_partials<5>(a.pin, a.pin_n, t5, a.pin_soa != 0);
        ss = t1[0]; ts.re = t5[0]; ts.im = t5[1]; tt = t5[2]; r0t.re = t5[3]; r0t.im = t5[4];
    } else {
        if (a.fold == 2) { double t1[1]; block_sum_partials<1>(a.pin2, a.pin2_n, t1); ss = t1[0]; }      // large lattices: the <= 1024 partials of |s|^2 are still summed here (one launch less)
        else ss = a.sc[B_SS];
        ts.re = a.sc[B_TS5]; ts.im = a.sc[B_TS5 + 1]; tt = a.sc[B_TS5 + 2]; r0t.re = a.sc[B_TS5 + 3]; r0t.im = a.sc[B_TS5 + 4];
    }
    const bool half = ss < a.sc[B_EPS];
    const c2 om = bicg_omega(ts, tt, half);
    c2 rho1;
    rho1.re = rho.re - (al.re * r0v.re - al.im * r0v.im) - (om.re * r0t.re - om.im * r0t.im);
    rho1.im = rho.im - (al.re * r0v.im + al.im * r0v.re) - (om.re * r0t.im + om.im * r0t.re);
    const double rrn = half ? ss : ss - (ts.re * ts.re + ts.im * ts.im) / tt;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const bool finite = fabs(rrn) <= 1.79e308 && fabs(ss) <= 1.79e308;
    const bool done = half || (finite && rrn < a.sc[B_EPS] && rrn > a.guard * ss);
    const bool unsure = !half && finite && !(rrn > a.guard * ss);
    if (lead) {
        a.sc[B_SS] = ss; a.sc[B_HALF] = half ? 1.0 : 0.0; a.sc[B_TS] = ts.re; a.sc[B_TS + 1] = ts.im; a.sc[B_TT] = tt;
        a.sc[B_OMEGA] = om.re; a.sc[B_OMEGA + 1] = om.im;
        a.sc[B_ITERS] += 1.0; a.sc[B_RES] = rrn; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im;
        if (done) a.sc[B_DONE] = 1.0;
        else if (!finite) a.sc[B_DONE] = 2.0;
    }
    const bool go_on = !done && finite;
    c2 be = {0.0, 0.0};
    if (go_on) {
        be = bicg_beta(rho1, rho, al, om);
        if (lead) { a.sc[B_BETA] = be.re; a.sc[B_BETA + 1] = be.im; a.sc[a.rho_out] = rho1.re; a.sc[a.rho_out + 1] = rho1.im; }
    }
    const float ar = (float)al.re, ai = (float)al.im, wr = (float)om.re, wi = (float)om.im, br = (float)be.re, bi = (float)be.im;
    double acc[1] = {0};
    auto one = [&](size_t i, float4 pv, const float4 sv, const float4 tv, float4 xv, const float4 vv) {
        float4 rv = sv;
        cfma4<PAIR>(xv, ar, ai, pv);
        cfma4<PAIR>(xv, wr, wi, sv);
        cfma4<PAIR>(rv, -wr, -wi, tv);
        x[i] = xv; r[i] = rv;
        acc[0] += norm4(rv);
        if (go_on) {
            cfma4<PAIR>(pv, -wr, -wi, vv);      // p - omega v
            float4 o = rv;
            cfma4<PAIR>(o, br, bi, pv);
            p[i] = o;
        }
    };
#pragma unroll
    for (int e = 0; e < 2; e++) { const size_t i = i0 + e * stride; if (i < n4) one(i, pp[e], ps[e], pt[e], px[e], pv_[e]); }
    for (size_t i = i0 + 2 * stride; i < n4; i += stride) one(i, p[i], s[i], t[i], x[i], v[i]);
    if (unsure) block_reduce_nv<1>(acc, a.pout);
    if (lead) a.sc[B_UNSURE] = unsure ? 1.0 : 0.0;
}
// partials |r|^2, <r0, r> (the three values bicgf32_xr leaves for bicgf32_p) of a residual that was replaced by the true one
template <bool PAIR>
__global__ __launch_bounds__(UB) void bicgf32_r0r(BicgF a, const float4* __restrict__ r0, const float4* __restrict__ r, size_t n4) {
    double acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n4; i += (size_t)gridDim.x * UB) {
        const float4 rv = r[i];
        acc[0] += norm4(rv);
        cdot4<PAIR>(acc[1], acc[2], r0[i], rv);
    }
    block_reduce_nv<3>(acc, a.pout);
}
// r = rhs - q (fp64) with |r|^2 partials is residual_kernel above (sigma = 0)

struct Eo32 {
    float2 *gauge, *gauge12;
    const void* gauge16 = nullptr;
    const float2* ainv = nullptr;                  // Wilson-clover: fp32 copy of the packed inverse clover blocks (applied to the hop sums inside the hops)
    float2 *x, *r, *r0, *p, *v, *s, *t, *to;      // half-lattice vectors
    int layout;                                    // 1: component pairs (fp32 build of stencil.hip), 2: site pairs (stencil_pair32.hip: half the launches' latencies per site)
};
// one Schur application on the fp32 fields: to = H_oe in, out = in - k^2 H_eo to [+ inner-product epilogue]
static int schur32(lqcd_op_s* op, const Eo32& m, float2* out, float2* in, const float2* z, double* dotp, int conj, int dg, const double* skip, const float2* z2 = nullptr, bool soa = false) {
    lqcd_ctx_s* c = op->ctx;
    StencilCall s1;
    s1.kind = LQCD_WILSON; s1.gauge = (const double2*)m.gauge; s1.gauge12 = (const double2*)m.gauge12; s1.gauge16 = m.gauge16;
    s1.out[0] = nullptr; s1.out[1] = (double2*)m.to; s1.in[0] = (const double2*)in; s1.in[1] = nullptr; s1.xin[0] = s1.xin[1] = nullptr;
    s1.a = 0.0; s1.b = 1.0; s1.r = 1.0; s1.dagger = dg; s1.parity_mode = 1; s1.prec = m.layout == 2 ? 2 : 1; s1.skip_flag = skip;
    if (m.ainv) { s1.clover = (const double2*)m.ainv; s1.clover_on_hop = 1; }
    LQCHK(stencil_apply(c, s1));
    StencilCall s2;
    s2.kind = LQCD_WILSON; s2.gauge = (const double2*)m.gauge; s2.gauge12 = (const double2*)m.gauge12; s2.gauge16 = m.gauge16;
    s2.out[0] = (double2*)out; s2.out[1] = nullptr; s2.in[0] = nullptr; s2.in[1] = (const double2*)m.to; s2.xin[0] = (const double2*)in; s2.xin[1] = nullptr;
    s2.a = 1.0; s2.b = -op->km * op->km; s2.r = 1.0; s2.dagger = dg; s2.parity_mode = 0; s2.prec = m.layout == 2 ? 2 : 1; s2.skip_flag = skip;
    if (m.ainv) { s2.clover = (const double2*)m.ainv; s2.clover_on_hop = 1; }
    if (z) { s2.dot_z[0] = (const double2*)z; s2.dot_z[1] = nullptr; s2.dot_partial = dotp; s2.dot_conj = conj | (soa ? 2 : 0); }
    if (z2) { s2.dot_z2[0] = (const double2*)z2; s2.dot_z2[1] = nullptr; }
    return stencil_apply(c, s2);
}
// e ~ M^-1 rhs32 (|rhs32|^2 = rho0, zero guess) until the recursive residual is below eps2; m.r holds rhs32 on entry
// cont (reliable update, tunable bicg_reliable): m.r holds the true residual in the units of the chain's first right-hand side; the chain keeps p, v, r0 and its scalars,
// x starts again at 0, <r0, r> is formed with the new r and p = r + beta (p - omega v) as the iteration that stopped would have done; *iters counts from the restart
static int inner_bicgstab_eo32(lqcd_op_s* op, const Eo32& m, size_t nh, int dg, double eps2, int maxiter, int* iters, bool cont = false, int first_burst = 0, int* full_stop = nullptr, bool pre_init = false, double rho0 = 1.0) {
    lqcd_ctx_s* c = op->ctx;
    const size_t n4 = (size_t)6 * c->geom.Vh, b32 = nh * sizeof(float2);      // the sites only: the padding chunk of a parity block is not part of a pair field
    const int nbs = m.layout == 2 ? pair32_num_blocks(c) / 2 : (c->geom.Vh + 63) / 64;      // (dot instances: one workgroup per 64-site chunk, whatever dslash_pipe says)
    const int nbk = (int)std::min<size_t>(1024, (n4 + UB - 1) / UB);
    const bool fold = c->tun.bicg_fused >= 2 && nbs <= 1024;
    // the merged update launch on the two recurrences (bicg_fused = 4, solvers.hip)
    const bool rec = c->tun.bicg_fused == 4;
    double* P0 = c->d_partial; double* P1 = P0 + (size_t)3 * nbs; double* P2 = P1 + nbk; double* P3 = P2 + (size_t)(rec ? 5 : 3) * nbs;
    const double* skip = c->d_scal + (B_DONE - S_DONE);
    const bool soa = c->tun.bicg_dot_soa >= 2 || (c->tun.bicg_dot_soa == 1 && !fold && nbs > 1024);      // (solvers.hip: [value][workgroup] dot partials)
    if (!pre_init) HIPCHK(hipMemsetAsync(m.x, 0, b32, c->stream));      // (pre_init: the conversion that made m.r also set x = 0, r0 = p = r)
    int it = 0, enq = 0;
    if (!cont) {
        if (!pre_init) {
            HIPCHK(hipMemcpyAsync(m.r0, m.r, b32, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(hipMemcpyAsync(m.p, m.r, b32, hipMemcpyDeviceToDevice, c->stream));
        }
        double init[B_END - B_RHO] = {0};
        init[B_RHO - B_RHO] = rho0; init[B_RHOB - B_RHO] = rho0; init[B_EPS - B_RHO] = eps2; init[B_RES - B_RHO] = rho0;      // rho_0 = <r0, r> = |r|^2 (1 for a normalised right-hand side)
        HIPCHK(hipMemcpyAsync(c->d_scal + B_RHO, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    } else {
        it = enq = *iters;      // iterations of the chain so far: iteration e reads rho from slot e & 1 and leaves the next one in the other
        BicgF a;
        a.sc = c->d_scal; a.fold = fold ? 1 : 0;
        a.rho_in = ((enq - 1) & 1) ? B_RHOB : B_RHO; a.rho_out = ((enq - 1) & 1) ? B_RHO : B_RHOB;
        a.pin = P3; a.pin_n = nbk; a.pin2 = nullptr; a.pin2_n = 0; a.pout = P3;
        a.cont = 1; a.cont_eps = eps2;
        if (m.layout == 2) hipLaunchKernelGGL(bicgf32_r0r<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (const float4*)m.r0, (const float4*)m.r, n4);
        else hipLaunchKernelGGL(bicgf32_r0r<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (const float4*)m.r0, (const float4*)m.r, n4);
        if (!fold) LQCHK(reduce_to_slot(c, nbk, 3, B_RR, true, 0, P3));
        a.pout = nullptr;
        if (m.layout == 2) hipLaunchKernelGGL(bicgf32_p<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.p, (const float4*)m.r, (const float4*)m.v, n4);
        else hipLaunchKernelGGL(bicgf32_p<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.p, (const float4*)m.r, (const float4*)m.v, n4);
        HIPCHK(hipGetLastError());
    }
    int check_every = first_burst > 0 ? std::min(first_burst, 64) : std::max(4, std::min(op->bicg_hint - 1, 64));
    double done = 0.0;
    while (done == 0.0 && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        check_every = 2;
        for (int q = 0; q < burst; q++, enq++) {
            BicgF a;
            a.sc = c->d_scal; a.fold = fold ? 1 : 0;
            a.rho_in = (enq & 1) ? B_RHOB : B_RHO; a.rho_out = (enq & 1) ? B_RHO : B_RHOB;
            a.pin2 = nullptr; a.pin2_n = 0;
            LQCHK(schur32(op, m, m.v, m.p, m.r0, P0, 0, dg, skip, nullptr, soa));
            if (!fold) LQCHK(reduce_to_slot(c, nbs, 3, B_R0V, true, 0, P0, soa));
            a.pin = P0; a.pin_n = nbs; a.pin_soa = soa ? 1 : 0; a.pout = P1;
            if (rec) { a.pin3 = P3; a.pin3_n = nbk; }
            if (m.layout == 2) hipLaunchKernelGGL(bicgf32_s<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.s, (const float4*)m.r, (const float4*)m.v, n4);
            else hipLaunchKernelGGL(bicgf32_s<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.s, (const float4*)m.r, (const float4*)m.v, n4);
            if (!fold && !rec) LQCHK(reduce_to_slot(c, nbk, 1, B_SS, true, 0, P1));      // (merged chain: the update launch sums the <= 1024 partials of |s|^2 itself)
            if (rec) {
                LQCHK(schur32(op, m, m.t, m.s, m.s, P2, 1, dg, skip, m.r0, soa));
                if (!fold) LQCHK(reduce_to_slot(c, nbs, 5, B_TS5, true, 0, P2, soa));
                a.pin = P2; a.pin_n = nbs; a.pin2 = P1; a.pin2_n = nbk; a.pout = P3;
                if (!fold) a.fold = 2;
                a.guard = std::pow(10.0, -(double)c->tun.bicg_rec_guard);
                if (m.layout == 2) hipLaunchKernelGGL(bicgf32_xrp_rec<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.x, (float4*)m.r, (float4*)m.p, (const float4*)m.s,
                                                      (const float4*)m.t, (const float4*)m.v, n4);
                else hipLaunchKernelGGL(bicgf32_xrp_rec<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.x, (float4*)m.r, (float4*)m.p, (const float4*)m.s,
                                        (const float4*)m.t, (const float4*)m.v, n4);
                HIPCHK(hipGetLastError());
                continue;
            }
            LQCHK(schur32(op, m, m.t, m.s, m.s, P2, 1, dg, skip, nullptr, soa));
            if (!fold) LQCHK(reduce_to_slot(c, nbs, 3, B_TS, true, 0, P2, soa));
            a.pin = P2; a.pin_n = nbs; a.pin2 = P1; a.pin2_n = nbk; a.pout = P3;
            if (m.layout == 2) hipLaunchKernelGGL(bicgf32_xr<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.x, (float4*)m.r, (const float4*)m.p, (const float4*)m.s,
                                                  (const float4*)m.t, (const float4*)m.r0, n4);
            else hipLaunchKernelGGL(bicgf32_xr<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.x, (float4*)m.r, (const float4*)m.p, (const float4*)m.s,
                                    (const float4*)m.t, (const float4*)m.r0, n4);
            if (!fold) LQCHK(reduce_to_slot(c, nbk, 3, B_RR, true, 0, P3));
            a.pin = P3; a.pin_n = nbk; a.pin_soa = 0; a.pin2 = nullptr; a.pin2_n = 0; a.pout = nullptr;
            if (m.layout == 2) hipLaunchKernelGGL(bicgf32_p<true>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.p, (const float4*)m.r, (const float4*)m.v, n4);
            else hipLaunchKernelGGL(bicgf32_p<false>, dim3(nbk), dim3(UB), 0, c->stream, a, (float4*)m.p, (const float4*)m.r, (const float4*)m.v, n4);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + B_RHO, (B_END - B_RHO) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        it = (int)c->h_scal[B_ITERS - B_RHO];
        done = c->h_scal[B_DONE - B_RHO];
    }
    *iters = it;
    if (full_stop) *full_stop = (done == 1.0 && c->h_scal[B_HALF - B_RHO] == 0.0) ? 1 : 0;      // stopped behind a whole iteration: the chain can go on after a reliable update
    if (done == 2.0) { set_error("mixed-precision even-odd BiCGStab: the fp32 chain broke down"); return LQCD_ERR_NOT_CONVERGED; }
    if (done == 1.0 && !cont && first_burst == 0) op->bicg_hint = it;
    return LQCD_OK;      // an inner solve that ran out of iterations still improves x: the outer loop decides
}

int bicgstab_eo_wilson_mixed(lqcd_op_s* op, lqcd_spinor_s& xe, lqcd_spinor_s* rhs, lqcd_spinor_s* const w[6], lqcd_spinor_s* to, int dg, double eps,
                             int maxiter, int* iters, double* final_rr, const double2* Ai) {
    lqcd_ctx_s* c = op->ctx;
    const size_t nh = xe.elems, nfull = 2 * nh;
    // fp32 links and work space: the buffers of the mixed-precision CG (four full-lattice vectors = eight halves), component-pair layout
    Mix32 mm;
    LQCHK(mix_prepare(op, nfull, mm, true));       // layout 2 (site pairs) where stencil_pair32.hip applies, else the component pairs of the fp32 build
    Eo32 m;
    m.layout = mm.layout;
    m.gauge = mm.gauge; m.gauge12 = mm.gauge12; m.gauge16 = mm.gauge16;
    if (Ai) {      // Wilson-clover (layout 1: the site-pair kernel carries no clover term): the inverse blocks in fp32, rebuilt with the inverse
        const size_t nc = clover_elems(c->geom);
        LQCHK(mix_alloc(c, 6, nc * sizeof(float2)));      // the slot of the fp32 clover blocks of the mixed-precision CG (converted per solve there as well)
        hipLaunchKernelGGL(cvt_to_f32, dim3(stream_grid(c, nc)), dim3(MB), 0, c->stream, (float2*)c->mix_buf[6], Ai, nc, 1.0);
        HIPCHK(hipGetLastError());
        m.ainv = (const float2*)c->mix_buf[6];
    }
    m.x = mm.x; m.r = mm.x + nh; m.r0 = mm.r; m.p = mm.r + nh; m.v = mm.p; m.s = mm.p + nh; m.t = mm.t; m.to = mm.t + nh;
    lqcd_spinor_s *r = w[0], *q = w[1];
    const int nb = stream_grid(c, nh);
    auto true_residual = [&](double* rr) -> int {      // r = rhs - M x in fp64
        LQCHK(schur_wilson(op, q, &xe, to, dg, Ai));
        hipLaunchKernelGGL(residual_kernel, dim3(nb), dim3(MB), 0, c->stream, r->data, rhs->data, q->data, (const double2*)nullptr, 0.0, nh, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_RED0, true, 0));
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        *rr = c->h_scal[0];
        return LQCD_OK;
    };
    double rr = 0, xx = 0;
    {      // |x|^2 and |rhs|^2 in one launch and one read-back (a caller that has just cleared x says so: |rhs|^2 twice, x is not read)
        hipLaunchKernelGGL(norm2_two_kernel, dim3(nb), dim3(MB), 0, c->stream, (const double2*)(c->zero_guess_hint ? rhs->data : xe.data), (const double2*)rhs->data, nh, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 2, S_RED0, true, 0));
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        xx = c->zero_guess_hint ? 0.0 : c->h_scal[0]; rr = c->h_scal[1];
    }
    const double2* rsrc = r->data;      // the fp64 residual the next correction step converts
    if (xx == 0.0) rsrc = rhs->data;    // zero guess (what the action solves pass): r = rhs, no Schur application, no copy
    else LQCHK(true_residual(&rr));
    int total = 0, outer = 0, chain_it = 0, live = 0;
    double scale0 = 1.0;
    // site pairs: the true residual behind a correction step is written straight into the fp32 fields of the next one (pair32_residual), scaled by an ESTIMATE of 1 / |r| --
    // the chain's stopping threshold carries the exact norm, the recurrences do not care about the scale
    bool have32 = false, have_full = false;      // have_full: r0 = p = r and x = 0 were written as well
    double scale_have = 1.0;
    // digits one correction step can gain: six with fp32 links; the int16 links of mixed_links16 differ from the true ones by 1.5e-5 per real, which the
    // inverse amplifies -- four and a half
    const double digits = (m.layout == 2 && m.gauge16) ? 4.5 : 6.0;
    while (rr >= eps && outer < 12 && total < maxiter) {
        // as few correction steps as the inner recurrence supports, the reduction split evenly
        const double togo = std::sqrt(eps / rr) * 0.5;
        const int nsteps = std::max(1, (int)std::ceil(std::log10(1.0 / std::min(togo, 0.1)) / digits));
        const double tol = std::max(std::pow(togo, 1.0 / nsteps), 2e-7);
        // reliable update (bicg_reliable): the chain that stopped behind a whole iteration goes on -- same r0, p, v and scalars, same units (scale0) -- with the true residual
        // in place of the recursive one; otherwise a new chain on the normalised residual
        const bool cont = live && c->tun.bicg_reliable;
        if (!cont) { scale0 = have32 ? scale_have : 1.0 / std::sqrt(rr); chain_it = 0; }
        const bool pre_init = m.layout == 2 && !cont && (!have32 || have_full);      // a new chain on site pairs: r, r0 = r, p = r and x = 0 in the one pass of the conversion
        if (have32) {}      // (done behind the last step)
        else if (m.layout == 2) LQCHK(pair32_cvt_spinor(c, m.r, rsrc, scale0, 1, pre_init ? m.r0 : nullptr, pre_init ? m.p : nullptr, pre_init ? m.x : nullptr));
        else LQCHK(to_f32(c, 1, m.r, rsrc, nh, scale0));
        have32 = false;
        rsrc = r->data;
        int it = chain_it;
        const int hint = op->bicg32_hint[dg ? 1 : 0][std::min(outer, 3)];
        LQCHK(inner_bicgstab_eo32(op, m, nh, dg, tol * tol * rr * scale0 * scale0, chain_it + (maxiter - total), &it, cont, hint, &live, pre_init, rr * scale0 * scale0));
        const int step_its = it - chain_it;
        total += step_its;
        chain_it = it;
        op->bicg32_hint[dg ? 1 : 0][std::min(outer, 3)] = step_its;
        if (m.layout == 2) LQCHK(pair32_axpy_to_f64(c, xe.data, m.x, 1.0 / scale0, 1));
        else LQCHK(add_from_f32(c, 1, xe.data, m.x, 1.0 / scale0, nh));
        double rrn = 0;
        if (m.layout == 2) {
            const bool goes_on = live && c->tun.bicg_reliable;
            scale_have = goes_on ? scale0 : scale0 / tol;
            have_full = !goes_on && nsteps > 1 && !c->tun.mixed_lean_residual;      // (the step that was to reach eps: the three extra fields are written only if it did not -- by the copies of a new chain)
            int nbp = 0;
            LQCHK(schur_wilson(op, q, &xe, to, dg, Ai));
            LQCHK(pair32_residual(c, m.r, rhs->data, q->data, scale_have, have_full ? m.r0 : nullptr, have_full ? m.p : nullptr, have_full ? m.x : nullptr, &nbp));
            LQCHK(reduce_to_slot(c, nbp, 1, S_RED0, true, 0));
            HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            rrn = c->h_scal[0];
            have32 = true;
        } else LQCHK(true_residual(&rrn));
        outer++;
        static const bool trace = getenv("LQCD_MIXED_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[lqcd] mixed e-o BiCGStab: step %d (%s), inner tolerance %.2e, %d iterations, |r|^2 %.3e -> %.3e (asked %.3e)\n", outer, cont ? "goes on" : "new chain", tol, step_its, rr, rrn, eps);
        const bool stalled = !(rrn < 0.0625 * rr);
        rr = rrn;
        if (stalled) break;
    }
    if (iters) *iters = total;
    if (final_rr) *final_rr = rr;
    if (rr < eps) return LQCD_OK;
    // fp32 accuracy exhausted (or a breakdown of the fp32 recurrence): the fp64 chain finishes from the current iterate
    int it64 = 0;
    const bool hint = c->zero_guess_hint;
    c->zero_guess_hint = false;      // (x is the current iterate now, whatever the caller said about its guess)
    const int st = bicgstab_eo_wilson(op, xe, rhs, w, to, dg, eps, std::max(1, maxiter - total), &it64, final_rr, Ai);
    c->zero_guess_hint = hint;
    if (iters) *iters = total + it64;
    return st;
}

// fp32 multi-shift CG: (A + sigma_j) e_j = rhs, j < ns, and A e = rhs if xbase is given; rhs in m.r with |rhs|^2 = 1, zero guesses.
// The loop of inner_cg32 with the zeta recurrences (solvers.hip ms_zeta, double precision scalars) and one fused update pass.  A shift
// is frozen once zeta_j^2 |r|^2 < eps2; without xbase the solve ends when every shift is frozen, with it when |r|^2 < eps2.
static int inner_ms32(lqcd_op_s* op, const Mix32& m, float2* xbase, const std::vector<float2*>& xj, const std::vector<float2*>& pj,
                      const double* sigma, int ns, char* d_blk, size_t n, double eps2, int maxiter, int* iters, double* rr_out) {
    lqcd_ctx_s* c = op->ctx;
    const size_t bytes = n * sizeof(float2), ms_doubles = 6 * (size_t)ns + 2;
    double* d_ms = (double*)d_blk;
    float4** d_ptr = (float4**)(d_blk + ms_doubles * sizeof(double));
    if (xbase) HIPCHK(hipMemsetAsync(xbase, 0, bytes, c->stream));
    HIPCHK(hipMemcpyAsync(m.p, m.r, bytes, hipMemcpyDeviceToDevice, c->stream));
    std::vector<double> hms(ms_doubles, 1.0);    // zeta_{-1} = zeta_0 = 1, alpha_{-1} = 1
    std::vector<float4*> hptr(2 * (size_t)ns);
    for (int j = 0; j < ns; j++) {
        HIPCHK(hipMemsetAsync(xj[j], 0, bytes, c->stream));
        HIPCHK(hipMemcpyAsync(pj[j], m.r, bytes, hipMemcpyDeviceToDevice, c->stream));
        hms[j] = sigma[j];
        hptr[j] = (float4*)xj[j];
        hptr[ns + j] = (float4*)pj[j];
    }
    hms[6 * (size_t)ns + 1] = 0.0;               // beta_{-1} = 0
    HIPCHK(hipMemcpyAsync(d_ms, hms.data(), ms_doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (ns) HIPCHK(hipMemcpyAsync(d_ptr, hptr.data(), 2 * (size_t)ns * sizeof(float4*), hipMemcpyHostToDevice, c->stream));
    double init[9] = {1.0, 0, 0, 0, 0, 0, eps2, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));     // hms / hptr are stack-owned host buffers
    const int nbs = stencil_num_partials(c, op->kind, op->r, 2, m.layout == 2 ? 2 : 1, op->csw != 0.0 && op->clover != nullptr), nbu = stream_grid(c, n / 2), check_every = 8;
    int it = 0;
    double rr = 1.0;
    bool done = false;
    while (!done && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            apply_bc(c, op->bc);
            StencilCall s1 = call32(op, m, m.t, m.p, 0);
            s1.norm_partial = c->d_partial;
            s1.skip_flag = c->d_scal;
            LQCHK(stencil_apply(c, s1));
            LQCHK(reduce_to_slot(c, nbs, 1, S_PQ, true, 1));
            StencilCall s2 = call32(op, m, m.t, m.t, 1);
            s2.norm_partial = c->d_partial;
            s2.upd_scal = c->d_scal;
            s2.upd[0] = (double2*)m.r;
            s2.upd[1] = (double2*)(m.r + m.blk);
            LQCHK(stencil_apply(c, s2));
            LQCHK(reduce_to_slot(c, nbs, 1, S_RRNEW, true, 2));
            if (ns) LQCHK(ms_zeta_launch(c, d_ms, ns, xbase ? 0 : 1));
            hipLaunchKernelGGL(ms32_update_all, dim3(nbu), dim3(MB), 0, c->stream, c->d_scal, d_ms, d_ptr, (float4*)xbase, (float4*)m.p,
                               (const float4*)m.r, n / 2, ns);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        it = (int)c->h_scal[S_ITERS - S_RR];
        done = c->h_scal[S_DONE - S_RR] != 0.0;
        if (!std::isfinite(rr)) break;
    }
    *iters = it;
    *rr_out = rr;
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

// out = D in (dagger: D^+ in) through the fp32 operator of the inner solver -- the fp32 copies of the links, the kernel and the field layout a
// mixed-precision solve on this operator would use (tunable mixed_pair32) -- converted back to fp64.  Diagnostic / timing entry point (no
// reference counterpart): parity tests of the fp32 kernels against the CPU restatement at fp32 accuracy, and their time per application
// (reps > 0: mean over reps applications between HIP events on the library's stream; 0: one application, no timing).
extern "C" int lqcd_op_apply_f32(lqcd_op_t op, lqcd_spinor_t out, lqcd_spinor_t in, int dagger, int reps, double* ms) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op && out && in && out->ctx == op->ctx && in->ctx == op->ctx && out->kind == op->kind && in->kind == op->kind && out->subset == LQCD_FULL &&
               in->subset == LQCD_FULL && out != in && reps >= 0,
           "lqcd_op_apply_f32: need two distinct FULL spinors of the operator's kind on the operator's context");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t n = in->elems;
    VariantPin pin(c);
    Mix32 m;
    LQCHK(mix_prepare(op, n, m));
    LQCHK(to_f32(c, m.layout, m.p, in->data, n, 1.0));
    apply_bc(c, op->bc);
    StencilCall s1 = call32(op, m, m.t, m.p, dagger);
    LQCHK(stencil_apply(c, s1));
    if (reps > 0) {
        HIPCHK(hipEventRecord(c->ev_t0, c->stream));
        for (int k = 0; k < reps; k++) LQCHK(stencil_apply(c, s1));
        HIPCHK(hipEventRecord(c->ev_t1, c->stream));
        HIPCHK(hipEventSynchronize(c->ev_t1));
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, c->ev_t0, c->ev_t1));
        if (ms) *ms = (double)t / reps;
    }
    HIPCHK(hipMemsetAsync(out->data, 0, n * sizeof(double2), c->stream));
    LQCHK(add_from_f32(c, m.layout, out->data, m.t, 1.0, n));
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

// Mixed-precision CG for D^+D x = b.  x holds the initial guess.  eps: absolute bound on the TRUE squared residual (the
// reference's rule real(r.r) < eps, evaluated in fp64); inner_tol: relative residual norm requested from each fp32 solve
// (<= 0 [default of the bindings]: chosen per step, see below).  iters = total fp32 iterations (+ fp64 iterations of the fall-back, if it ran); outer = defect-correction steps.
extern "C" int lqcd_solve_mixed_cg_DdagD(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, double eps, int maxiter, double inner_tol, int* iters,
                                         int* outer, double* final_rr) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op && x && b && x->ctx == op->ctx && b->ctx == op->ctx && x->kind == op->kind && b->kind == op->kind && x->subset == LQCD_FULL &&
               b->subset == LQCD_FULL && x != b && maxiter >= 0,
           "lqcd_solve_mixed_cg_DdagD: need two distinct FULL spinors of the operator's kind on the operator's context");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    const bool adaptive = inner_tol <= 0.0;       // default: the fewest defect-correction steps fp32 allows, the required reduction split evenly over them
    const size_t n = x->elems;
    VariantPin pin(c);
    Mix32 m;
    LQCHK(mix_prepare(op, n, m));
    lqcd_spinor_s* r = scratch_get(c, x->kind, LQCD_FULL);
    lqcd_spinor_s* q = scratch_get(c, x->kind, LQCD_FULL);
    lqcd_spinor_s* t = scratch_get(c, x->kind, LQCD_FULL);
    auto release = [&]() { scratch_put(r); scratch_put(q); scratch_put(t); };
    if (!(r && q && t)) { release(); return LQCD_ERR_HIP; }
    int total = 0, nout = 0, st = LQCD_OK;
    double rr = 0;
    auto true_residual = [&]() -> int {     // r = b - D^+D x, rr = |r|^2 (all ranks)
        LQCHK(op_apply_async(op, t, x, 0, nullptr));
        LQCHK(op_apply_async(op, q, t, 1, nullptr));
        const int nb = stream_grid(c, n);
        hipLaunchKernelGGL(residual_kernel, dim3(nb), dim3(MB), 0, c->stream, r->data, b->data, q->data, (const double2*)nullptr, 0.0, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_RED0, true, 0));
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        return LQCD_OK;
    };
    auto run = [&]() -> int {
        // a zero initial guess (what the force and action evaluations pass): r = b without the two fp64 Dslash of a residual evaluation
        double xx = 1.0;
        LQCHK(blas_norm2(c, x->data, n, &xx, true));
        if (xx == 0.0) {
            HIPCHK(hipMemcpyAsync(r->data, b->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
            LQCHK(blas_norm2(c, r->data, n, &rr, true));
        } else {
            LQCHK(true_residual());
        }
        bool fallback = false;
        while (rr >= eps && total < maxiter) {
            if (!std::isfinite(rr)) { set_error("mixed CG: residual is not finite"); return LQCD_ERR_NOT_CONVERGED; }
            const double nrm = std::sqrt(rr);
            LQCHK(to_f32(c, m.layout, m.r, r->data, n, 1.0 / nrm));
            // no tighter than needed to reach eps (with a 10x margin), no tighter than fp32 can deliver.  Every defect-correction step costs a true
            // residual (two fp64 Dslash + conversions ~ 1.2 ms at 32^3x64) and a restart of the Krylov space, so the default asks for as few as an
            // fp32 recurrence can support -- one step per 1e-12 of |r|^2 still to go (1e-6 in norm) -- and splits what is left evenly over them
            // (32^3x64 to 1e-16: 2 steps / 77 fp32 iterations instead of 3 / 79 with a fixed 1e-4: 52.1 -> 49.3 ms, profiles/r03_mixed_precision.log)
            const double need = 0.1 * eps / rr;      // < 1 here
            double eps2;
            if (adaptive) {
                const double floor2 = 1e-12;
                const int steps = std::max(1, (int)std::ceil(std::log(eps / rr) / std::log(floor2) - 1e-9));      // counted without the margin
                eps2 = std::max(floor2, std::pow(need, 1.0 / steps));
            } else {
                eps2 = std::max(inner_tol * inner_tol, need);
            }
            int it = 0;
            double rin = 0;
            LQCHK(inner_cg32(op, m, n, eps2, maxiter - total, &it, &rin));
            total += it;
            nout++;
            LQCHK(add_from_f32(c, m.layout, x->data, m.x, nrm, n));
            const double rr_old = rr;
            LQCHK(true_residual());
            if (!(rr < 0.5 * rr_old)) { fallback = rr >= eps; break; }   // fp32 accuracy exhausted
        }
        if (fallback && total < maxiter) {
            int it64 = 0;
            const int s64 = cg_run(op, x, b, eps, maxiter - total, false, &it64, nullptr);
            total += it64;
            if (s64 != LQCD_OK && s64 != LQCD_ERR_NOT_CONVERGED) return s64;
            LQCHK(true_residual());
        }
        return LQCD_OK;
    };
    st = run();
    release();
    if (iters) *iters = total;
    if (outer) *outer = nout;
    if (final_rr) *final_rr = rr;
    if (st != LQCD_OK) return st;
    if (!(rr < eps)) {
        set_error("The mixed-precision CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// Mixed-precision multi-shift CG: (D^+D + sigma_j) xs[j] = b for j < ns and (x0 != NULL) D^+D x0 = b.  Same contract as
// lqcd_solve_multishift_cg (zero initial guesses, shiftedcg of the RHMC path, README.md:132) with the stopping rule enforced on the TRUE
// fp64 residual of EVERY system: on return |b - (D^+D + sigma_j) xs[j]|^2 < eps for all j.
//   phase 1: one fp32 multi-shift CG on b/|b| -- one Krylov space for all shifts, half the bytes per iteration -- to the relative
//            residual inner_tol (<= 0: 1e-6; or to eps with a 10x margin, whichever is looser) on every system;
//   phase 2: a multi-shift recurrence cannot be restarted (the shifted residuals stop being collinear), so each system is finished on
//            its own by fp64 defect correction: r_j = b - (A + sigma_j) x_j in fp64, (A + sigma_j) e = r_j/|r_j| by the fp32 solver
//            (a one-shift multi-shift CG: the Krylov space of A, stopped when the shifted residual is below its target),
//            x_j += |r_j| e, until |r_j|^2 < eps.  A correction that fails to halve the residual is redone in fp64.
// Worth it when the target is loose enough for phase 1 to do most of the work (MD-force tolerances); at 1e-10 relative and tighter
// the per-shift corrections cost about what the shared Krylov space saved (LABNOTES.md).  iters: fp32 iterations of phase 1 + all
// corrections (+ fp64 iterations of fall-backs); outer: number of fp32 correction solves; final_rr: the largest true residual.
extern "C" int lqcd_solve_multishift_mixed_cg(lqcd_op_t op, lqcd_spinor_t x0, lqcd_spinor_t* xs, lqcd_spinor_t b, const double* sigma, int ns,
                                              double eps, int maxiter, double inner_tol, int* iters, int* outer, double* final_rr) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op && b && ns >= 0 && ns <= 1024 && (ns == 0 || (xs && sigma)) && maxiter >= 0,
           "lqcd_solve_multishift_mixed_cg: null argument or more than 1024 shifts");
    ARGCHK(b->ctx == op->ctx && b->kind == op->kind && b->subset == LQCD_FULL, "lqcd_solve_multishift_mixed_cg: b must be a FULL spinor of the operator");
    for (int j = 0; j < ns; j++) {
        ARGCHK(xs[j] && xs[j]->ctx == op->ctx && xs[j]->kind == op->kind && xs[j]->subset == LQCD_FULL && xs[j] != b,
               "lqcd_solve_multishift_mixed_cg: xs[j] must be distinct FULL spinors of the operator");
        ARGCHK(sigma[j] >= 0.0, "lqcd_solve_multishift_mixed_cg: shifts must be non-negative");
        ARGCHK(xs[j] != x0, "lqcd_solve_multishift_mixed_cg: xs[j] and x0 must be different fields (every system is updated in place on its own handle)");
        for (int i = 0; i < j; i++) ARGCHK(xs[i] != xs[j], "lqcd_solve_multishift_mixed_cg: the xs[j] must be pairwise different fields");
    }
    if (x0) LQCHK(check_full(op, x0, b, "lqcd_solve_multishift_mixed_cg"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (inner_tol <= 0.0) inner_tol = 1e-6;     // about as far as an fp32 recurrence follows the true residual
    const size_t n = b->elems, bytes64 = n * sizeof(double2);
    VariantPin pin(c);
    Mix32 m;
    LQCHK(mix_prepare(op, n, m));
    // fp32 pool: x_j, p_j for every shift (+ the base solution), and the coefficient block of the recurrences
    const size_t ms_bytes = (6 * (size_t)ns + 2) * sizeof(double) + 2 * (size_t)std::max(ns, 1) * sizeof(float4*);
    LQCHK(mix_alloc(c, 7, (2 * (size_t)ns + 1) * n * sizeof(float2) + ms_bytes + 256));
    char* pool = (char*)c->mix_buf[7];
    std::vector<float2*> xj(ns), pj(ns);
    for (int j = 0; j < ns; j++) { xj[j] = (float2*)pool + (size_t)(2 * j) * n; pj[j] = (float2*)pool + (size_t)(2 * j + 1) * n; }
    float2* xb32 = (float2*)pool + (size_t)(2 * ns) * n;
    char* d_blk = pool + (2 * (size_t)ns + 1) * n * sizeof(float2);
    d_blk += (256 - ((size_t)d_blk & 255)) & 255;
    ScratchScope sc(c);
    lqcd_spinor_s* r = sc.get(op->kind, LQCD_FULL);
    lqcd_spinor_s* q = sc.get(op->kind, LQCD_FULL);
    lqcd_spinor_s* t = sc.get(op->kind, LQCD_FULL);
    lqcd_spinor_s* e64 = nullptr;       // fp64 correction of a fall-back (allocated on first use)
    if (!(r && q && t)) { set_error("lqcd_solve_multishift_mixed_cg: out of device memory"); return LQCD_ERR_HIP; }
    int total = 0, nout = 0;
    double worst = 0.0, rr = 0.0;
    auto true_residual = [&](lqcd_spinor_s* x, double sg) -> int {     // r = b - (D^+D + sg) x, rr = |r|^2 (all ranks)
        LQCHK(op_apply_async(op, t, x, 0, nullptr));
        LQCHK(op_apply_async(op, q, t, 1, nullptr));
        const int nb = stream_grid(c, n);
        hipLaunchKernelGGL(residual_kernel, dim3(nb), dim3(MB), 0, c->stream, r->data, b->data, q->data, (const double2*)x->data, sg, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_RED0, true, 0));
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RED0, sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        return LQCD_OK;
    };
    auto run = [&]() -> int {
        double bb = 0.0;
        LQCHK(blas_norm2(c, b->data, n, &bb, true));
        if (x0) HIPCHK(hipMemsetAsync(x0->data, 0, bytes64, c->stream));
        for (int j = 0; j < ns; j++) HIPCHK(hipMemsetAsync(xs[j]->data, 0, bytes64, c->stream));
        if (!(bb >= eps)) { worst = bb; return LQCD_OK; }            // the zero vectors already satisfy the stopping rule
        if (!std::isfinite(bb)) { set_error("mixed multi-shift CG: the right-hand side is not finite"); return LQCD_ERR_NOT_CONVERGED; }
        // ---- phase 1
        const double nb_ = std::sqrt(bb);
        LQCHK(to_f32(c, m.layout, m.r, b->data, n, 1.0 / nb_));
        int it = 0;
        double rin = 0;
        LQCHK(inner_ms32(op, m, x0 ? xb32 : nullptr, xj, pj, sigma, ns, d_blk, n, std::max(inner_tol * inner_tol, 0.1 * eps / bb), maxiter, &it, &rin));
        total += it;
        if (x0) LQCHK(add_from_f32(c, m.layout, x0->data, xb32, nb_, n));
        for (int j = 0; j < ns; j++) LQCHK(add_from_f32(c, m.layout, xs[j]->data, xj[j], nb_, n));
        // ---- phase 2: every system on its own
        for (int j = (x0 ? -1 : 0); j < ns; j++) {
            lqcd_spinor_s* x = j < 0 ? x0 : xs[j];
            const double sg = j < 0 ? 0.0 : sigma[j];
            LQCHK(true_residual(x, sg));
            while (rr >= eps && total < maxiter) {
                if (!std::isfinite(rr)) { set_error("mixed multi-shift CG: residual is not finite"); return LQCD_ERR_NOT_CONVERGED; }
                const double nrm = std::sqrt(rr), rr_old = rr;
                LQCHK(to_f32(c, m.layout, m.r, r->data, n, 1.0 / nrm));
                const double eps2 = std::max(inner_tol * inner_tol, 0.1 * eps / rr);
                it = 0;
                const std::vector<float2*> x1(1, xj.empty() ? xb32 : xj[0]), p1(1, pj.empty() ? xb32 : pj[0]);
                if (j < 0) LQCHK(inner_ms32(op, m, xb32, xj, pj, sigma, 0, d_blk, n, eps2, maxiter - total, &it, &rin));
                else LQCHK(inner_ms32(op, m, nullptr, x1, p1, &sg, 1, d_blk, n, eps2, maxiter - total, &it, &rin));
                total += it;
                nout++;
                LQCHK(add_from_f32(c, m.layout, x->data, j < 0 ? xb32 : x1[0], nrm, n));
                LQCHK(true_residual(x, sg));
                if (rr < 0.5 * rr_old) continue;
                if (rr < eps) break;
                // fp32 accuracy exhausted: the correction (A + sg) e = r in fp64 (zero guess), x += e
                if (!e64) e64 = sc.get(op->kind, LQCD_FULL);
                if (!e64) { set_error("lqcd_solve_multishift_mixed_cg: out of device memory"); return LQCD_ERR_HIP; }
                int it64 = 0;
                double r64 = 0;
                lqcd_spinor_s* rhs = q;      // r is overwritten by nothing in the fp64 solver, but keep an own copy of the right-hand side
                HIPCHK(hipMemcpyAsync(rhs->data, r->data, bytes64, hipMemcpyDeviceToDevice, c->stream));
                lqcd_spinor_t ex[1] = {e64};
                const int s64 = j < 0 ? lqcd_solve_multishift_cg(op, e64, nullptr, rhs, nullptr, 0, eps, maxiter - total, &it64, &r64)
                                      : lqcd_solve_multishift_cg(op, nullptr, ex, rhs, &sg, 1, eps, maxiter - total, &it64, &r64);
                total += it64;
                if (s64 != LQCD_OK && s64 != LQCD_ERR_NOT_CONVERGED) return s64;
                LQCHK(blas_axpy(c, 1.0, 0.0, e64->data, x->data, n));
                LQCHK(true_residual(x, sg));
                break;
            }
            worst = std::max(worst, rr);
        }
        return LQCD_OK;
    };
    const int st = run();
    (void)hipStreamSynchronize(c->stream);
    if (iters) *iters = total;
    if (outer) *outer = nout;
    if (final_rr) *final_rr = worst;
    if (st != LQCD_OK) return st;
    if (!(worst < eps)) {
        set_error("The mixed-precision shifted CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(worst));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}
