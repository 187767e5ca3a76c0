// solvers.hip -- Krylov solvers with device-resident scalars: CG on D^+D (fused iteration), BiCGStab and its even-odd (Schur) form,
// the parity-block CG and the multi-shift CG of the RHMC path.
//
// Replaces, behind the C ABI, LatticeDiracOperators.jl's solve_DinvX! (cg / bicgstab / even-odd bicgstab / shiftedcg) -- SURVEY.md
// 8(a) a4-a5, 8(f) rank 3; reference call sites /root/reference/src/md/AbstractMD.jl:129, src/updates/standardHMC.jl:71.
#include "ops_internal.h"
#include "stencil_common.h"      // HArgs, wilson_pack_axpy_block: pack blocks appended to the x/p update launch (partitioned lattices)

#include <algorithm>
#include <cstring>
#include <cmath>
#include <complex>
#include <functional>

namespace lqcd {

// ---------------------------------------------------------------------------------- CG with device-resident scalars

__global__ void cg_scalar_alpha(double* s) {
    if (s[S_DONE] != 0.0) { s[S_XDONE] = 1.0; return; }   // the iterate of the converging iteration has been written
    s[S_ALPHA] = s[S_RR] / s[S_PQ];
}
__global__ void cg_scalar_beta(double* s) {
    if (s[S_DONE] != 0.0) return;
    const double rrn = s[S_RRNEW];
    s[S_BETA] = rrn / s[S_RR];
    s[S_RR] = rrn;
    s[S_ITERS] += 1.0;
    if (rrn < s[S_EPS]) s[S_DONE] = 1.0;
}

// x += alpha p ; r -= alpha q ; partial |r|^2
__global__ __launch_bounds__(UB) void cg_update_xr(const double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ r,
                                                    const double2* __restrict__ p, const double2* __restrict__ q, size_t n,
                                                    double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    if (s[S_DONE] == 0.0) {
        const double al = s[S_ALPHA];
        for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
            const double2 pv = p[i], qv = q[i];
            double2 xv = x[i], rv = r[i];
            xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
            rv.x = fma(-al, qv.x, rv.x); rv.y = fma(-al, qv.y, rv.y);
            x[i] = xv; r[i] = rv;
            acc = fma(rv.x, rv.x, acc); acc = fma(rv.y, rv.y, acc);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// fused tail of an iteration:  x += alpha p ;  p = r + beta p   (x is still updated in the iteration that converges,
// p only while the solve continues; nothing is touched in later, overshooting launches)
__global__ __launch_bounds__(UB) void cg_update_xp(const double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ p,
                                                    const double2* __restrict__ r, size_t n) {
    if (s[S_XDONE] != 0.0) return;
    const double al = s[S_ALPHA], be = s[S_BETA];
    const bool cont = s[S_DONE] == 0.0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        double2 pv = p[i], xv = x[i];
        xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
        x[i] = xv;
        if (cont) {
            const double2 rv = r[i];
            pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
            p[i] = pv;
        }
    }
}
// cg_small: the same tail with the second reduction folded in.  Every wave sums the block partials of |r|^2 the update-mode stencil left
// (reduce_final's order), forms beta = rr' / rr (rr = the value this iteration started from, S_RROLD: nobody writes it here) and the
// convergence test itself; block 0 records rr', beta, the iteration count and the done flag for the kernels behind this one.
__global__ __launch_bounds__(UB) void cg_update_xp_small(double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ p,
                                                          const double2* __restrict__ r, size_t n, const double* __restrict__ part, int nb) {
    if (s[S_XDONE] != 0.0) return;
    const double rrn = sum_partials_small(part, nb);
    const double al = s[S_ALPHA], be = rrn / s[S_RROLD];
    const bool cont = !(rrn < s[S_EPS]);
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        double2 pv = p[i], xv = x[i];
        xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
        x[i] = xv;
        if (cont) {
            const double2 rv = r[i];
            pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
            p[i] = pv;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        s[S_RRNEW] = rrn;
        s[S_BETA] = be;
        s[S_RR] = rrn;
        s[S_ITERS] += 1.0;
        if (!cont) s[S_DONE] = 1.0;
    }
}
typedef double v2dd __attribute__((ext_vector_type(2)));
template <bool NT> __device__ inline double2 ldx(const double2* p) {
    if constexpr (NT) { v2dd v = __builtin_nontemporal_load(reinterpret_cast<const v2dd*>(p)); double2 r; r.x = v.x; r.y = v.y; return r; }
    else return *p;
}
template <bool NT> __device__ inline void stx(double2* p, double2 v) {
    if constexpr (NT) { v2dd t = {v.x, v.y}; __builtin_nontemporal_store(t, reinterpret_cast<v2dd*>(p)); }
    else *p = v;
}
// Deferred x update (cg_defer_x): iteration k even:  p_{k+1} = r + beta p_k into the OTHER buffer, x untouched, alpha_k kept;
//                                    iteration k odd:   x += alpha_{k-1} p_{k-1} + alpha_k p_k (same order of operations as two single
// updates: identical bits), p_{k+1} = r + beta p_k over the dead p_{k-1}.  3 + 6 streams per pair of iterations instead of 5 + 5.  The
// iteration that converges completes x itself; nothing is touched in later, overshooting launches.
// FOLD (several ranks): the scalar step behind the second all-reduce (beta = rr'/rr, rr = rr', iteration count, convergence flag) is done here --
// every block forms beta and the test from the all-reduced rr' (S_RRNEW) and the rr this iteration started from (S_RROLD, written by the
// update-mode stencil), block 0 records the results -- instead of by a one-thread kernel between the all-reduce and this launch.
struct FoldBeta { double be; bool cont; };
template <bool FOLD>
__device__ inline FoldBeta cg_beta(double* s) {
    FoldBeta f;
    if constexpr (FOLD) {
        const double rrn = s[S_RRNEW];
        f.be = rrn / s[S_RROLD];
        f.cont = !(rrn < s[S_EPS]);
    } else {
        f.be = s[S_BETA];
        f.cont = s[S_DONE] == 0.0;
    }
    return f;
}
template <bool FOLD>
__device__ inline void cg_beta_commit(double* s, const FoldBeta& f, bool first_block) {
    if constexpr (FOLD) {
        if (first_block && threadIdx.x == 0) {
            const double rrn = s[S_RRNEW];
            s[S_BETA] = f.be;
            s[S_RR] = rrn;
            s[S_ITERS] += 1.0;
            if (!f.cont) s[S_DONE] = 1.0;
        }
    }
}
// PACK (partitioned lattice, halo_fuse bit 1): the FIRST npk blocks of the launch are pack blocks (dispatched first: the faces are what the next
// launch on the critical path waits for) -- they form p' = r + beta p at the face sites in registers and write the send buffers of the next
// D p (stencil_common.h wilson_pack_axpy_block), so that application needs no pack launch; the other nbf blocks do the flat update.
template <bool NT, bool FOLD, bool PACK = false>      // NT: streaming (non-temporal) loads and stores for fields that are not re-used before they fall out of every cache
__global__ __launch_bounds__(UB) void cg_update_even(double* __restrict__ s, double2* __restrict__ x, const double2* __restrict__ pk,
                                                      double2* __restrict__ pnext, const double2* __restrict__ r, size_t n, int nbf, HArgs h, int npx) {
    if (s[S_XDONE] != 0.0) return;
    const FoldBeta fb = cg_beta<FOLD>(s);
    const double al = s[S_ALPHA], be = fb.be;
    const bool cont = fb.cont;
    const int npk = PACK ? 8 * npx : 0, fb0 = (int)blockIdx.x - npk;      // fb0: index among the flat blocks
    if constexpr (PACK) {
        if (fb0 < 0) {
            if (cont) wilson_pack_axpy_block(h, (int)blockIdx.x, npx, be);
            return;
        }
    }
    if (cont) {
        for (size_t i = (size_t)fb0 * UB + threadIdx.x; i < n; i += (size_t)nbf * UB) {
            const double2 pv = ldx<NT>(pk + i), rv = ldx<NT>(r + i);
            double2 o;
            o.x = fma(be, pv.x, rv.x); o.y = fma(be, pv.y, rv.y);
            stx<NT>(pnext + i, o);
        }
        if (fb0 == 0 && threadIdx.x == 0) s[S_APREV] = al;
    } else {
        for (size_t i = (size_t)fb0 * UB + threadIdx.x; i < n; i += (size_t)nbf * UB) {
            const double2 pv = pk[i];
            double2 xv = x[i];
            xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
            x[i] = xv;
        }
    }
    cg_beta_commit<FOLD>(s, fb, fb0 == 0);
}
template <bool NT, bool FOLD, bool PACK = false>
__global__ __launch_bounds__(UB) void cg_update_odd(double* __restrict__ s, double2* __restrict__ x, double2* __restrict__ pprev,
                                                     const double2* __restrict__ pk, const double2* __restrict__ r, size_t n, int nbf, HArgs h, int npx) {
    if (s[S_XDONE] != 0.0) return;
    const FoldBeta fb = cg_beta<FOLD>(s);
    const double ap = s[S_APREV], al = s[S_ALPHA], be = fb.be;
    const bool cont = fb.cont;
    const int npk = PACK ? 8 * npx : 0, fb0 = (int)blockIdx.x - npk;
    if constexpr (PACK) {
        if (fb0 < 0) {
            if (cont) wilson_pack_axpy_block(h, (int)blockIdx.x, npx, be);
            return;
        }
    }
    for (size_t i = (size_t)fb0 * UB + threadIdx.x; i < n; i += (size_t)nbf * UB) {
        const double2 pp = ldx<NT>(pprev + i), pv = ldx<NT>(pk + i);
        double2 xv = ldx<NT>(x + i);
        xv.x = fma(ap, pp.x, xv.x); xv.y = fma(ap, pp.y, xv.y);
        xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
        stx<NT>(x + i, xv);
        if (cont) {
            const double2 rv = ldx<NT>(r + i);
            double2 o;
            o.x = fma(be, pv.x, rv.x); o.y = fma(be, pv.y, rv.y);
            stx<NT>(pprev + i, o);
        }
    }
    cg_beta_commit<FOLD>(s, fb, fb0 == 0);
}
// x += alpha_k p_k for a window that ended (unconverged) on an even iteration
__global__ __launch_bounds__(UB) void cg_flush_kernel(const double* __restrict__ s, double2* __restrict__ x, const double2* __restrict__ pk, size_t n) {
    if (s[S_DONE] != 0.0) return;
    const double ap = s[S_APREV];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 pv = pk[i];
        double2 xv = x[i];
        xv.x = fma(ap, pv.x, xv.x); xv.y = fma(ap, pv.y, xv.y);
        x[i] = xv;
    }
}
// Deferred x update over a ring of K search-direction buffers (cg_defer_x = K >= 3): p_k lives in buffer k % K.  Iteration k with m = k % K < K - 1:
// p_{k+1} = r + beta p_k into the next buffer, alpha_k kept in S_AHIST[m], x untouched -- 3 streams.  m = K - 1 (or the iteration that converges, whatever m):
// x += alpha_{k-m} p_{k-m} + ... + alpha_k p_k in that order (the operations two single updates would do: identical bits), then p_{k+1} over the oldest buffer,
// which this thread has just read -- K + 4 streams.  (4 K + 1) / K passes per iteration instead of the 4.5 of the two-buffer form.
struct CgRing { const double2* b[8]; };      // the buffers that hold pending directions, oldest first (entries >= m unused); indexed by unrolled constants only
template <bool NT, bool FOLD, bool PACK = false>
__global__ __launch_bounds__(UB) void cg_update_ring(double* __restrict__ s, double2* __restrict__ x, CgRing ring, int m, int flush, const double2* __restrict__ pk,
                                                      double2* pnext, const double2* __restrict__ r, size_t n, int nbf, HArgs h, int npx) {
    if (s[S_XDONE] != 0.0) return;
    const FoldBeta fb = cg_beta<FOLD>(s);
    const double al = s[S_ALPHA], be = fb.be;
    const bool cont = fb.cont;
    const int npk = PACK ? 8 * npx : 0, fb0 = (int)blockIdx.x - npk;
    if constexpr (PACK) {
        if (fb0 < 0) {
            if (cont) wilson_pack_axpy_block(h, (int)blockIdx.x, npx, be);
            return;
        }
    }
    if (flush || !cont) {
        double ah[8];
#pragma unroll
        for (int j = 0; j < 8; j++) ah[j] = j < m ? s[S_AHIST + j] : 0.0;
        for (size_t i = (size_t)fb0 * UB + threadIdx.x; i < n; i += (size_t)nbf * UB) {
            double2 xv = ldx<NT>(x + i);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (j < m) {
                    const double2 pj = ldx<NT>(ring.b[j] + i);
                    xv.x = fma(ah[j], pj.x, xv.x); xv.y = fma(ah[j], pj.y, xv.y);
                }
            }
            const double2 pv = ldx<NT>(pk + i);
            xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
            stx<NT>(x + i, xv);
            if (cont) {
                const double2 rv = ldx<NT>(r + i);
                double2 o;
                o.x = fma(be, pv.x, rv.x); o.y = fma(be, pv.y, rv.y);
                stx<NT>(pnext + i, o);
            }
        }
    } else {
        for (size_t i = (size_t)fb0 * UB + threadIdx.x; i < n; i += (size_t)nbf * UB) {
            const double2 pv = ldx<NT>(pk + i), rv = ldx<NT>(r + i);
            double2 o;
            o.x = fma(be, pv.x, rv.x); o.y = fma(be, pv.y, rv.y);
            stx<NT>(pnext + i, o);
        }
        if (fb0 == 0 && threadIdx.x == 0) s[S_AHIST + m] = al;
    }
    cg_beta_commit<FOLD>(s, fb, fb0 == 0);
}
// x += sum of the m pending terms, for a window that ended (unconverged) in the middle of the ring
__global__ __launch_bounds__(UB) void cg_flush_ring(const double* __restrict__ s, double2* __restrict__ x, CgRing ring, int m, size_t n) {
    if (s[S_DONE] != 0.0) return;
    double ah[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ah[j] = j < m ? s[S_AHIST + j] : 0.0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        double2 xv = x[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < m) {
                const double2 pj = ring.b[j][i];
                xv.x = fma(ah[j], pj.x, xv.x); xv.y = fma(ah[j], pj.y, xv.y);
            }
        }
        x[i] = xv;
    }
}
// p = r + beta p
__global__ __launch_bounds__(UB) void cg_update_p(const double* __restrict__ s, double2* __restrict__ p, const double2* __restrict__ r, size_t n) {
    if (s[S_DONE] != 0.0) return;
    const double be = s[S_BETA];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 rv = r[i];
        double2 pv = p[i];
        pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
        p[i] = pv;
    }
}
__global__ __launch_bounds__(UB) void norm2_partial_kernel(const double2* __restrict__ a, size_t n, double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 v = a[i];
        acc = fma(v.x, v.x, acc); acc = fma(v.y, v.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}
// re(p.q) partials (unfused reference form c1 = p.q)
__global__ __launch_bounds__(UB) void redot_partial_kernel(const double2* __restrict__ a, const double2* __restrict__ b, size_t n, double* partial) {
    __shared__ double red[UB / 64];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = a[i], y = b[i];
        acc = fma(x.x, y.x, acc); acc = fma(x.y, y.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < UB / 64; w++) t += red[w];
        partial[blockIdx.x] = t;
    }
}


// cg_small applies when the reductions are local (no communicator), at most 1024 block partials exist, and the stencil kernel in use
// takes alpha from emit()'s argument (every variant but the hop-split ones)
static bool cg_small_ok(lqcd_op_s* op, int nbs) {
    lqcd_ctx_s* c = op->ctx;
    if (any_partitioned(c) || c->has_comm || nbs > 1024) return false;
    const int v = c->tun.dslash_variant;
#ifdef LQCD_VARIANTS
    if (op->kind == LQCD_WILSON && op->r == 1.0 && (v == 2 || v == 3)) return false;
#endif
    (void)v;
    if ((c->tun.dslash_pipe == 1 || c->tun.dslash_pipe == 3) && wilson_pipe_applies(c, op->kind, op->r, 2, op_fused_clover(op))) return false;    // pipelined kernels: no alpha_partials form
    return true;
}

// enqueue one CG iteration on the compute stream (no host synchronisation)
static bool cg_defers_x(lqcd_op_s* op) {
    lqcd_ctx_s* c = op->ctx;
    if (!(c->tun.cg_fused >= 2 && c->tun.cg_defer_x)) return false;
    return !(c->tun.cg_small && cg_small_ok(op, stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op))));
}
int cg_work_get(lqcd_ctx_s* c, int kind, CgWork& w) {
    w.r = scratch_get(c, kind, LQCD_FULL); w.p = scratch_get(c, kind, LQCD_FULL);
    w.q = scratch_get(c, kind, LQCD_FULL); w.tmp = scratch_get(c, kind, LQCD_FULL);
    if (!(w.r && w.p && w.q && w.tmp)) { cg_work_put(w); return LQCD_ERR_HIP; }
    return LQCD_OK;
}
void cg_work_put(CgWork& w) {
    for (lqcd_spinor_s** f : {&w.r, &w.p, &w.q, &w.tmp}) { if (*f) scratch_put(*f); *f = nullptr; }
    for (lqcd_spinor_s*& f : w.more) { if (f) scratch_put(f); f = nullptr; }
}
// buffers of the deferred-x form: 2, or the ring of cg_defer_x = K = 3..8 (cg_setup takes the extra ones from the scratch pool; if the pool cannot grow: 2)
// Measured (profiles/r05_cg_ring.log): at 32^3x64 on one GPU the ring does not pay (K = 2 / 4 / 8: 919.8 / 912.6 / 913.5 iter/s -- the flush kernel's K + 4 streams cost what
// the saved passes win); at the N = 8 local volume, where the vectors live in the Infinity Cache, it does (5207 / 5254 / 5286 iter/s).  cg_defer_x = 1 (default) therefore
// means: two buffers on an unpartitioned lattice, a ring of 8 on a partitioned one; 2 and 3..8 force a size.
static int cg_ring_wanted(lqcd_op_s* op) {
    const int k = op->ctx->tun.cg_defer_x;
    if (op->kind == LQCD_DOMAINWALL) return 2;
    if (k >= 3) return std::min(k, 8);
    return (k == 1 && any_partitioned(op->ctx)) ? 8 : 2;
}
static CgRing cg_ring_pending(const CgWork& w, int m) {      // the buffers of p_{k-m} .. p_{k-1}, oldest first, for k = w.k
    CgRing ring;
    for (int j = 0; j < 8; j++) ring.b[j] = j < m ? w.buf((w.k - m + j) % w.ring)->data : nullptr;
    return ring;
}
int cg_flush_x(lqcd_op_s* op, lqcd_spinor_s* x, CgWork& w) {
    lqcd_ctx_s* c = op->ctx;
    const bool deferred = w.form >= 0 ? w.form == 1 : cg_defers_x(op);     // the form the iterations were enqueued in (recorded by cg_setup)
    if (deferred && w.ring > 2) {
        const int m = w.k % w.ring;       // directions p_{k-m} .. p_{k-1} are still to be added
        if (m == 0) return LQCD_OK;
        hipLaunchKernelGGL(cg_flush_ring, dim3(stream_grid(c, x->elems)), dim3(UB), 0, c->stream, c->d_scal, x->data, cg_ring_pending(w, m), m, x->elems);
        HIPCHK(hipGetLastError());
        w.k += w.ring - m;      // nothing pending any more (a window is never continued after its flush)
        return LQCD_OK;
    }
    if (!deferred || !(w.k & 1)) return LQCD_OK;
    const size_t n = x->elems;
    hipLaunchKernelGGL(cg_flush_kernel, dim3(stream_grid(c, n)), dim3(UB), 0, c->stream, c->d_scal, x->data, w.p->data, n);   // p_k of an even k lives in w.p
    HIPCHK(hipGetLastError());
    w.k++;      // nothing pending any more (a window is never continued after its flush)
    return LQCD_OK;
}

int cg_enqueue_iteration(lqcd_op_s* op, lqcd_spinor_s* x, CgWork& w) {
    lqcd_ctx_s* c = op->ctx;
    const size_t n = x->elems;
    const int nbs_small = stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op));
    // the iteration form was fixed when the solve / session was set up (a tunable changed in between must not split a pending deferred update)
    const int form = w.form >= 0 ? w.form : ((c->tun.cg_fused >= 2 && c->tun.cg_small && cg_small_ok(op, nbs_small)) ? 2 : (cg_defers_x(op) ? 1 : 0));
    if (c->tun.cg_fused >= 2 && form == 2) {
        // small lattices: launch latency is the cost.  Same arithmetic as the fused form below, but the two single-block reductions
        // are folded into the prologues of their consumers -- 3 dependent launches per iteration instead of 5, identical iterates.
        double* part_a = c->d_partial;            // |D p|^2 block partials
        double* part_b = c->d_partial + 2048;     // |r|^2 block partials (the update-mode kernel reads part_a while its blocks write these)
        LQCHK(op_apply_async(op, w.tmp, w.p, 0, part_a, c->tun.cg_skip_done ? c->d_scal : nullptr));
        apply_bc(c, op->bc);
        StencilCall s2;
        LQCHK(make_full_call(op, w.q, w.tmp, 1, s2));
        s2.norm_partial = part_b;
        s2.upd_scal = c->d_scal;
        s2.upd[0] = spinor_block(w.r, 0);
        s2.upd[1] = spinor_block(w.r, 1);
        s2.alpha_partials = part_a; s2.alpha_n = nbs_small; s2.scal_w = c->d_scal;
        LQCHK(stencil_apply(c, s2));
        const int nbu = stream_grid(c, n);
        hipLaunchKernelGGL(cg_update_xp_small, dim3(nbu), dim3(UB), 0, c->stream, c->d_scal, x->data, w.p->data, w.r->data, n, part_b, nbs_small);
        HIPCHK(hipGetLastError());
        return LQCD_OK;
    }
    if (c->tun.cg_fused >= 2) {
        // fully fused form: 10 spinor passes per iteration instead of 13, q = D^+ D p is never written
        //   tmp = D p [+ |tmp|^2 partials] ; alpha = rr / |tmp|^2 ; D^+ tmp with epilogue r -= alpha q [+ |r|^2 partials] ;
        //   beta, convergence ; x += alpha p, p = r + beta p
        const int nbs = stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op));
        const bool defer = form == 1;
        const bool ringed = defer && w.ring > 2;
        lqcd_spinor_s* pk = ringed ? w.buf(w.k % w.ring) : (defer && (w.k & 1)) ? w.q : w.p;        // q = D^+D p is never written in this form: its buffer is the second p
        lqcd_spinor_s* po = ringed ? w.buf((w.k + 1) % w.ring) : (defer && (w.k & 1)) ? w.p : w.q;
        // several ranks: reduce_final -> all-reduce -> one-thread scalar kernel are three dependent launches per reduction; with `fold` the scalar
        // steps move into the prologues of the kernels that consume them (deferred-x form only)
        const bool fold = c->has_comm && defer && c->tun.cg_fold_scalars;
        // partitioned lattice (RCCL path): at small local volumes the iteration is a chain of short dependent launches.  halo_fuse bit 0: the
        // exterior kernel's last block sums the |.|^2 partials (no reduce_final launch); bit 1: the exterior of D p packs the faces of
        // tmp = D p for the D^+ that follows and the x/p update packs the new search direction for the next D p (no pack launches).
        // Wilson r = 1 only (a general-r application is two r = 1 passes with different projectors).
        const bool part = any_partitioned(c) && c->has_comm && c->local_peers.empty();
        const bool folded = part && halo_fold_applies(c, op->kind, op->r, 2, 0, op_fused_clover(op));      // no exterior launch: nothing for bit 0 to ride on
        const bool fr = part && (c->tun.halo_fuse & 1) && !folded;
        const bool fp = part && defer && (c->tun.halo_fuse & 2) && op->kind == LQCD_WILSON && op->r == 1.0;
        {
            apply_bc(c, op->bc);      // another operator of this context (other boundary signs) may have been applied since the last iteration of an open session
            StencilCall s1;
            LQCHK(make_full_call(op, w.tmp, pk, 0, s1));
            s1.norm_partial = c->d_partial;
            s1.skip_flag = c->tun.cg_skip_done ? c->d_scal : nullptr;      // a no-op once the solve has converged inside a burst
            if (fr) s1.red_slot = S_PQ;
            if (fp) { s1.pack_next = 1; s1.prepacked = (w.p_packed && w.pack_epoch == c->halo_epoch) ? 1 : 0; s1.defer_pack = folded ? 1 : 0; }
            LQCHK(stencil_apply(c, s1));
        }
        LQCHK(fr ? reduce_tail(c, 1, S_PQ, fold ? 0 : 1) : reduce_pack_to_slot(c, nbs, S_PQ, fold ? 0 : 1));      // + alpha = rr / pq (+ the pack launch the folded schedule left waiting)
        apply_bc(c, op->bc);
        StencilCall s2;
        LQCHK(make_full_call(op, po, w.tmp, 1, s2));          // update mode writes r only: `out` is a placeholder (the buffer that is dead until the p update)
        s2.norm_partial = c->d_partial;
        if (fold) s2.scal_w = c->d_scal;
        s2.upd_scal = c->d_scal;
        s2.upd[0] = spinor_block(w.r, 0);
        s2.upd[1] = spinor_block(w.r, 1);
        if (fr) s2.red_slot = S_RRNEW;
        if (fp) s2.prepacked = 1;
        LQCHK(stencil_apply(c, s2));
        LQCHK(fr ? reduce_tail(c, 1, S_RRNEW, fold ? 0 : 2) : reduce_to_slot(c, nbs, 1, S_RRNEW, true, fold ? 0 : 2));   // + beta, convergence flag
        const int nbu = stream_grid(c, n);
        HArgs h = HArgs();
        int npx = 0, npack = 0;
        if (fp) {
            int nt = 0;
            for (int mu = 0; mu < 4; mu++)
                if (c->geom.part[mu]) nt = std::max(nt, 2 * face_half_sites(c->geom, mu));
            npx = (nt + UB - 1) / UB; npack = 8 * npx;
            h.g = c->geom;
            h.gauge = op->gauge->data;
            h.parity_mode = 2; h.dagger = 0;
            for (int q = 0; q < 2; q++) { h.in[q] = spinor_block(pk, q); h.upd[q] = spinor_block(w.r, q); }
            for (int mu = 0; mu < 4; mu++) {
                const size_t cnt = (size_t)2 * 6 * face_half_sites(c->geom, mu);      // [send_fwd | send_bwd] back to back (stencil.hip make_hargs)
                h.send_fwd[mu] = halo_send_base(c, mu, 0); h.send_bwd[mu] = halo_send_base(c, mu, 1) + cnt;
            }
        }
        const dim3 ug(nbu + npack), ub(UB);
#define LQ_UPD3(KERN, NT_, FO_, A, B, C) do { \
            if (fp) hipLaunchKernelGGL((KERN<NT_, FO_, true>), ug, ub, 0, c->stream, c->d_scal, x->data, A, B, C, n, nbu, h, npx); \
            else hipLaunchKernelGGL((KERN<NT_, FO_, false>), ug, ub, 0, c->stream, c->d_scal, x->data, A, B, C, n, nbu, h, npx); } while (0)
#define LQ_UPD(KERN, A, B, C) do { \
            if (c->tun.nt_blas) { if (fold) LQ_UPD3(KERN, true, true, A, B, C); else LQ_UPD3(KERN, true, false, A, B, C); } \
            else { if (fold) LQ_UPD3(KERN, false, true, A, B, C); else LQ_UPD3(KERN, false, false, A, B, C); } } while (0)
#define LQ_RING3(NT_, FO_) do { \
            if (fp) hipLaunchKernelGGL((cg_update_ring<NT_, FO_, true>), ug, ub, 0, c->stream, c->d_scal, x->data, ring, rm, rflush, pk->data, po->data, w.r->data, n, nbu, h, npx); \
            else hipLaunchKernelGGL((cg_update_ring<NT_, FO_, false>), ug, ub, 0, c->stream, c->d_scal, x->data, ring, rm, rflush, pk->data, po->data, w.r->data, n, nbu, h, npx); } while (0)
        if (!defer) hipLaunchKernelGGL(cg_update_xp, ug, ub, 0, c->stream, c->d_scal, x->data, w.p->data, w.r->data, n);
        else if (ringed) {
            const int rm = w.k % w.ring, rflush = rm == w.ring - 1 ? 1 : 0;
            const CgRing ring = cg_ring_pending(w, rm);
            if (c->tun.nt_blas) { if (fold) LQ_RING3(true, true); else LQ_RING3(true, false); }
            else { if (fold) LQ_RING3(false, true); else LQ_RING3(false, false); }
        }
        else if (w.k & 1) LQ_UPD(cg_update_odd, po->data, pk->data, w.r->data);
        else LQ_UPD(cg_update_even, pk->data, po->data, w.r->data);
#undef LQ_RING3
#undef LQ_UPD
#undef LQ_UPD3
        w.p_packed = fp;
        if (fp) w.pack_epoch = ++c->halo_epoch;
        HIPCHK(hipGetLastError());
        w.k++;
        return LQCD_OK;
    }
    const bool fuse = c->tun.cg_fused && !any_partitioned(c);
    // tmp = D p  (|tmp|^2 block partials fused into the stencil when the lattice is not partitioned)
    LQCHK(op_apply_async(op, w.tmp, w.p, 0, fuse ? c->d_partial : nullptr));
    int nb;
    if (fuse) {
        nb = stencil_num_blocks(c, op->kind, op->r, 2, 0, op_fused_clover(op));
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    } else if (c->tun.cg_fused) {
        nb = stream_grid(c, n);
        hipLaunchKernelGGL(norm2_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, w.tmp->data, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    } else {
        // reference form: c1 = p . q
        LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
        nb = stream_grid(c, n);
        hipLaunchKernelGGL(redot_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, w.p->data, w.q->data, n, c->d_partial);
        HIPCHK(hipGetLastError());
        LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
    }
    hipLaunchKernelGGL(cg_scalar_alpha, dim3(1), dim3(1), 0, c->stream, c->d_scal);
    nb = stream_grid(c, n);
    hipLaunchKernelGGL(cg_update_xr, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, x->data, w.r->data, w.p->data, w.q->data, n, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 1, S_RRNEW, true));
    hipLaunchKernelGGL(cg_scalar_beta, dim3(1), dim3(1), 0, c->stream, c->d_scal);
    hipLaunchKernelGGL(cg_update_p, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, w.p->data, w.r->data, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int cg_setup(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, CgWork& w, double eps, double* rr0) {
    lqcd_ctx_s* c = op->ctx;
    const size_t n = x->elems;
    LQCHK(halo_schedule_settle(op));      // the iteration counts |.|^2 partials: the halo schedule (folded or not) is fixed from here on
    // r = b - D^+ D x ; p = r
    LQCHK(op_apply_async(op, w.tmp, x, 0, nullptr));
    LQCHK(op_apply_async(op, w.q, w.tmp, 1, nullptr));
    HIPCHK(hipMemcpyAsync(w.r->data, b->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, w.q->data, w.r->data, n));
    HIPCHK(hipMemcpyAsync(w.p->data, w.r->data, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_norm2(c, w.r->data, n, rr0, true));
    double init[9] = {*rr0, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    w.k = 0; w.p_packed = false;
    w.form = (c->tun.cg_fused >= 2 && c->tun.cg_small && cg_small_ok(op, stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op)))) ? 2
             : (cg_defers_x(op) ? 1 : 0);
    w.ring = 2;
    if (w.form == 1) {      // the ring of search-direction buffers (cg_defer_x = K): K - 2 more vectors, or the two-buffer form if the pool cannot grow
        int want = cg_ring_wanted(op);
        // a captured burst (tunable graph) replays the launches of iterations k = 0..7 with their buffer roles baked in: the ring must return to its starting state after
        // 8 iterations, i.e. divide the burst (K = 3, 5, 6, 7 silently broke the replay) -- otherwise the two-buffer form
        if (c->tun.graph != 0 && !any_partitioned(c) && (8 % want) != 0) want = 2;
        bool ok = true;
        for (int j = 0; j < want - 2 && ok; j++) {
            if (!w.more[j]) w.more[j] = scratch_get(c, x->kind, LQCD_FULL);
            ok = w.more[j] != nullptr;
        }
        if (ok) w.ring = want;
        else for (lqcd_spinor_s*& f : w.more) { if (f) scratch_put(f); f = nullptr; }
    }
    return LQCD_OK;
}

int cg_run(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double eps, int maxiter, bool fixed, int* iters, double* final_rr) {
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    CgWork w;
    int st = cg_work_get(c, x->kind, w);
    double rr = 0;
    int it = 0;
    bool converged = false;
    // launch-bound staggered lattices: initial residual and all iterations in one launch (cg_persist.hip); x is complete on return
    bool one_launch = st == LQCD_OK && maxiter > 0 && c->tun.graph == 0 && cg_persist_ok(op);
    if (one_launch) {
        bool gave_up = false;
        st = cg_persist_run(op, x, b, w, fixed ? -1.0 : eps, maxiter, &it, &rr, &converged, &gave_up);
        if (gave_up) { one_launch = false; it = 0; rr = 0; converged = false; c->tun.cg_persist = 0; }      // not all workgroups were resident: the chain from here on
        else if (st == LQCD_OK && !std::isfinite(rr)) { set_error("CG: residual is not finite"); st = LQCD_ERR_NOT_CONVERGED; }
    }
    if (st == LQCD_OK && !one_launch) st = cg_setup(op, x, b, w, fixed ? -1.0 : eps, &rr);
    if (st == LQCD_OK && !one_launch && !fixed && rr < eps) converged = true;
    const int check_every = 8;
    // Launch-bound lattices (the cg_small regime: <= 1024 stencil workgroups, an iteration is three ~5 us launches): every readback of the
    // scalar block is a host synchronisation worth about two iterations, an iteration enqueued behind the converging one only three no-op
    // launches.  There the burst length follows the observed convergence rate -- enough iterations to reach eps by the rate of the last
    // burst, at most 32 -- instead of a fixed 8.  The returned iterate and count do not depend on the burst length (done-flag protocol).
    const bool adaptive = !fixed && c->tun.graph == 0 && c->tun.cg_fused >= 2 && c->tun.cg_small && cg_small_ok(op, stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op)));
    int next_burst = check_every, it_prev = 0;
    double rr_prev = rr;
    // tunable "graph": a burst of check_every iterations is captured once into a hipGraph and replayed -- one launch per
    // burst instead of 5 per iteration.  Pays on launch-bound (small) lattices; single-stream (unpartitioned) contexts only.
    const bool use_graph = c->tun.graph != 0 && !any_partitioned(c) && !(c->has_comm && c->peer.on);      // (the peer-mapped reductions carry a sequence number per launch)
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    while (st == LQCD_OK && !one_launch && !converged && it < maxiter) {
        int burst = std::min(adaptive ? next_burst : check_every, maxiter - it);
        if (fixed && !use_graph) burst = maxiter - it;
        if (use_graph && burst == check_every) {
            if (!gexec) {
                if (c->tun.halo_stream_mode < 0) c->tun.halo_stream_mode = 0;    // the auto-tuning pass synchronises: not inside a capture
                hipError_t ge = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
                if (ge != hipSuccess) { st = hip_fail(ge, "hipStreamBeginCapture", __FILE__, __LINE__); break; }
                for (int k = 0; k < burst && st == LQCD_OK; k++) st = cg_enqueue_iteration(op, x, w);
                ge = hipStreamEndCapture(c->stream, &graph);
                if (st == LQCD_OK && ge != hipSuccess) st = hip_fail(ge, "hipStreamEndCapture", __FILE__, __LINE__);
                if (st == LQCD_OK) {
                    ge = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
                    if (ge != hipSuccess) st = hip_fail(ge, "hipGraphInstantiate", __FILE__, __LINE__);
                }
                if (st != LQCD_OK) break;
            }
            hipError_t ge = hipGraphLaunch(gexec, c->stream);
            if (ge != hipSuccess) { st = hip_fail(ge, "hipGraphLaunch", __FILE__, __LINE__); break; }
        } else {
            for (int k = 0; k < burst && st == LQCD_OK; k++) st = cg_enqueue_iteration(op, x, w);
        }
        if (st != LQCD_OK) break;
        if (fixed && use_graph && it + burst < maxiter) { it += burst; continue; }   // timing window: no readback between bursts
        hipError_t e = hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { st = hip_fail(e, "cg scalar readback", __FILE__, __LINE__); break; }
        rr = c->h_scal[S_RR - S_RR];
        it = (int)c->h_scal[S_ITERS - S_RR];
        if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
        if (!std::isfinite(rr)) { set_error("CG: residual is not finite"); st = LQCD_ERR_NOT_CONVERGED; }
        if (adaptive && !converged && it > it_prev && rr > 0.0 && rr < rr_prev) {
            const double per_it = std::log(rr / rr_prev) / (double)(it - it_prev);     // < 0
            const double need = std::log(eps / rr) / per_it;
            next_burst = (int)std::min(32.0, std::max(2.0, std::ceil(need)));
        } else if (adaptive) {
            next_burst = check_every;
        }
        it_prev = it; rr_prev = rr;
    }
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    if (st == LQCD_OK && !converged && !one_launch) {          // a window / an exhausted solve that stopped on an even iteration: complete x
        st = cg_flush_x(op, x, w);
        if (st == LQCD_OK) { hipError_t e = hipStreamSynchronize(c->stream); if (e != hipSuccess) st = hip_fail(e, "cg flush", __FILE__, __LINE__); }
    }
    cg_work_put(w);
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (st != LQCD_OK) { c->has_waiting_pack = false; return st; }      // (a pack left waiting by the failed iteration names work vectors that went back to the pool)
    if (comm_check(c) != LQCD_OK) return LQCD_ERR_COMM;                 // peer-mapped backend: a wait gave up (dead rank)
    if (!fixed && !converged) {
        set_error("The CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- BiCGStab (device-resident scalars)
// One iteration = 2 operator applications + 5 streaming kernels + 4 single-block reductions, all enqueued without a host
// round trip; complex alpha/omega/beta live in d_scal[B_*] (scalar steps: blas.hip cg_scalar_step ops 3..6).  The host
// polls the done flag every few iterations.  Same recurrences, stopping rule (|s|^2 < eps half-step exit, |r|^2 < eps) and
// iteration count as the textbook van der Vorst loop the parity tests compare against.

// <a,b> = sum conj(a) b
__global__ __launch_bounds__(UB) void bicg_dot(const double* __restrict__ sc, const double2* __restrict__ a, const double2* __restrict__ b, size_t n,
                                                double* partial) {
    if (sc[B_DONE] != 0.0) return;
    double acc[2] = {0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = a[i], y = b[i];
        acc[0] = fma(x.x, y.x, acc[0]); acc[0] = fma(x.y, y.y, acc[0]);
        acc[1] = fma(x.x, y.y, acc[1]); acc[1] = fma(-x.y, y.x, acc[1]);
    }
    block_reduce_nv<2>(acc, partial);
}
// s = r - alpha v ; partial |s|^2
__global__ __launch_bounds__(UB) void bicg_s(const double* __restrict__ sc, double2* __restrict__ s, const double2* __restrict__ r,
                                              const double2* __restrict__ v, size_t n, double* partial) {
    if (sc[B_DONE] != 0.0) return;
    const double ar = sc[B_ALPHA], ai = sc[B_ALPHA + 1];
    double acc[1] = {0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 vv = v[i];
        double2 sv = r[i];
        sv.x = fma(-ar, vv.x, sv.x); sv.x = fma(ai, vv.y, sv.x);
        sv.y = fma(-ar, vv.y, sv.y); sv.y = fma(-ai, vv.x, sv.y);
        s[i] = sv;
        acc[0] = fma(sv.x, sv.x, acc[0]); acc[0] = fma(sv.y, sv.y, acc[0]);
    }
    block_reduce_nv<1>(acc, partial);
}
// partials of <t,s> (2 values) and |t|^2
__global__ __launch_bounds__(UB) void bicg_ts(const double* __restrict__ sc, const double2* __restrict__ t, const double2* __restrict__ s, size_t n,
                                               double* partial) {
    if (sc[B_DONE] != 0.0) return;
    double acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 x = t[i], y = s[i];
        acc[0] = fma(x.x, y.x, acc[0]); acc[0] = fma(x.y, y.y, acc[0]);
        acc[1] = fma(x.x, y.y, acc[1]); acc[1] = fma(-x.y, y.x, acc[1]);
        acc[2] = fma(x.x, x.x, acc[2]); acc[2] = fma(x.y, x.y, acc[2]);
    }
    block_reduce_nv<3>(acc, partial);
}
// x += alpha p + omega s ; r = s - omega t ; partials |r|^2, <r0,r>
__global__ __launch_bounds__(UB) void bicg_xr(const double* __restrict__ sc, double2* __restrict__ x, double2* __restrict__ r,
                                               const double2* __restrict__ p, const double2* __restrict__ s, const double2* __restrict__ t,
                                               const double2* __restrict__ r0, size_t n, double* partial) {
    if (sc[B_DONE] != 0.0) return;
    const double ar = sc[B_ALPHA], ai = sc[B_ALPHA + 1], wr = sc[B_OMEGA], wi = sc[B_OMEGA + 1];
    double acc[3] = {0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 pv = p[i], sv = s[i], tv = t[i], zv = r0[i];
        double2 xv = x[i], rv = sv;
        xv.x = fma(ar, pv.x, xv.x); xv.x = fma(-ai, pv.y, xv.x);
        xv.y = fma(ar, pv.y, xv.y); xv.y = fma(ai, pv.x, xv.y);
        xv.x = fma(wr, sv.x, xv.x); xv.x = fma(-wi, sv.y, xv.x);
        xv.y = fma(wr, sv.y, xv.y); xv.y = fma(wi, sv.x, xv.y);
        rv.x = fma(-wr, tv.x, rv.x); rv.x = fma(wi, tv.y, rv.x);
        rv.y = fma(-wr, tv.y, rv.y); rv.y = fma(-wi, tv.x, rv.y);
        x[i] = xv; r[i] = rv;
        acc[0] = fma(rv.x, rv.x, acc[0]); acc[0] = fma(rv.y, rv.y, acc[0]);
        acc[1] = fma(zv.x, rv.x, acc[1]); acc[1] = fma(zv.y, rv.y, acc[1]);
        acc[2] = fma(zv.x, rv.y, acc[2]); acc[2] = fma(-zv.y, rv.x, acc[2]);
    }
    block_reduce_nv<3>(acc, partial);
}
// p = r + beta (p - omega v)
__global__ __launch_bounds__(UB) void bicg_p(const double* __restrict__ sc, double2* __restrict__ p, const double2* __restrict__ r,
                                              const double2* __restrict__ v, size_t n) {
    if (sc[B_DONE] != 0.0) return;
    const double br = sc[B_BETA], bi = sc[B_BETA + 1], wr = sc[B_OMEGA], wi = sc[B_OMEGA + 1];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 vv = v[i], rv = r[i];
        double2 pv = p[i];
        pv.x = fma(-wr, vv.x, pv.x); pv.x = fma(wi, vv.y, pv.x);
        pv.y = fma(-wr, vv.y, pv.y); pv.y = fma(-wi, vv.x, pv.y);
        double2 o;
        o.x = fma(br, pv.x, rv.x); o.x = fma(-bi, pv.y, o.x);
        o.y = fma(br, pv.y, rv.y); o.y = fma(bi, pv.x, o.y);
        p[i] = o;
    }
}

static int bicgstab_core(lqcd_ctx_s* c, const ApplyFn& A, size_t n, double2* x, const double2* b, double2* const w[6], double eps,
                         int maxiter, int* iters, double* final_rr) {
    double2 *r = w[0], *r0 = w[1], *p = w[2], *v = w[3], *s = w[4], *t = w[5];
    const size_t bytes = n * sizeof(double2);
    LQCHK(A(v, x));
    HIPCHK(hipMemcpyAsync(r, b, bytes, hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, v, r, n));
    HIPCHK(hipMemcpyAsync(r0, r, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(p, r, bytes, hipMemcpyDeviceToDevice, c->stream));
    double rr;
    LQCHK(blas_norm2(c, r, n, &rr, true));
    double init[B_END - B_RHO] = {0};
    init[B_RHO - B_RHO] = rr;
    init[B_EPS - B_RHO] = eps;
    init[B_RES - B_RHO] = rr;
    HIPCHK(hipMemcpyAsync(c->d_scal + B_RHO, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int it = 0, st = LQCD_ERR_NOT_CONVERGED;
    bool breakdown = false;
    if (rr < eps) st = LQCD_OK;
    const int nb = stream_grid(c, n), check_every = 4;
    const double* sc = c->d_scal;
    while (st != LQCD_OK && !breakdown && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            LQCHK(A(v, p));
            hipLaunchKernelGGL(bicg_dot, dim3(nb), dim3(UB), 0, c->stream, sc, r0, v, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 2, B_R0V, true, 3));
            hipLaunchKernelGGL(bicg_s, dim3(nb), dim3(UB), 0, c->stream, sc, s, r, v, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 1, B_SS, true, 4));
            LQCHK(A(t, s));
            hipLaunchKernelGGL(bicg_ts, dim3(nb), dim3(UB), 0, c->stream, sc, t, s, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 3, B_TS, true, 5));
            hipLaunchKernelGGL(bicg_xr, dim3(nb), dim3(UB), 0, c->stream, sc, x, r, p, s, t, r0, n, c->d_partial);
            LQCHK(reduce_to_slot(c, nb, 3, B_RR, true, 6));
            hipLaunchKernelGGL(bicg_p, dim3(nb), dim3(UB), 0, c->stream, sc, p, r, v, n);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + B_RHO, (B_END - B_RHO) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        it = (int)c->h_scal[B_ITERS - B_RHO];
        rr = c->h_scal[B_RES - B_RHO];
        const double done = c->h_scal[B_DONE - B_RHO];
        if (done == 1.0) st = LQCD_OK;
        else if (done != 0.0) breakdown = true;
    }
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (breakdown) { set_error("BiCGStab: residual is not finite (breakdown)"); return LQCD_ERR_NOT_CONVERGED; }
    if (st != LQCD_OK) {
        set_error("The BiCGStab is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// ---------------------------------------------------------------------------------- even-odd BiCGStab, plain Wilson: the fused chain (tunable bicg_fused)
// The Schur operator M = 1 - k^2 H_eo H_oe is two hops; its SECOND hop forms the inner product an iteration needs next in its epilogue
// (StencilCall::dot_z: <r0, v> with v = M p, and <t, s>, |t|^2 with t = M s), so no pass over the vectors exists only to multiply them.
// fold (lattices of <= 1024 chunks per parity, where an iteration is a chain of short dependent launches): the block partials of a producer are
// summed by EVERY workgroup of the consumer in its prologue (sum_partials_small_nv: the order of the one-block reduction kernel) and the scalar
// steps (bicg_alpha / bicg_omega / bicg_beta, shared with the scalar kernels of blas.hip) run there too:
//     hop, hop+<r0,v> | s = r - alpha v, |s|^2 | hop, hop+<t,s>,|t|^2 | x += alpha p + omega s, r = s - omega t, |r|^2, <r0,r> | p = r + beta (p - omega v)
// = 7 dependent launches per iteration instead of 17, the same bits in every vector as the unfolded form (partials, summation order and scalar
// expressions are the same; tests/test_gpu_solver_edges.py).  rho lives in two slots used alternately: block 0 of the p update writes the new value
// while the other workgroups still read the old one.
// The three streaming kernels request the first KE elements of every thread BEFORE the prologue (whose partial sums are a memory round trip of their
// own and do not depend on them), then walk the rest of a large vector in the usual grid-stride loop: same element -> thread map, same order of the
// per-thread additions, hence the same partials whether the prologue folds a reduction or not.
constexpr int KE = 3;
// s = r - alpha v ; partial |s|^2          (fold: alpha = rho / <r0, v> from the partials of the Schur operator's epilogue)
__global__ __launch_bounds__(UB) void bicgf_s(BicgF a, double2* __restrict__ s, const double2* __restrict__ r, const double2* __restrict__ v, size_t n) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    double2 pr[KE], pv[KE];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) { pr[e] = r[i]; pv[e] = v[i]; }
    }
    // the sums: from the producer's partials (fold) or from the slots a one-block reduction launch filled; the scalar step is formed HERE in both
    // forms -- the same instructions, hence the same bits
    c2 r0v, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]};
    if (a.fold) {
        double t3[3];
        block_sum_partials<3>(a.pin, a.pin_n, t3, a.pin_soa != 0);
        r0v.re = t3[0]; r0v.im = t3[1];
    } else { r0v.re = a.sc[B_R0V]; r0v.im = a.sc[B_R0V + 1]; }
    const c2 al = bicg_alpha(rho, r0v);
    if (a.pin3 && a.sc[B_UNSURE] != 0.0) {      // bicg_fused = 4: the last update launch left the stopping test to the |r'|^2 it summed (every workgroup reaches the same verdict)
        double t1[1];
        block_sum_partials<1>(a.pin3, a.pin3_n, t1);
        if (blockIdx.x == 0 && threadIdx.x == 0) { a.sc[B_RES] = t1[0]; a.sc[B_RR] = t1[0]; }
        if (t1[0] < a.sc[B_EPS]) { if (blockIdx.x == 0 && threadIdx.x == 0) a.sc[B_DONE] = 1.0; return; }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.sc[B_R0V] = r0v.re; a.sc[B_R0V + 1] = r0v.im; a.sc[B_ALPHA] = al.re; a.sc[B_ALPHA + 1] = al.im; }
    const double ar = al.re, ai = al.im;
    double acc[1] = {0};
    auto one = [&](size_t i, double2 sv, const double2 vv) {
        sv.x = fma(-ar, vv.x, sv.x); sv.x = fma(ai, vv.y, sv.x);
        sv.y = fma(-ar, vv.y, sv.y); sv.y = fma(-ai, vv.x, sv.y);
        s[i] = sv;
        acc[0] = fma(sv.x, sv.x, acc[0]); acc[0] = fma(sv.y, sv.y, acc[0]);
    };
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) one(i, pr[e], pv[e]);
    }
    for (size_t i = i0 + KE * stride; i < n; i += stride) one(i, r[i], v[i]);
    block_reduce_nv<1>(acc, a.pout);
}
// x += alpha p + omega s ; r = s - omega t ; partials |r|^2, <r0, r>       (fold: half-step test on |s|^2 and omega = <t, s> / |t|^2 in the prologue)
__global__ __launch_bounds__(UB) void bicgf_xr(BicgF a, double2* __restrict__ x, double2* __restrict__ r, const double2* __restrict__ p,
                                                const double2* __restrict__ s, const double2* __restrict__ t, const double2* __restrict__ r0, size_t n) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    double2 pp[KE], ps[KE], pt[KE], pz[KE], px[KE];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) { pp[e] = p[i]; ps[e] = s[i]; pt[e] = t[i]; pz[e] = r0[i]; px[e] = x[i]; }
    }
    const double ar = a.sc[B_ALPHA], ai = a.sc[B_ALPHA + 1];
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
    const double wr = om.re, wi = om.im;
    double acc[3] = {0, 0, 0};
    auto one = [&](size_t i, const double2 pv, const double2 sv, const double2 tv, const double2 zv, double2 xv) {
        double2 rv = sv;
        xv.x = fma(ar, pv.x, xv.x); xv.x = fma(-ai, pv.y, xv.x);
        xv.y = fma(ar, pv.y, xv.y); xv.y = fma(ai, pv.x, xv.y);
        xv.x = fma(wr, sv.x, xv.x); xv.x = fma(-wi, sv.y, xv.x);
        xv.y = fma(wr, sv.y, xv.y); xv.y = fma(wi, sv.x, xv.y);
        rv.x = fma(-wr, tv.x, rv.x); rv.x = fma(wi, tv.y, rv.x);
        rv.y = fma(-wr, tv.y, rv.y); rv.y = fma(-wi, tv.x, rv.y);
        x[i] = xv; r[i] = rv;
        acc[0] = fma(rv.x, rv.x, acc[0]); acc[0] = fma(rv.y, rv.y, acc[0]);
        acc[1] = fma(zv.x, rv.x, acc[1]); acc[1] = fma(zv.y, rv.y, acc[1]);
        acc[2] = fma(zv.x, rv.y, acc[2]); acc[2] = fma(-zv.y, rv.x, acc[2]);
    };
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) one(i, pp[e], ps[e], pt[e], pz[e], px[e]);
    }
    for (size_t i = i0 + KE * stride; i < n; i += stride) one(i, p[i], s[i], t[i], r0[i], x[i]);
    block_reduce_nv<3>(acc, a.pout);
}
// p = r + beta (p - omega v)       (fold: iteration count, convergence / breakdown and beta = (rho'/rho)(alpha/omega) in the prologue)
__global__ __launch_bounds__(UB) void bicgf_p(BicgF a, double2* __restrict__ p, const double2* __restrict__ r, const double2* __restrict__ v, size_t n) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    double2 pv_[KE], pr[KE], pp[KE];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) { pv_[e] = v[i]; pr[e] = r[i]; pp[e] = p[i]; }
    }
    const double wr = a.sc[B_OMEGA], wi = a.sc[B_OMEGA + 1];
    const bool half = a.sc[B_HALF] != 0.0;
    double rrn;
    c2 rho1, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]}, al = {a.sc[B_ALPHA], a.sc[B_ALPHA + 1]}, om = {wr, wi};
    if (a.fold) {
        double t3[3];
        block_sum_partials<3>(a.pin, a.pin_n, t3);
        rrn = t3[0]; rho1.re = t3[1]; rho1.im = t3[2];
    } else { rrn = a.sc[B_RR]; rho1.re = a.sc[B_RHO1]; rho1.im = a.sc[B_RHO1 + 1]; }
    const double rr = half ? a.sc[B_SS] : rrn;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead) { a.sc[B_ITERS] += 1.0; a.sc[B_RES] = rr; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im; }
    if (half || rr < a.sc[B_EPS]) { if (lead) a.sc[B_DONE] = 1.0; return; }
    if (!(fabs(rr) <= 1.79e308)) { if (lead) a.sc[B_DONE] = 2.0; return; }      // NaN / inf: breakdown
    const c2 be = bicg_beta(rho1, rho, al, om);
    if (lead) { a.sc[B_BETA] = be.re; a.sc[B_BETA + 1] = be.im; a.sc[a.rho_out] = rho1.re; a.sc[a.rho_out + 1] = rho1.im; }
    const double br = be.re, bi = be.im;
    auto one = [&](size_t i, const double2 vv, const double2 rv, double2 pv) {
        pv.x = fma(-wr, vv.x, pv.x); pv.x = fma(wi, vv.y, pv.x);
        pv.y = fma(-wr, vv.y, pv.y); pv.y = fma(-wi, vv.x, pv.y);
        double2 o;
        o.x = fma(br, pv.x, rv.x); o.x = fma(-bi, pv.y, o.x);
        o.y = fma(br, pv.y, rv.y); o.y = fma(bi, pv.x, o.y);
        p[i] = o;
    };
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) one(i, pv_[e], pr[e], pp[e]);
    }
    for (size_t i = i0 + KE * stride; i < n; i += stride) one(i, v[i], r[i], p[i]);
}

// ---- bicg_fused = 3 (round 6, opt-in: measured SLOWER than two launches, 128.6 vs 112.4 us per iteration at 16^3x32 -- a barrier of 1024 workgroups costs more than the launch boundary): the x / r update and the p update
// as ONE launch with a grid-wide barrier between them.  The p update needs rho' = <r0, r> of the WHOLE new
// residual, i.e. a global dependency -- but not a new launch: all <= 1024 workgroups of the streaming launch are resident at once (checked with the occupancy query), so
// they can meet at a barrier (arrivals spread over eight counters 128 B apart, the scheme of cg_persist.hip), sum the block partials themselves and go on with the
// elements they still hold in registers: the new r and the old p are not read again, one launch boundary and its prologue are gone.  Same operations on the same values
// in the same order as bicgf_xr + bicgf_p: the same bits (tests/test_gpu_solver_edges.py).  The partials cross workgroups (and XCDs, each with its own L2) inside a
// running kernel: they are stored and loaded with agent-scope atomics.
__device__ inline double ldc_a(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline double sum_partials_small_nv_c(const double* partial, int n, int nvals, int v) {      // sum_partials_small_nv (lqcd_internal.h) on agent-scope loads: the same order
    const int lane = threadIdx.x & 63;
    double t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = lane + 64 * k;
        t[k] = i < n ? ldc_a(partial + (size_t)i * nvals + v) : 0.0;
    }
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) s += t[k];
    return wave_sum(s);
}
constexpr unsigned long long kBarrierLimit = 20000000ull;      // ticks of 10 ns: 0.2 s
// all workgroups of the launch; false: gave up (somebody else holds the device: the workgroups were not all resident)
__device__ inline bool grid_barrier_sharded(unsigned* ctr, unsigned epoch, int nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's published stores are acknowledged
    __syncthreads();                                        // ... and those of the other waves of the workgroup
    __shared__ int okw;
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane == 0) __hip_atomic_fetch_add(ctr + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned mine = lane < 8 ? (unsigned)((nwg - lane + 7) / 8) * epoch : 0u;      // arrivals this lane's counter must show
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        for (;;) {
            const unsigned v = lane < 8 ? __hip_atomic_load(ctr + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (__all((int)(v - mine) >= 0)) break;      // wrap-safe comparison
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > kBarrierLimit) { ok = 0; break; }
        }
        if (lane == 0) okw = ok;
    }
    __syncthreads();
    return okw != 0;
}
__global__ __launch_bounds__(UB, 4) void bicgf_xrp(BicgF a, double2* __restrict__ x, double2* __restrict__ r, double2* __restrict__ p, const double2* __restrict__ s,
                                                 const double2* __restrict__ t, const double2* __restrict__ r0, const double2* __restrict__ v, size_t n,
                                                 unsigned* ctr, unsigned epoch) {
    const bool done0 = a.sc[B_DONE] != 0.0;      // (every workgroup still meets the barrier: the host counts one per launch)
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    double2 pp[KE], ps[KE], pt[KE], pz[KE], px[KE], pv_[KE], rn[KE];
    double wr = 0.0, wi = 0.0, ss = 0.0;
    bool half = false;
    if (!done0) {
#pragma unroll
        for (int e = 0; e < KE; e++) {
            const size_t i = i0 + e * stride;
            if (i < n) { pp[e] = p[i]; ps[e] = s[i]; pt[e] = t[i]; pz[e] = r0[i]; px[e] = x[i]; }
        }
        // ---- the body of bicgf_xr
        const double ar = a.sc[B_ALPHA], ai = a.sc[B_ALPHA + 1];
        double tt;
        c2 ts;
        {
            double t1[1], t3[3];
            block_sum_partials<1>(a.pin2, a.pin2_n, t1);
            block_sum_partials<3>(a.pin, a.pin_n, t3);      // (this form keeps the [workgroup][value] partials: the host never asks for the other layout with it -- the kernel sits at its register cap)
            ss = t1[0]; ts.re = t3[0]; ts.im = t3[1]; tt = t3[2];
        }
        half = ss < a.sc[B_EPS];
        const c2 om = bicg_omega(ts, tt, half);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            a.sc[B_SS] = ss; a.sc[B_HALF] = half ? 1.0 : 0.0; a.sc[B_TS] = ts.re; a.sc[B_TS + 1] = ts.im; a.sc[B_TT] = tt;
            a.sc[B_OMEGA] = om.re; a.sc[B_OMEGA + 1] = om.im;
        }
        wr = om.re; wi = om.im;
        double acc[3] = {0, 0, 0};
        auto one = [&](size_t i, const double2 pv, const double2 sv, const double2 tv, const double2 zv, double2 xv) -> double2 {
            double2 rv = sv;
            xv.x = fma(ar, pv.x, xv.x); xv.x = fma(-ai, pv.y, xv.x);
            xv.y = fma(ar, pv.y, xv.y); xv.y = fma(ai, pv.x, xv.y);
            xv.x = fma(wr, sv.x, xv.x); xv.x = fma(-wi, sv.y, xv.x);
            xv.y = fma(wr, sv.y, xv.y); xv.y = fma(wi, sv.x, xv.y);
            rv.x = fma(-wr, tv.x, rv.x); rv.x = fma(wi, tv.y, rv.x);
            rv.y = fma(-wr, tv.y, rv.y); rv.y = fma(-wi, tv.x, rv.y);
            x[i] = xv; r[i] = rv;
            acc[0] = fma(rv.x, rv.x, acc[0]); acc[0] = fma(rv.y, rv.y, acc[0]);
            acc[1] = fma(zv.x, rv.x, acc[1]); acc[1] = fma(zv.y, rv.y, acc[1]);
            acc[2] = fma(zv.x, rv.y, acc[2]); acc[2] = fma(-zv.y, rv.x, acc[2]);
            return rv;
        };
#pragma unroll
        for (int e = 0; e < KE; e++) {
            const size_t i = i0 + e * stride;
            if (i < n) rn[e] = one(i, pp[e], ps[e], pt[e], pz[e], px[e]);
        }
        for (size_t i = i0 + KE * stride; i < n; i += stride) (void)one(i, p[i], s[i], t[i], r0[i], x[i]);
        // v for the p update: requested now (the registers of s, t, r0, x are free), it lands while the workgroups meet at the barrier
#pragma unroll
        for (int e = 0; e < KE; e++) {
            const size_t i = i0 + e * stride;
            if (i < n) pv_[e] = v[i];
        }
        // block partials in block_reduce_nv's order, published with agent-scope stores
        __shared__ double red[3][UB / 64];
#pragma unroll
        for (int q = 0; q < 3; q++) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, 64);
            if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = acc[q];
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            double tsum = 0;
#pragma unroll
            for (int w = 0; w < UB / 64; w++) tsum += red[threadIdx.x][w];
            __hip_atomic_store(a.pout + (size_t)blockIdx.x * 3 + threadIdx.x, tsum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const bool met = grid_barrier_sharded(ctr, epoch, (int)gridDim.x);
    if (done0) return;
    if (!met) { if (threadIdx.x == 0) a.sc[B_DONE] = 3.0; return; }      // the host reports it (and switches the fusion off for this context)
    // ---- the body of bicgf_p, on the registers of the elements this thread has just updated
    double rrn;
    c2 rho1, rho = {a.sc[a.rho_in], a.sc[a.rho_in + 1]}, al = {a.sc[B_ALPHA], a.sc[B_ALPHA + 1]}, om = {wr, wi};
    {
        __shared__ double sh3[3];
        if (threadIdx.x < 64) {
#pragma unroll
            for (int q = 0; q < 3; q++) {
                const double tq = sum_partials_small_nv_c(a.pout, (int)gridDim.x, 3, q);
                if (threadIdx.x == 0) sh3[q] = tq;
            }
        }
        __syncthreads();
        rrn = sh3[0]; rho1.re = sh3[1]; rho1.im = sh3[2];
    }
    const double rr = half ? ss : rrn;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    if (lead) { a.sc[B_ITERS] += 1.0; a.sc[B_RES] = rr; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im; }
    if (half || rr < a.sc[B_EPS]) { if (lead) a.sc[B_DONE] = 1.0; return; }
    if (!(fabs(rr) <= 1.79e308)) { if (lead) a.sc[B_DONE] = 2.0; return; }      // NaN / inf: breakdown
    const c2 be = bicg_beta(rho1, rho, al, om);
    if (lead) { a.sc[B_BETA] = be.re; a.sc[B_BETA + 1] = be.im; a.sc[a.rho_out] = rho1.re; a.sc[a.rho_out + 1] = rho1.im; }
    const double br = be.re, bi = be.im;
    auto onep = [&](size_t i, const double2 vv, const double2 rv, double2 pv) {
        pv.x = fma(-wr, vv.x, pv.x); pv.x = fma(wi, vv.y, pv.x);
        pv.y = fma(-wr, vv.y, pv.y); pv.y = fma(-wi, vv.x, pv.y);
        double2 o;
        o.x = fma(br, pv.x, rv.x); o.x = fma(-bi, pv.y, o.x);
        o.y = fma(br, pv.y, rv.y); o.y = fma(bi, pv.x, o.y);
        p[i] = o;
    };
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) onep(i, pv_[e], rn[e], pp[e]);
    }
    for (size_t i = i0 + KE * stride; i < n; i += stride) onep(i, v[i], r[i], p[i]);
}


// bicg_fused = 4: x / r update and p update as ONE launch WITHOUT a barrier.  What the p update needs from the new residual -- rho' = <r0, r'> and the stopping test --
// follows from inner products that exist before r' does:  r' = s - omega t and s = r - alpha v give
//     rho' = rho - alpha <r0, v> - omega <r0, t> ,        |r'|^2 = |s|^2 - |<t, s>|^2 / |t|^2
// with <r0, t> formed next to <t, s>, |t|^2 in the epilogue of the Schur operator's second hop (StencilCall::dot_z2: five values per workgroup).  One launch, one
// reduction and two vector passes less per iteration than bicg_fused = 2 (x, p, s, t, v read, x, r, p written; r0 is read by the hop instead of here).  The iterates
// equal those of the other forms up to the rounding of the two recurrences (fp64: ~1e-16 of |r0| |r| per iteration, tests/test_gpu_solver_edges.py); the
// stopping test trusts the recurrence for |r'|^2 only while it is free of cancellation (|r'|^2 > 1e-6 |s|^2), otherwise this launch also sums the |r'|^2 it
// writes and the NEXT iteration's first streaming kernel decides (bicgf_s, a.pin3; the two hops in between are wasted once).
__global__ __launch_bounds__(UB) void bicgf_xrp_rec(BicgF a, double2* __restrict__ x, double2* __restrict__ r, double2* __restrict__ p, const double2* __restrict__ s,
                                                     const double2* __restrict__ t, const double2* __restrict__ v, size_t n) {
    if (a.sc[B_DONE] != 0.0) return;
    const size_t i0 = (size_t)blockIdx.x * UB + threadIdx.x, stride = (size_t)gridDim.x * UB;
    double2 pp[KE], ps[KE], pt[KE], px[KE], pv_[KE];
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) { pp[e] = p[i]; ps[e] = s[i]; pt[e] = t[i]; px[e] = x[i]; pv_[e] = v[i]; }
    }
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
    // rho' = rho - alpha <r0, v> - omega <r0, t>
    c2 rho1;
    rho1.re = rho.re - (al.re * r0v.re - al.im * r0v.im) - (om.re * r0t.re - om.im * r0t.im);
    rho1.im = rho.im - (al.re * r0v.im + al.im * r0v.re) - (om.re * r0t.im + om.im * r0t.re);
    const double rrn = half ? ss : ss - (ts.re * ts.re + ts.im * ts.im) / tt;
    const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
    const bool finite = fabs(rrn) <= 1.79e308 && fabs(ss) <= 1.79e308;
    const bool done = half || (finite && rrn < a.sc[B_EPS] && rrn > a.guard * ss);
    const bool unsure = !half && finite && !(rrn > a.guard * ss);      // the recurrence for |r'|^2 has cancelled six digits: the sum below decides, one kernel later
    if (lead) {
        a.sc[B_SS] = ss; a.sc[B_HALF] = half ? 1.0 : 0.0; a.sc[B_TS] = ts.re; a.sc[B_TS + 1] = ts.im; a.sc[B_TT] = tt;
        a.sc[B_OMEGA] = om.re; a.sc[B_OMEGA + 1] = om.im;
        a.sc[B_ITERS] += 1.0; a.sc[B_RES] = rrn; a.sc[B_RR] = rrn; a.sc[B_RHO1] = rho1.re; a.sc[B_RHO1 + 1] = rho1.im;
        if (done) a.sc[B_DONE] = 1.0;
        else if (!finite) a.sc[B_DONE] = 2.0;
    }
    const double ar = al.re, ai = al.im, wr = om.re, wi = om.im;
    c2 be = {0.0, 0.0};
    const bool go_on = !done && finite;
    if (go_on) {
        be = bicg_beta(rho1, rho, al, om);
        if (lead) { a.sc[B_BETA] = be.re; a.sc[B_BETA + 1] = be.im; a.sc[a.rho_out] = rho1.re; a.sc[a.rho_out + 1] = rho1.im; }
    }
    const double br = be.re, bi = be.im;
    double acc[1] = {0};
    auto one = [&](size_t i, double2 pv, const double2 sv, const double2 tv, double2 xv, const double2 vv) {
        double2 rv = sv;
        xv.x = fma(ar, pv.x, xv.x); xv.x = fma(-ai, pv.y, xv.x);
        xv.y = fma(ar, pv.y, xv.y); xv.y = fma(ai, pv.x, xv.y);
        xv.x = fma(wr, sv.x, xv.x); xv.x = fma(-wi, sv.y, xv.x);
        xv.y = fma(wr, sv.y, xv.y); xv.y = fma(wi, sv.x, xv.y);
        rv.x = fma(-wr, tv.x, rv.x); rv.x = fma(wi, tv.y, rv.x);
        rv.y = fma(-wr, tv.y, rv.y); rv.y = fma(-wi, tv.x, rv.y);
        x[i] = xv; r[i] = rv;
        acc[0] = fma(rv.x, rv.x, acc[0]); acc[0] = fma(rv.y, rv.y, acc[0]);
        if (go_on) {
            pv.x = fma(-wr, vv.x, pv.x); pv.x = fma(wi, vv.y, pv.x);
            pv.y = fma(-wr, vv.y, pv.y); pv.y = fma(-wi, vv.x, pv.y);
            double2 o;
            o.x = fma(br, pv.x, rv.x); o.x = fma(-bi, pv.y, o.x);
            o.y = fma(br, pv.y, rv.y); o.y = fma(bi, pv.x, o.y);
            p[i] = o;
        }
    };
#pragma unroll
    for (int e = 0; e < KE; e++) {
        const size_t i = i0 + e * stride;
        if (i < n) one(i, pp[e], ps[e], pt[e], px[e], pv_[e]);
    }
    for (size_t i = i0 + KE * stride; i < n; i += stride) one(i, p[i], s[i], t[i], x[i], v[i]);
    if (unsure) block_reduce_nv<1>(acc, a.pout);      // (uniform over the grid: every workgroup computed the same scalars)
    if (lead) a.sc[B_UNSURE] = unsure ? 1.0 : 0.0;    // "the sum of |r'|^2 in a.pout is waiting for a verdict"
}

// start of a solve in two launches and no host round trip (round 6; it used to be three copies, an axpy, a norm, a reduction, a read-back and an upload of the scalar
// block: 135 us in front of the first iteration of a 12-iteration solve at 16^3x32): r = rhs - v (v = M x0), r0 = r, p = r, |r|^2 partials ...
__global__ __launch_bounds__(UB) void bicgf_init(double2* __restrict__ r, double2* __restrict__ r0, double2* __restrict__ p, const double2* __restrict__ rhs,
                                                  const double2* __restrict__ v, size_t n, double* partial) {      // v == nullptr: zero guess, r = rhs
    double acc[1] = {0};
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 b = rhs[i], q = v ? v[i] : make_double2(0.0, 0.0);
        double2 o;
        o.x = b.x - q.x; o.y = b.y - q.y;
        r[i] = o; r0[i] = o; p[i] = o;
        acc[0] = fma(o.x, o.x, acc[0]); acc[0] = fma(o.y, o.y, acc[0]);
    }
    block_reduce_nv<1>(acc, partial);
}
// ... and the scalar block of the chain from their sum (<= 1024 partials, one wave): rho = rho' = |r|^2, eps, the residual, done if it is below eps already
__global__ __launch_bounds__(64) void bicgf_init_scal(const double* __restrict__ partial, int nb, double* sc, double eps) {
    const double rr = sum_partials_small_nv(partial, nb, 1, 0);
    if (threadIdx.x == 0) {
        for (int j = B_RHO; j < B_END; j++) sc[j] = 0.0;
        sc[B_RHO] = rr; sc[B_RHOB] = rr; sc[B_EPS] = eps; sc[B_RES] = rr;
        if (rr < eps) sc[B_DONE] = 1.0;
    }
}

// xe = M^-1 rhs on the even sites, M = 1 - k^2 H_eo H_oe (dagger: H -> H^+).  w[0..5] = r, r0, p, v, s, t; to: an odd-parity work vector.
// Same recurrences, stopping rule (|s|^2 < eps half-step exit, |r|^2 < eps) and iteration count as bicgstab_core.
// Ai != nullptr: Wilson-clover, M = 1 - k^2 A_ee^-1 H_eo A_oo^-1 H_oe with the packed inverse blocks applied to the hop sums inside the two hops
// (StencilCall::clover_on_hop) -- still two launches per M, no intermediate field.
int schur_wilson(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, lqcd_spinor_s* to, int dg, const double2* Ai) {      // out = (1 - k^2 [A_ee^-1] H_eo [A_oo^-1] H_oe) in, fp64
    lqcd_ctx_s* c = op->ctx;
    StencilCall s1 = make_hop_call(op, to, in, nullptr, 0.0, 1.0, dg);
    if (Ai) { s1.clover = Ai; s1.clover_on_hop = 1; }
    LQCHK(stencil_apply(c, s1));
    StencilCall s2 = make_hop_call(op, out, to, in, 1.0, -op->km * op->km, dg);
    if (Ai) { s2.clover = Ai; s2.clover_on_hop = 1; }
    return stencil_apply(c, s2);
}
int bicgstab_eo_wilson(lqcd_op_s* op, lqcd_spinor_s& xe, lqcd_spinor_s* rhs, lqcd_spinor_s* const w[6], lqcd_spinor_s* to, int dg, double eps,
                       int maxiter, int* iters, double* final_rr, const double2* Ai) {
    lqcd_ctx_s* c = op->ctx;
    const double k = op->km;
    const size_t n = xe.elems, bytes = n * sizeof(double2);
    lqcd_spinor_s *r = w[0], *r0 = w[1], *p = w[2], *v = w[3], *s = w[4], *t = w[5];
    const int nbs = (c->geom.Vh + 63) / 64;                                     // workgroups (= partials) of one DOT hop on one parity: always one per 64-site chunk (the dot
                                                                                // instances have no multi-chunk / persistent form; stencil_num_blocks would say otherwise under dslash_pipe = 1 / 3)
    const int nbk = (int)std::min<size_t>(1024, (n + UB - 1) / UB);             // streaming kernels: at most 1024 partials (one prologue sums them)
    const bool fold = c->tun.bicg_fused >= 2 && nbs <= 1024;
    // bicg_fused = 3 (4 is not a superset of it): x / r and p update as one launch with a grid barrier -- only while every workgroup of it is resident at once (and nobody shares the device: a
    // barrier that is not met within 0.2 s ends the solve with an error and switches the fusion off)
    bool xrp = fold && c->tun.bicg_fused == 3;
    if (xrp) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bicgf_xrp, UB, 0) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
        if ((long)per_cu * c->num_cu < nbk) xrp = false;
    }
    // bicg_fused = 4: the merged update launch on the two recurrences (bicgf_xrp_rec), <r0, t> from the second inner product of the dot epilogue
    const bool rec = c->tun.bicg_fused == 4;      // (every dot-mode kernel forms the second inner product: plain, clover-on-hop, scalar-addressing)
    if (rec) xrp = false;
    c->tun.bicg_xrp_active = xrp ? 1 : (rec ? 2 : 0);
    if (xrp && (c->cgp_nwg != nbk || c->cgp_epoch > 100000000u)) {      // the barrier counters (shared with the one-launch CG): never reset between launches of one grid size
        HIPCHK(hipMemsetAsync(c->cgp_ctr, 0, 9 * 32 * sizeof(unsigned), c->stream));
        c->cgp_epoch = 0; c->cgp_nwg = nbk;
    }
    double* P0 = c->d_partial;                  // <r0, v> (+ |v|^2)      [nbs x 3]
    double* P1 = P0 + (size_t)3 * nbs;          // |s|^2                  [nbk]
    double* P2 = P1 + nbk;                      // <t, s>, |t|^2          [nbs x 3]   (bicg_fused = 4: + <r0, t>, nbs x 5)
    double* P3 = P2 + (size_t)(rec ? 5 : 3) * nbs;   // |r|^2, <r0, r>    [nbk x 3]
    const bool soa = !xrp && (c->tun.bicg_dot_soa >= 2 || (c->tun.bicg_dot_soa == 1 && !fold && nbs > 1024));      // the dot partials of the hops as [value][workgroup] (tunable bicg_dot_soa)
    const double* skip_ = c->d_scal + (B_DONE - S_DONE);      // the kernels test skip[S_DONE]: the hops become no-ops once the solve is done
    auto schur = [&](lqcd_spinor_s* out, lqcd_spinor_s* in, const lqcd_spinor_s* z, double* dotp, int conj, bool skippable = true, const lqcd_spinor_s* z2 = nullptr) -> int {
        const double* skip = skippable ? skip_ : nullptr;
        StencilCall s1 = make_hop_call(op, to, in, nullptr, 0.0, 1.0, dg);      // t_o = [A_oo^-1] H_oe in
        s1.skip_flag = skip;
        if (Ai) { s1.clover = Ai; s1.clover_on_hop = 1; }
        LQCHK(stencil_apply(c, s1));
        StencilCall s2 = make_hop_call(op, out, to, in, 1.0, -k * k, dg);       // out = in - k^2 [A_ee^-1] H_eo t_o
        s2.skip_flag = skip;
        if (Ai) { s2.clover = Ai; s2.clover_on_hop = 1; }
        if (z) { s2.dot_z[0] = z->data; s2.dot_z[1] = nullptr; s2.dot_partial = dotp; s2.dot_conj = conj | (soa ? 2 : 0); }
        if (z2) { s2.dot_z2[0] = z2->data; s2.dot_z2[1] = nullptr; }
        return stencil_apply(c, s2);
    };
    // v = M x0 with hops that do not look at the done flag (the LAST solve left it raised), then r = rhs - v, r0 = p = r and the scalar block, all on the device
    const bool zero_guess = c->zero_guess_hint;      // the caller has just cleared x (the action / force solves): M x0 = 0 is not computed
    if (!zero_guess) LQCHK(schur(v, &xe, nullptr, nullptr, 0, false));
    hipLaunchKernelGGL(bicgf_init, dim3(nbk), dim3(UB), 0, c->stream, r->data, r0->data, p->data, rhs->data, zero_guess ? (const double2*)nullptr : (const double2*)v->data, n, P1);
    hipLaunchKernelGGL(bicgf_init_scal, dim3(1), dim3(64), 0, c->stream, P1, nbk, c->d_scal, eps);
    HIPCHK(hipGetLastError());
    (void)bytes;
    double rr = 0.0;
    int it = 0, st = LQCD_ERR_NOT_CONVERGED, enq = 0;
    bool breakdown = false;
    // Polling the done flag is a host round trip that idles the GPU for ~40 us: the first burst runs up to one iteration short of what the last
    // solve with this operator took (successive solves of an MD trajectory take the same count within one or two; iterations enqueued behind the
    // converging one are no-ops), later bursts are short.
    int check_every = std::max(4, std::min(op->bicg_hint, 64));      // (round 6: the last count itself -- a solve that takes it again is polled ONCE; one short made every solve pay two polls)
    while (st != LQCD_OK && !breakdown && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        check_every = 2;
        for (int q = 0; q < burst; q++, enq++) {
            BicgF a;
            a.sc = c->d_scal;
            a.fold = fold ? 1 : 0;
            a.rho_in = (enq & 1) ? B_RHOB : B_RHO;       // rho alternates between two slots: block 0 of the p update writes the next value while
            a.rho_out = (enq & 1) ? B_RHO : B_RHOB;      // the other workgroups still read this one
            a.pin2 = nullptr; a.pin2_n = 0;
            LQCHK(schur(v, p, r0, P0, 0));                                                                   // v = M p, <r0, v>
            if (!fold) LQCHK(reduce_to_slot(c, nbs, 3, B_R0V, true, 0, P0, soa));
            a.pin = P0; a.pin_n = nbs; a.pin_soa = soa ? 1 : 0; a.pout = P1;
            if (rec) { a.pin3 = P3; a.pin3_n = nbk; }
            hipLaunchKernelGGL(bicgf_s, dim3(nbk), dim3(UB), 0, c->stream, a, s->data, r->data, v->data, n);
            if (!fold && !rec) LQCHK(reduce_to_slot(c, nbk, 1, B_SS, true, 0, P1));      // (merged chain: the update launch sums the <= 1024 partials of |s|^2 itself)
            if (rec) {
                LQCHK(schur(t, s, s, P2, 1, true, r0));                                                      // t = M s, <t, s>, |t|^2, <r0, t>
                if (!fold) LQCHK(reduce_to_slot(c, nbs, 5, B_TS5, true, 0, P2, soa));
                a.pin = P2; a.pin_n = nbs; a.pin2 = P1; a.pin2_n = nbk; a.pout = P3;
                if (!fold) a.fold = 2;
                a.guard = std::pow(10.0, -(double)c->tun.bicg_rec_guard);
                hipLaunchKernelGGL(bicgf_xrp_rec, dim3(nbk), dim3(UB), 0, c->stream, a, xe.data, r->data, p->data, s->data, t->data, v->data, n);
                HIPCHK(hipGetLastError());
                continue;
            }
            LQCHK(schur(t, s, s, P2, 1));                                                                    // t = M s, <t, s>, |t|^2
            if (!fold) LQCHK(reduce_to_slot(c, nbs, 3, B_TS, true, 0, P2, soa));
            a.pin = P2; a.pin_n = nbs; a.pin2 = P1; a.pin2_n = nbk; a.pout = P3;
            if (xrp) {
                hipLaunchKernelGGL(bicgf_xrp, dim3(nbk), dim3(UB), 0, c->stream, a, xe.data, r->data, p->data, s->data, t->data, r0->data, v->data, n, c->cgp_ctr, ++c->cgp_epoch);
                HIPCHK(hipGetLastError());
                continue;
            }
            hipLaunchKernelGGL(bicgf_xr, dim3(nbk), dim3(UB), 0, c->stream, a, xe.data, r->data, p->data, s->data, t->data, r0->data, n);
            if (!fold) LQCHK(reduce_to_slot(c, nbk, 3, B_RR, true, 0, P3));
            a.pin = P3; a.pin_n = nbk; a.pin_soa = 0; a.pin2 = nullptr; a.pin2_n = 0; a.pout = nullptr;
            hipLaunchKernelGGL(bicgf_p, dim3(nbk), dim3(UB), 0, c->stream, a, p->data, r->data, v->data, n);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + B_RHO, (B_END - B_RHO) * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        it = (int)c->h_scal[B_ITERS - B_RHO];
        rr = c->h_scal[B_RES - B_RHO];
        const double done = c->h_scal[B_DONE - B_RHO];
        if (done == 1.0) st = LQCD_OK;
        else if (done == 3.0) {      // the grid barrier of the fused x / r / p launch was not met: the device is shared with somebody who keeps its workgroups from all being resident
            c->tun.bicg_fused = 2; c->cgp_nwg = -1;
            set_error("even-odd BiCGStab: the grid barrier of the fused update launch (bicg_fused = 3) timed out -- is the device shared?  bicg_fused is 2 for this context from here on; solve again");
            return LQCD_ERR_HIP;
        }
        else if (done != 0.0) breakdown = true;
    }
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (breakdown) { set_error("BiCGStab: residual is not finite (breakdown)"); return LQCD_ERR_NOT_CONVERGED; }
    if (st != LQCD_OK) {
        set_error("The BiCGStab is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    op->bicg_hint = it;
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

// ---------------------------------------------------------------------------------- C API: solvers
extern "C" int lqcd_solve_cg_DdagD(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, double eps, int maxiter, int* iters, double* final_rr) {
    if (op && op->kind == LQCD_DOMAINWALL) return dw_solve_cg(op, x, b, eps, maxiter, iters, final_rr);
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD"));
    ARGCHK(maxiter >= 0, "lqcd_solve_cg_DdagD: maxiter < 0");
    return cg_run(op, x, b, eps, maxiter, false, iters, final_rr);
}
extern "C" int lqcd_solve_cg_DdagD_fixed(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int niter) {
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD_fixed"));
    return cg_run(op, x, b, 0.0, niter, true, nullptr, nullptr);
}


// CG with device-resident scalars for a Hermitian positive operator given as an enqueue function (reference form:
// alpha = rr / <p, A p>); x holds the initial guess, work = three fields of n elements.  Used where the fused full-lattice
// iteration of cg_run does not apply (parity blocks).
int lqcd::cg_launch_update_xp(lqcd_ctx_s* c, double2* x, double2* p, const double2* r, size_t n) {
    hipLaunchKernelGGL(cg_update_xp, dim3(stream_grid(c, n)), dim3(UB), 0, c->stream, c->d_scal, x, p, r, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

int lqcd::cg_generic(lqcd_ctx_s* c, const ApplyFn& A, size_t n, double2* x, const double2* b, double2* r, double2* p, double2* q, double eps,
                     int maxiter, int* iters, double* final_rr) {
    LQCHK(A(q, x));
    HIPCHK(hipMemcpyAsync(r, b, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, q, r, n));
    HIPCHK(hipMemcpyAsync(p, r, n * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    double rr = 0;
    LQCHK(blas_norm2(c, r, n, &rr, true));
    double init[9] = {rr, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
    HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int it = 0;
    bool converged = rr < eps;
    const int nb = stream_grid(c, n), check_every = 8;
    while (!converged && it < maxiter) {
        const int burst = std::min(check_every, maxiter - it);
        for (int k = 0; k < burst; k++) {
            LQCHK(A(q, p));
            hipLaunchKernelGGL(redot_partial_kernel, dim3(nb), dim3(UB), 0, c->stream, p, q, n, c->d_partial);
            HIPCHK(hipGetLastError());
            LQCHK(reduce_to_slot(c, nb, 1, S_PQ, true));
            hipLaunchKernelGGL(cg_scalar_alpha, dim3(1), dim3(1), 0, c->stream, c->d_scal);
            hipLaunchKernelGGL(cg_update_xr, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, x, r, p, q, n, c->d_partial);
            HIPCHK(hipGetLastError());
            LQCHK(reduce_to_slot(c, nb, 1, S_RRNEW, true));
            hipLaunchKernelGGL(cg_scalar_beta, dim3(1), dim3(1), 0, c->stream, c->d_scal);
            hipLaunchKernelGGL(cg_update_p, dim3(nb), dim3(UB), 0, c->stream, c->d_scal, p, r, n);
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        rr = c->h_scal[0];
        it = (int)c->h_scal[S_ITERS - S_RR];
        if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
        if (!std::isfinite(rr)) { set_error("CG: residual is not finite"); return LQCD_ERR_NOT_CONVERGED; }
    }
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (!converged) {
        set_error("The CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// Staggered D^+D = m^2 - H^2 is block diagonal in parity: (D^+D)_pp = m^2 - H_pq H_qp.  Solves that block for the parity-p halves of
// the FULL fields x (initial guess / solution) and b with half-lattice vectors -- one Dslash-equivalent per iteration instead of
// two.  The other parity of x is not touched.  This is the solve behind the reference's "4 tastes" (Nf = 4) staggered action, whose
// pseudofermion lives on the even sites (test/test_staggered.toml).
extern "C" int lqcd_solve_cg_DdagD_parity(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int parity, double eps, int maxiter, int* iters,
                                          double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_cg_DdagD_parity"));
    ARGCHK(op->kind == LQCD_STAGGERED && (parity == 0 || parity == 1) && maxiter >= 0,
           "lqcd_solve_cg_DdagD_parity: staggered operators only, parity 0 (even) or 1 (odd)");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    apply_bc(c, op->bc);
    const size_t nh = x->elems / 2;
    const int mine = parity ? LQCD_ODD : LQCD_EVEN, other = parity ? LQCD_EVEN : LQCD_ODD;
    ScratchScope pool(c);
    lqcd_spinor_s *r = pool.get(op->kind, mine), *p = pool.get(op->kind, mine), *q = pool.get(op->kind, mine), *t = pool.get(op->kind, other);
    if (!(r && p && q && t)) return LQCD_ERR_HIP;
    lqcd_spinor_s xv = *x, bv = *b;
    xv.subset = bv.subset = mine;
    xv.elems = bv.elems = nh;
    xv.data = x->data + (size_t)parity * nh;
    bv.data = b->data + (size_t)parity * nh;
    lqcd_spinor_s vin = xv, vout = xv;
    const double m2 = op->km * op->km;
    ApplyFn A = [&](double2* out, const double2* in) -> int {
        vin.data = const_cast<double2*>(in);
        vout.data = out;
        StencilCall s1 = make_hop_call(op, t, &vin, nullptr, 0.0, 1.0, 0);          // t = H in (other parity)
        LQCHK(stencil_apply(c, s1));
        StencilCall s2 = make_hop_call(op, &vout, t, &vin, m2, -1.0, 0);            // out = m^2 in - H t
        return stencil_apply(c, s2);
    };
    return cg_generic(c, A, nh, xv.data, bv.data, r->data, p->data, q->data, eps, maxiter, iters, final_rr);
}

extern "C" int lqcd_solve_bicgstab(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters,
                                   double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_bicgstab"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    ScratchScope pool(c);
    double2* wd[6];
    for (int i = 0; i < 6; i++) {
        lqcd_spinor_s* wi = pool.get(op->kind, LQCD_FULL);
        if (!wi) return LQCD_ERR_HIP;
        wd[i] = wi->data;
    }
    // the stencil works on spinor handles; wrap raw pointers of the scratch fields
    lqcd_spinor_s vin = *x, vout = *x;
    ApplyFn A = [&](double2* out, const double2* in) -> int {
        vin.data = const_cast<double2*>(in);
        vout.data = out;
        return op_apply_async(op, &vout, &vin, dagger ? 1 : 0, nullptr);
    };
    return bicgstab_core(c, A, x->elems, x->data, b->data, wd, eps, maxiter, iters, final_rr);
}

// BiCG (`bicg`, the default method_CG of solve_DinvX!(y, D, x): SURVEY.md 3.3): coupled recurrences with A and A^+, shadow residual
// r~_0 = r_0, stopping rule real(r.r) < eps.  Offered for completeness of the reference's solver list: the scalars go through the host
// (three synchronising reductions per iteration); the hot paths use the device-scalar CG / BiCGStab above.  x holds the initial guess.
extern "C" int lqcd_solve_bicg(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters, double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_bicg"));
    ARGCHK(maxiter >= 0, "lqcd_solve_bicg: maxiter < 0");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    ScratchScope pool(c);
    lqcd_spinor_s* w[6];
    for (auto& f : w) { f = pool.get(op->kind, LQCD_FULL); if (!f) return LQCD_ERR_HIP; }
    lqcd_spinor_s *r = w[0], *rt = w[1], *p = w[2], *pt = w[3], *q = w[4], *qt = w[5];
    const size_t n = x->elems, bytes = n * sizeof(double2);
    const int dg = dagger ? 1 : 0;
    LQCHK(op_apply_async(op, q, x, dg, nullptr));
    HIPCHK(hipMemcpyAsync(r->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
    LQCHK(blas_axpy(c, -1.0, 0.0, q->data, r->data, n));
    for (lqcd_spinor_s* f : {rt, p, pt}) HIPCHK(hipMemcpyAsync(f->data, r->data, bytes, hipMemcpyDeviceToDevice, c->stream));
    double rr = 0, im = 0;
    LQCHK(blas_norm2(c, r->data, n, &rr, true));
    std::complex<double> rho(rr, 0.0);      // <r~, r> with r~ = r
    int it = 0;
    bool converged = rr < eps;
    while (!converged && it < maxiter) {
        it++;
        LQCHK(op_apply_async(op, q, p, dg, nullptr));
        LQCHK(op_apply_async(op, qt, pt, 1 - dg, nullptr));
        double dr = 0, di = 0;
        LQCHK(blas_dot(c, pt->data, q->data, n, &dr, &di, true));
        const std::complex<double> alpha = rho / std::complex<double>(dr, di);
        if (!std::isfinite(alpha.real()) || !std::isfinite(alpha.imag())) { set_error("BiCG: breakdown (<p~, A p> = 0)"); return LQCD_ERR_NOT_CONVERGED; }
        LQCHK(blas_axpy(c, alpha.real(), alpha.imag(), p->data, x->data, n));
        LQCHK(blas_axpy(c, -alpha.real(), -alpha.imag(), q->data, r->data, n));
        LQCHK(blas_axpy(c, -alpha.real(), alpha.imag(), qt->data, rt->data, n));        // r~ -= conj(alpha) A^+ p~
        LQCHK(blas_norm2(c, r->data, n, &rr, true));
        if (rr < eps) { converged = true; break; }
        LQCHK(blas_dot(c, rt->data, r->data, n, &dr, &im, true));
        const std::complex<double> rho1(dr, im), beta = rho1 / rho;
        LQCHK(blas_axpby(c, 1.0, 0.0, r->data, beta.real(), beta.imag(), p->data, n));          // p = r + beta p
        LQCHK(blas_axpby(c, 1.0, 0.0, rt->data, beta.real(), -beta.imag(), pt->data, n));       // p~ = r~ + conj(beta) p~
        rho = rho1;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    if (iters) *iters = it;
    if (final_rr) *final_rr = rr;
    if (!converged) {
        set_error("The BiCG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
        return LQCD_ERR_NOT_CONVERGED;
    }
    return LQCD_OK;
}

// even-odd (Schur) preconditioned BiCGStab, Wilson:
//   (1 - k^2 H_eo H_oe) x_e = b_e + k H_eo b_o ;  x_o = b_o + k H_oe x_e
// Wilson-clover (D_sw = A - k H, A block diagonal in parity): with the packed inverse blocks A^-1 (clover.hip)
//   (1 - k^2 A_ee^-1 H_eo A_oo^-1 H_oe) x_e = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o) ;  x_o = A_oo^-1 (b_o + k H_oe x_e)
// (D_sw^+: H -> H^+ through the dagger flag of the hop, A is Hermitian).
extern "C" int lqcd_solve_bicgstab_eo(lqcd_op_t op, lqcd_spinor_t x, lqcd_spinor_t b, int dagger, double eps, int maxiter, int* iters,
                                      double* final_rr) {
    LQCHK(check_full(op, x, b, "lqcd_solve_bicgstab_eo"));
    ARGCHK(op->kind == LQCD_WILSON, "lqcd_solve_bicgstab_eo: Wilson only");
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    apply_bc(c, op->bc);
    const bool clov = op->csw != 0.0 && op->clover;
    if (clov) {     // A follows the links, A^-1 follows A
        if (op->clover_version != op->gauge->version) {
            LQCHK(clover_build(c, op->gauge, op->clover, op->km, op->csw));
            op->clover_version = op->gauge->version;
        }
        if (!op->clover_inv) HIPCHK(hipMalloc((void**)&op->clover_inv, clover_elems(c->geom) * sizeof(double2)));
        if (op->clover_inv_version != op->clover_version) {
            LQCHK(clover_invert(c, op->clover, op->clover_inv));
            op->clover_inv_version = op->clover_version;
        }
    }
    const double2* Ai = op->clover_inv;
    const double k = op->km;
    const int dg = dagger ? 1 : 0;
    const size_t nh = x->elems / 2;
    ScratchScope pool(c);
    double2* wd[6];
    lqcd_spinor_s* wsp[6];
    for (int i = 0; i < 6; i++) {
        lqcd_spinor_s* wi = pool.get(op->kind, LQCD_EVEN);
        if (!wi) return LQCD_ERR_HIP;
        wd[i] = wi->data;
        wsp[i] = wi;
    }
    // plain Wilson r = 1 on an unpartitioned lattice: the chain whose inner products come from the Schur operator's epilogue (bicgstab_eo_wilson)
    const bool fused = c->tun.bicg_fused >= 1 && op->r == 1.0 && c->tun.dslash_variant == 1 && !any_partitioned(c) && !c->has_comm && c->geom.Vh % 64 == 0;
    lqcd_spinor_s* rhs = pool.get(op->kind, LQCD_EVEN);
    lqcd_spinor_s* te = clov ? pool.get(op->kind, LQCD_EVEN) : nullptr;
    lqcd_spinor_s* to = pool.get(op->kind, LQCD_ODD);
    lqcd_spinor_s* uo = clov ? pool.get(op->kind, LQCD_ODD) : nullptr;
    if (!rhs || !to || (clov && (!te || !uo))) return LQCD_ERR_HIP;
    // views of the even/odd halves of b and x
    lqcd_spinor_s be = *b, bo = *b, xe = *x, xo = *x;
    be.subset = xe.subset = LQCD_EVEN; bo.subset = xo.subset = LQCD_ODD;
    be.elems = bo.elems = xe.elems = xo.elems = nh;
    bo.data = b->data + nh; xo.data = x->data + nh;
    int st = LQCD_OK;
    auto run = [&]() -> int {
        lqcd_spinor_s vin = xe, vout = xe;
        ApplyFn A;
        if (!clov) {
            // rhs = b_e + k H_eo b_o
            { StencilCall s = make_hop_call(op, rhs, &bo, &be, 1.0, k, dg); LQCHK(stencil_apply(c, s)); }
            A = [&](double2* out, const double2* in) -> int {
                vin.data = const_cast<double2*>(in);
                vout.data = out;
                StencilCall s1 = make_hop_call(op, to, &vin, nullptr, 0.0, 1.0, dg);         // t_o = H_oe in
                LQCHK(stencil_apply(c, s1));
                StencilCall s2 = make_hop_call(op, &vout, to, &vin, 1.0, -k * k, dg);        // out = in - k^2 H_eo t_o
                return stencil_apply(c, s2);
            };
        } else {
            // rhs = A_ee^-1 (b_e + k H_eo A_oo^-1 b_o)
            LQCHK(clover_apply_parity(c, Ai, 1, uo->data, bo.data, 1.0, nullptr, 0.0));
            { StencilCall s = make_hop_call(op, te, uo, &be, 1.0, k, dg); LQCHK(stencil_apply(c, s)); }
            LQCHK(clover_apply_parity(c, Ai, 0, rhs->data, te->data, 1.0, nullptr, 0.0));
            A = [&](double2* out, const double2* in) -> int {
                vin.data = const_cast<double2*>(in);
                StencilCall s1 = make_hop_call(op, to, &vin, nullptr, 0.0, 1.0, dg);         // t_o = H_oe in
                LQCHK(stencil_apply(c, s1));
                LQCHK(clover_apply_parity(c, Ai, 1, uo->data, to->data, 1.0, nullptr, 0.0)); // u_o = A_oo^-1 t_o
                StencilCall s2 = make_hop_call(op, te, uo, nullptr, 0.0, 1.0, dg);           // t_e = H_eo u_o
                LQCHK(stencil_apply(c, s2));
                return clover_apply_parity(c, Ai, 0, out, te->data, -k * k, in, 1.0);        // out = in - k^2 A_ee^-1 t_e
            };
        }
        const bool mixed = fused && c->tun.bicg_mixed;                // fp32 inner chain, fp64 defect correction (mixed.hip); same contract
        const int sc = mixed ? bicgstab_eo_wilson_mixed(op, xe, rhs, wsp, to, dg, eps, maxiter, iters, final_rr, clov ? Ai : nullptr)
                     : fused ? bicgstab_eo_wilson(op, xe, rhs, wsp, to, dg, eps, maxiter, iters, final_rr, clov ? Ai : nullptr)
                             : bicgstab_core(c, A, nh, xe.data, rhs->data, wd, eps, maxiter, iters, final_rr);
        // the odd half (also on non-convergence, so x is a consistent best effort)
        if (!clov) {
            StencilCall s = make_hop_call(op, &xo, &xe, &bo, 1.0, k, dg);                    // x_o = b_o + k H_oe x_e
            LQCHK(stencil_apply(c, s));
        } else {
            StencilCall s = make_hop_call(op, to, &xe, &bo, 1.0, k, dg);
            LQCHK(stencil_apply(c, s));
            LQCHK(clover_apply_parity(c, Ai, 1, xo.data, to->data, 1.0, nullptr, 0.0));      // x_o = A_oo^-1 (b_o + k H_oe x_e)
        }
        return sc;
    };
    st = run();
    hipError_t e = hipStreamSynchronize(c->stream);      // before the scratch fields go back to the pool
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync bicgstab_eo", __FILE__, __LINE__);
    return st;
}

// ---------------------------------------------------------------------------------- multi-shift CG (RHMC solver)
namespace lqcd {
// per-shift coefficient block in device memory: [sigma | zeta_{n-1} | zeta_n | a | b | z] (ns doubles each), alpha_{n-1}, beta_{n-1}
// zeta recurrence (Jegerlehner hep-lat/9612014) after the base system's alpha_n, beta_n are known:
//   zeta_{n+1} = zeta_n zeta_{n-1} alpha_{n-1} / (zeta_{n-1} alpha_{n-1} (1 + alpha_n sigma) + alpha_n beta_{n-1} (zeta_{n-1} - zeta_n))
//   x_j += (zeta_{n+1}/zeta_n) alpha_n p_j ;  p_j = (zeta_{n+1}/zeta_n)^2 beta_n p_j + zeta_{n+1} r
// stop_when_frozen: the base system only drives the Krylov space (no unshifted solution is wanted): once every shift is frozen the
// solve is complete -- S_DONE is raised here, the update kernel behind this launch applies the last x_j steps, later launches are no-ops.
__global__ void ms_zeta(double* __restrict__ sc, double* __restrict__ ms, int ns, int stop_when_frozen) {
    __shared__ int active;
    if (threadIdx.x == 0) active = 0;
    __syncthreads();
    if (sc[S_XDONE] != 0.0) return;
    const double alpha = sc[S_ALPHA], beta = sc[S_BETA], alpha_m = ms[6 * ns], beta_m = ms[6 * ns + 1];
    for (int j = threadIdx.x; j < ns; j += blockDim.x) {
        const double sigma = ms[j], zm = ms[ns + j], z0 = ms[2 * ns + j];
        if (fabs(z0) < 1e-100) {      // this shift converged long ago (its residual is zeta^2 |r|^2): freeze it before zeta underflows to 0/0
            ms[3 * ns + j] = 0.0; ms[4 * ns + j] = 0.0; ms[5 * ns + j] = 0.0;
            continue;
        }
        const double den = zm * alpha_m * (1.0 + alpha * sigma) + alpha * beta_m * (zm - z0);
        const double zp = z0 * zm * alpha_m / den, ratio = zp / z0;
        ms[3 * ns + j] = ratio * alpha;
        if (zp * zp * sc[S_RR] < sc[S_EPS]) {
            // the residual of this shift, zeta^2 |r|^2, is below the target once x_j has taken this step: last update, then the
            // shift is frozen (p_j = 0, no further traffic) -- large shifts drop out after a few tens of iterations
            ms[4 * ns + j] = 0.0; ms[5 * ns + j] = 0.0; ms[ns + j] = 0.0; ms[2 * ns + j] = 0.0;
            continue;
        }
        ms[4 * ns + j] = ratio * ratio * beta;
        ms[5 * ns + j] = zp;
        ms[ns + j] = z0;
        ms[2 * ns + j] = zp;
        active = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ms[6 * ns] = alpha; ms[6 * ns + 1] = beta;
        if (stop_when_frozen && !active) sc[S_DONE] = 1.0;
    }
}
int ms_zeta_launch(lqcd_ctx_s* c, double* d_ms, int ns, int stop_when_frozen) {
    hipLaunchKernelGGL(ms_zeta, dim3(1), dim3(64), 0, c->stream, c->d_scal, d_ms, ns, stop_when_frozen);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
// base system (x += alpha p ; p = r + beta p) and every active shifted system j (x_j += a_j p_j ; p_j = b_j p_j + z_j r) in one pass:
// r is read once per element, frozen shifts cost nothing.  x is still updated in the iteration that converges; nothing is touched
// afterwards.
// NT (tunable nt_blas): the shifted x_j, p_j are streamed -- an element is not touched again before 2 x (active shifts) fields have gone by
template <bool NT>
__global__ __launch_bounds__(UB) void ms_update_all(const double* __restrict__ sc, const double* __restrict__ ms, double2* const* __restrict__ ptr,
                                                     double2* __restrict__ x0, double2* __restrict__ p0, const double2* __restrict__ r, size_t n,
                                                     int ns) {
    if (sc[S_XDONE] != 0.0) return;
    const double al = sc[S_ALPHA], be = sc[S_BETA];
    for (size_t i = (size_t)blockIdx.x * UB + threadIdx.x; i < n; i += (size_t)gridDim.x * UB) {
        const double2 rv = r[i];
        {
            double2 pv = p0[i];
            if (x0) {          // the unshifted solution is optional (a rational action only wants the shifted ones)
                double2 xv = ldx<NT>(x0 + i);
                xv.x = fma(al, pv.x, xv.x); xv.y = fma(al, pv.y, xv.y);
                stx<NT>(x0 + i, xv);
            }
            pv.x = fma(be, pv.x, rv.x); pv.y = fma(be, pv.y, rv.y);
            p0[i] = pv;
        }
        for (int j = 0; j < ns; j++) {
            const double a = ms[3 * ns + j], bb = ms[4 * ns + j], z = ms[5 * ns + j];
            if (a == 0.0 && bb == 0.0 && z == 0.0) continue;       // frozen shift
            double2* __restrict__ x = ptr[j];
            double2* __restrict__ p = ptr[ns + j];
            double2 pv = ldx<NT>(p + i), xv = ldx<NT>(x + i);
            xv.x = fma(a, pv.x, xv.x); xv.y = fma(a, pv.y, xv.y);
            pv.x = fma(bb, pv.x, z * rv.x); pv.y = fma(bb, pv.y, z * rv.y);
            stx<NT>(x + i, xv); stx<NT>(p + i, pv);
        }
    }
}
}  // namespace lqcd

// (D^+D + sigma_j) x_j = b for all j < ns, plus the unshifted solution x0 (may be NULL): one Krylov space, the shifted
// iterates follow from the zeta recurrences, which run on the device next to the CG scalars (no host round trip inside an
// iteration; the host polls the convergence flag every 8 iterations).  Zero initial guesses.  Stops when |r|^2 < eps
// (for sigma_j >= 0 every |zeta_j| <= 1, so the shifted residuals zeta_j r are then below eps as well).
extern "C" int lqcd_solve_multishift_cg(lqcd_op_t op, lqcd_spinor_t x0, lqcd_spinor_t* xs, lqcd_spinor_t b, const double* sigma, int ns,
                                        double eps, int maxiter, int* iters, double* final_rr) {
    LQCHK(lqcd::links_flush_of(op));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(op && b && ns >= 0 && ns <= 1024 && (ns == 0 || (xs && sigma)), "lqcd_solve_multishift_cg: null argument or more than 1024 shifts");
    ARGCHK(b->ctx == op->ctx && b->kind == op->kind && b->subset == LQCD_FULL, "lqcd_solve_multishift_cg: b must be a FULL spinor of the operator");
    for (int j = 0; j < ns; j++) {
        ARGCHK(xs[j] && xs[j]->ctx == op->ctx && xs[j]->kind == op->kind && xs[j]->subset == LQCD_FULL && xs[j] != b,
               "lqcd_solve_multishift_cg: xs[j] must be distinct FULL spinors of the operator");
        ARGCHK(sigma[j] >= 0.0, "lqcd_solve_multishift_cg: shifts must be non-negative");
        ARGCHK(xs[j] != x0, "lqcd_solve_multishift_cg: xs[j] and x0 must be different fields (every system is updated in place on its own handle)");
        for (int i = 0; i < j; i++) ARGCHK(xs[i] != xs[j], "lqcd_solve_multishift_cg: the xs[j] must be pairwise different fields");
    }
    if (x0) LQCHK(check_full(op, x0, b, "lqcd_solve_multishift_cg"));
    lqcd_ctx_s* c = op->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t n = b->elems, bytes = n * sizeof(double2);
    lqcd_spinor_s* xbase = x0;          // may stay null: then the base system only drives the Krylov space
    lqcd_spinor_s* r = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* p = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* q = scratch_get(c, op->kind, LQCD_FULL);
    lqcd_spinor_s* tmp = scratch_get(c, op->kind, LQCD_FULL);
    std::vector<lqcd_spinor_s*> ps(ns, nullptr);
    bool ok = r && p && q && tmp;
    for (int j = 0; j < ns && ok; j++) { ps[j] = scratch_get(c, op->kind, LQCD_FULL); ok = ps[j] != nullptr; }
    const size_t ms_doubles = 6 * (size_t)ns + 2, ms_bytes = ms_doubles * sizeof(double) + 2 * (size_t)ns * sizeof(double2*);
    char* d_blk = nullptr;
    if (ok && hipMalloc((void**)&d_blk, ms_bytes) != hipSuccess) ok = false;
    auto release = [&]() {
        scratch_put(r); scratch_put(p); scratch_put(q); scratch_put(tmp);
        for (auto* s : ps) scratch_put(s);
        if (d_blk) (void)hipFree(d_blk);
    };
    if (!ok) { release(); set_error("lqcd_solve_multishift_cg: out of device memory"); return LQCD_ERR_HIP; }
    double* d_ms = (double*)d_blk;
    double2** d_ptr = (double2**)(d_blk + ms_doubles * sizeof(double));
    auto run = [&]() -> int {
        if (xbase) HIPCHK(hipMemsetAsync(xbase->data, 0, bytes, c->stream));
        HIPCHK(hipMemcpyAsync(r->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(p->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
        std::vector<double> hms(ms_doubles, 1.0);    // zeta_{-1} = zeta_0 = 1, alpha_{-1} = 1
        std::vector<double2*> hptr(2 * (size_t)ns);
        for (int j = 0; j < ns; j++) {
            HIPCHK(hipMemsetAsync(xs[j]->data, 0, bytes, c->stream));
            HIPCHK(hipMemcpyAsync(ps[j]->data, b->data, bytes, hipMemcpyDeviceToDevice, c->stream));
            hms[j] = sigma[j];
            hptr[j] = xs[j]->data;
            hptr[ns + j] = ps[j]->data;
        }
        hms[6 * (size_t)ns + 1] = 0.0;               // beta_{-1} = 0
        HIPCHK(hipMemcpyAsync(d_ms, hms.data(), ms_doubles * sizeof(double), hipMemcpyHostToDevice, c->stream));
        if (ns) HIPCHK(hipMemcpyAsync(d_ptr, hptr.data(), 2 * (size_t)ns * sizeof(double2*), hipMemcpyHostToDevice, c->stream));
        double rr = 0.0;
        LQCHK(blas_norm2(c, r->data, n, &rr, true));
        double init[9] = {rr, 0, 0, 0, 0, 0, eps, 0, 0};   // S_RR .. S_XDONE
        HIPCHK(hipMemcpyAsync(c->d_scal + S_RR, init, sizeof(init), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        int it = 0;
        bool converged = rr < eps;
        LQCHK(halo_schedule_settle(op));
        const int nbs = stencil_num_partials(c, op->kind, op->r, 2, 0, op_fused_clover(op)), nbu = stream_grid(c, n), check_every = 8;
        while (!converged && it < maxiter) {
            const int burst = std::min(check_every, maxiter - it);
            for (int k = 0; k < burst; k++) {
                // tmp = D p, alpha = rr / |tmp|^2 ; r -= alpha D^+ tmp in the stencil epilogue, beta = rr'/rr
                LQCHK(op_apply_async(op, tmp, p, 0, c->d_partial, c->d_scal));
                LQCHK(reduce_to_slot(c, nbs, 1, S_PQ, true, 1));
                apply_bc(c, op->bc);
                StencilCall s2;
                LQCHK(make_full_call(op, q, tmp, 1, s2));
                s2.norm_partial = c->d_partial;
                s2.upd_scal = c->d_scal;
                s2.upd[0] = spinor_block(r, 0);
                s2.upd[1] = spinor_block(r, 1);
                LQCHK(stencil_apply(c, s2));
                LQCHK(reduce_to_slot(c, nbs, 1, S_RRNEW, true, 2));
                if (ns) LQCHK(ms_zeta_launch(c, d_ms, ns, 0));
                if (c->tun.nt_blas) hipLaunchKernelGGL(ms_update_all<true>, dim3(nbu), dim3(UB), 0, c->stream, c->d_scal, d_ms, d_ptr, xbase ? xbase->data : (double2*)nullptr, p->data,
                                   r->data, n, ns);
                else hipLaunchKernelGGL(ms_update_all<false>, dim3(nbu), dim3(UB), 0, c->stream, c->d_scal, d_ms, d_ptr, xbase ? xbase->data : (double2*)nullptr, p->data,
                                   r->data, n, ns);
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            rr = c->h_scal[S_RR - S_RR];
            it = (int)c->h_scal[S_ITERS - S_RR];
            if (c->h_scal[S_DONE - S_RR] != 0.0) converged = true;
            if (!std::isfinite(rr)) { set_error("multi-shift CG: residual is not finite"); break; }
        }
        if (iters) *iters = it;
        if (final_rr) *final_rr = rr;
        if (!converged) {
            if (std::isfinite(rr))
                set_error("The shifted CG is not converged! maxsteps = " + std::to_string(maxiter) + ", residual = " + std::to_string(rr));
            return LQCD_ERR_NOT_CONVERGED;
        }
        return LQCD_OK;
    };
    const int st = run();
    (void)hipStreamSynchronize(c->stream);
    release();
    return st;
}
