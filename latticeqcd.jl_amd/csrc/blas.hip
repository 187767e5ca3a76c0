// blas.hip -- BLAS-1 on fermion fields and deterministic two-stage reductions (SURVEY.md 8(a) a6:
// LinearAlgebra.dot / add_fermion! / clear_fermion! of LatticeDiracOperators.jl; call sites
// /root/reference/src/updates/standardHMC.jl:54, src/md/standardMD.jl:50-51).
// All kernels stream double2 (16 B/lane, 1 KiB per wave instruction) with a grid-stride loop; reductions write
// per-block partials that a single-block kernel sums in a fixed order (bit-reproducible run to run).
#include "lqcd_internal.h"

#include <cstring>

namespace lqcd {

constexpr int RB = 256;  // reduction / streaming block size

__device__ inline void block_reduce2(double a, double b, double* partial, int stride) {
    __shared__ double red[2][RB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0, sb = 0;
#pragma unroll
        for (int w = 0; w < RB / 64; w++) { sa += red[0][w]; sb += red[1][w]; }
        partial[blockIdx.x * stride] = sa;
        if (stride > 1) partial[blockIdx.x * stride + 1] = sb;
    }
}

__global__ __launch_bounds__(RB) void dot_kernel(const double2* __restrict__ a, const double2* __restrict__ b, size_t n,
                                                 double* partial) {
    double re = 0, im = 0;
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const double2 x = a[i], y = b[i];
        re = fma(x.x, y.x, re); re = fma(x.y, y.y, re);
        im = fma(x.x, y.y, im); im = fma(-x.y, y.x, im);
    }
    block_reduce2(re, im, partial, 2);
}

__global__ __launch_bounds__(RB) void norm2_kernel(const double2* __restrict__ a, size_t n, double* partial) {
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const double2 x = a[i];
        s = fma(x.x, x.x, s); s = fma(x.y, x.y, s);
    }
    block_reduce2(s, 0.0, partial, 1);
}

// sums nblocks partials of `nvals` interleaved values into scal[slot..slot+nvals) in a fixed order (1024 threads, each a
// strided partial sum, then a wave/LDS tree), optionally followed by a CG scalar step on the same thread (single rank):
//   op 1: alpha = rr / pq      op 2: beta = rr'/rr, rr = rr', iters++, done = rr' < eps     (flags: see ops.hip)
__global__ __launch_bounds__(FB) void reduce_final(const double* __restrict__ partial, int nblocks, int nvals, double* scal, int slot, int op, PeerRedArgs pr, int soa) {      // soa: [value][block] partials
    __shared__ double tot[PEER_RED_VALS];
    // pr.nranks > 0 (peer-mapped backend, comm.hip): the sum over the ranks happens HERE, in the wave that holds the local sums -- no all-reduce launch
    if (nblocks <= 1024) {      // small reductions: one wave, in the order the folded prologues use (sum_partials_small_nv) -- a latency chain of
        if (threadIdx.x < 64) { //  16 loads + one DPP tree instead of a 1024-thread tree with two barriers per value
            double mine = 0.0;
            for (int v = 0; v < nvals; v++) {
                const double t = sum_partials_small_nv(partial, nblocks, nvals, v, soa != 0);
                if (pr.nranks) { if (((int)threadIdx.x >> 3) == v) mine = t; }
                else if (threadIdx.x == 0) scal[slot + v] = t;
            }
            if (pr.nranks) {
                const double s = peer_allreduce_wave(pr, mine, nvals);
                for (int q = 0; q < nvals; q++) {
                    const double v = __shfl(s, 8 * q, 64);
                    if (threadIdx.x == 0) scal[slot + q] = v;
                }
            }
            if (op && threadIdx.x == 0) cg_scalar_step(scal, op);
        }
        return;
    }
    // every value's class sums first (all their loads in flight together), then the trees, ONE barrier, then the sixteen wave sums of a value added in sequence by one
    // thread per value: each value goes through exactly the additions of the one-value-at-a-time loop this replaces (same bits), in a third of the time for five values
    __shared__ double redv[PEER_RED_VALS][FB / 64];
    double cls[PEER_RED_VALS];
#pragma unroll
    for (int v = 0; v < PEER_RED_VALS; v++)
        if (v < nvals) cls[v] = sum_partials_class(partial, nblocks, nvals, v, (int)threadIdx.x, soa != 0);
#pragma unroll
    for (int v = 0; v < PEER_RED_VALS; v++)
        if (v < nvals) {
            const double w = shfl_tree_sum(cls[v]);
            if ((threadIdx.x & 63) == 0) redv[v][threadIdx.x >> 6] = w;
        }
    __syncthreads();
    if ((int)threadIdx.x < nvals) {
        double t = 0;
        for (int w = 0; w < FB / 64; w++) t += redv[threadIdx.x][w];
        if (pr.nranks) tot[threadIdx.x] = t; else scal[slot + threadIdx.x] = t;
    }
    __syncthreads();
    if (pr.nranks && threadIdx.x < 64) {
        const int k = (int)threadIdx.x >> 3;
        const double s = peer_allreduce_wave(pr, k < nvals ? tot[k] : 0.0, nvals);
        for (int q = 0; q < nvals; q++) {
            const double v = __shfl(s, 8 * q, 64);
            if (threadIdx.x == 0) scal[slot + q] = v;
        }
    }
    if (op && threadIdx.x == 0) cg_scalar_step(scal, op);
}

__global__ __launch_bounds__(RB) void axpy_kernel(double ar, double ai, const double2* __restrict__ x, double2* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const double2 xv = x[i];
        double2 yv = y[i];
        yv.x = fma(ar, xv.x, yv.x); yv.x = fma(-ai, xv.y, yv.x);
        yv.y = fma(ar, xv.y, yv.y); yv.y = fma(ai, xv.x, yv.y);
        y[i] = yv;
    }
}

__global__ __launch_bounds__(RB) void axpby_kernel(double ar, double ai, const double2* __restrict__ x, double br, double bi,
                                                   double2* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const double2 xv = x[i], yv = y[i];
        double2 r;
        r.x = ar * xv.x - ai * xv.y + br * yv.x - bi * yv.y;
        r.y = ar * xv.y + ai * xv.x + br * yv.y + bi * yv.x;
        y[i] = r;
    }
}

__global__ __launch_bounds__(RB) void scale_kernel(double ar, double ai, double2* __restrict__ x, size_t n) {
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < n; i += (size_t)gridDim.x * RB) {
        const double2 xv = x[i];
        x[i] = make_double2(ar * xv.x - ai * xv.y, ar * xv.y + ai * xv.x);
    }
}

int stream_grid(lqcd_ctx_s* c, size_t n) {
    size_t nb = (n + RB - 1) / RB;
    const size_t cap = (size_t)c->num_cu * 8;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

// sum over ranks of n host doubles (all-reduce through the device scalar block)
int allreduce_host(lqcd_ctx_s* c, double* vals, int n) {
    if (!c->has_comm) return LQCD_OK;
    ARGCHK(n <= 8, "allreduce_host: too many values");
    double* d = c->d_scal + SCAL_DOUBLES - 8;
    HIPCHK(hipMemcpyAsync(d, vals, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LQCHK(comm_allreduce(c, d, n));
    HIPCHK(hipMemcpyAsync(vals, d, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return comm_check(c);
}

// the part of reduce_to_slot behind the one-block sum, for a slot that the producing launch has already filled (the exterior kernel's last
// block, halo_fuse bit 0): all-reduce over the ranks and the CG's scalar step
int reduce_tail(lqcd_ctx_s* c, int nvals, int slot, int cg_op) {
    if (c->has_comm || cg_op) return comm_allreduce(c, c->d_scal + slot, nvals, cg_op);
    return LQCD_OK;
}

// device-side reduction of partials into d_scal[slot..], followed by an all-reduce when running on several ranks.  RCCL: reduce_final -> ncclAllReduce -> one-thread
// scalar step; peer-mapped backend: ONE launch (the reduction block adds the ranks' slots itself and does the scalar step)
int reduce_to_slot(lqcd_ctx_s* c, int nblocks, int nvals, int slot, bool allreduce, int cg_op, const double* partial, bool soa) {
    const bool multi = allreduce && c->has_comm;   // also at world size 1 (self-partition tests exercise the collective)
    const bool peer = multi && c->peer.on;
    ARGCHK(nvals >= 1 && nvals <= PEER_RED_VALS, "reduce_to_slot: one to eight values per reduction");
    if (peer) ARGCHK(nvals <= PEER_RED_VALS, "reduce_to_slot: more than 8 values in one reduction over the ranks");
    PeerRedArgs pr;
    if (peer) pr = comm_red_args(c); else memset(&pr, 0, sizeof pr);
    hipLaunchKernelGGL(reduce_final, dim3(1), dim3(FB), 0, c->stream, partial ? partial : c->d_partial, nblocks, nvals, c->d_scal, slot, (multi && !peer) ? 0 : cg_op, pr, soa ? 1 : 0);
    HIPCHK(hipGetLastError());
    if (multi && !peer) LQCHK(comm_allreduce(c, c->d_scal + slot, nvals, cg_op));
    return LQCD_OK;
}

// reduce_to_slot(nvals = 1, with the all-reduce) when the folded halo schedule left a pack launch waiting (StencilCall::defer_pack): both in ONE launch
// (stencil.hip wilson_pack_reduce: the same summation order as reduce_final, bit for bit)
int reduce_pack_to_slot(lqcd_ctx_s* c, int nblocks, int slot, int cg_op) {
    if (!c->has_waiting_pack) return reduce_to_slot(c, nblocks, 1, slot, true, cg_op);
    c->has_waiting_pack = false;
    const bool multi = c->has_comm, peer = multi && c->peer.on;
    LQCHK(launch_pack_reduce(c, *static_cast<StencilCall*>(c->waiting_pack), c->d_partial, nblocks, slot, (multi && !peer) ? 0 : cg_op));
    if (multi && !peer) LQCHK(comm_allreduce(c, c->d_scal + slot, 1, cg_op));
    return LQCD_OK;
}

static int fetch_slot(lqcd_ctx_s* c, int slot, int nvals, double* out) {
    HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + slot, nvals * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < nvals; i++) out[i] = c->h_scal[i];
    return LQCD_OK;
}

int blas_dot(lqcd_ctx_s* c, const double2* a, const double2* b, size_t n, double* re, double* im, bool allreduce) {
    HIPCHK(hipSetDevice(c->device));
    const int nb = stream_grid(c, n);
    hipLaunchKernelGGL(dot_kernel, dim3(nb), dim3(RB), 0, c->stream, a, b, n, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 2, 0, allreduce, 0));
    double v[2];
    LQCHK(fetch_slot(c, 0, 2, v));
    *re = v[0];
    *im = v[1];
    return LQCD_OK;
}

int blas_norm2(lqcd_ctx_s* c, const double2* a, size_t n, double* n2, bool allreduce) {
    HIPCHK(hipSetDevice(c->device));
    const int nb = stream_grid(c, n);
    hipLaunchKernelGGL(norm2_kernel, dim3(nb), dim3(RB), 0, c->stream, a, n, c->d_partial);
    HIPCHK(hipGetLastError());
    LQCHK(reduce_to_slot(c, nb, 1, 0, allreduce, 0));
    return fetch_slot(c, 0, 1, n2);
}

int blas_axpy(lqcd_ctx_s* c, double ar, double ai, const double2* x, double2* y, size_t n) {
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(axpy_kernel, dim3(stream_grid(c, n)), dim3(RB), 0, c->stream, ar, ai, x, y, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
int blas_axpby(lqcd_ctx_s* c, double ar, double ai, const double2* x, double br, double bi, double2* y, size_t n) {
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(axpby_kernel, dim3(stream_grid(c, n)), dim3(RB), 0, c->stream, ar, ai, x, br, bi, y, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
int blas_scale(lqcd_ctx_s* c, double ar, double ai, double2* x, size_t n) {
    HIPCHK(hipSetDevice(c->device));
    hipLaunchKernelGGL(scale_kernel, dim3(stream_grid(c, n)), dim3(RB), 0, c->stream, ar, ai, x, n);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

}  // namespace lqcd

using namespace lqcd;

static bool same_shape(lqcd_spinor_t a, lqcd_spinor_t b) {
    return a && b && a->ctx == b->ctx && a->kind == b->kind && a->subset == b->subset && a->elems == b->elems;
}

extern "C" int lqcd_dot(lqcd_spinor_t a, lqcd_spinor_t b, double* re, double* im) {
    ARGCHK(same_shape(a, b) && re && im, "lqcd_dot: shape mismatch or null");
    return blas_dot(a->ctx, a->data, b->data, a->elems, re, im, true);
}
extern "C" int lqcd_norm2(lqcd_spinor_t a, double* n2) {
    ARGCHK(a && n2, "lqcd_norm2: null");
    return blas_norm2(a->ctx, a->data, a->elems, n2, true);
}
extern "C" int lqcd_axpy(double ar, double ai, lqcd_spinor_t x, lqcd_spinor_t y) {
    ARGCHK(same_shape(x, y), "lqcd_axpy: shape mismatch");
    LQCHK(blas_axpy(x->ctx, ar, ai, x->data, y->data, x->elems));
    HIPCHK(hipStreamSynchronize(x->ctx->stream));
    return LQCD_OK;
}
extern "C" int lqcd_axpby(double ar, double ai, lqcd_spinor_t x, double br, double bi, lqcd_spinor_t y) {
    ARGCHK(same_shape(x, y), "lqcd_axpby: shape mismatch");
    LQCHK(blas_axpby(x->ctx, ar, ai, x->data, br, bi, y->data, x->elems));
    HIPCHK(hipStreamSynchronize(x->ctx->stream));
    return LQCD_OK;
}
extern "C" int lqcd_scale(double ar, double ai, lqcd_spinor_t x) {
    ARGCHK(x, "lqcd_scale: null");
    LQCHK(blas_scale(x->ctx, ar, ai, x->data, x->elems));
    HIPCHK(hipStreamSynchronize(x->ctx->stream));
    return LQCD_OK;
}
