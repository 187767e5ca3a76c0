// ops_internal.h -- declarations shared by apply.hip / solvers.hip / actions.hip / bench_api.hip / mdom.hip (the former ops.hip).
#pragma once
#include "lqcd_internal.h"

#include <functional>
#include <vector>

namespace lqcd {

double2* spinor_block(lqcd_spinor_s* s, int p);
int stream_grid(lqcd_ctx_s* c, size_t n);

// apply.hip
StencilCall make_hop_call(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, lqcd_spinor_s* xin, double a, double b, int dagger);
int check_full(lqcd_op_s* op, lqcd_spinor_s* a, lqcd_spinor_s* b, const char* who);     // four-dimensional operators only
int make_full_call(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger, StencilCall& s);
void apply_bc(lqcd_ctx_s* c, const int bc[4]);
void split_general_r(const StencilCall& s, StencilCall& s1, StencilCall& s2);   // Wilson r != 1 on a partitioned lattice = two r = 1 calls

// solvers.hip
constexpr int UB = 256;     // block size of the solvers' streaming kernels
struct CgWork {
    lqcd_spinor_s *r, *p, *q, *tmp;
    uint64_t pack_epoch = 0; // value of the context's halo_epoch right after that pack
    bool p_packed = false;   // partitioned lattice, halo_fuse bit 1: the send buffers hold the faces of the current search direction (packed by the last x/p update)
    int form = -1;      // iteration form fixed at cg_setup (0 plain, 1 deferred x, 2 small-lattice): the tunables may change while a session is open
    int k = 0;          // iterations enqueued so far (parity selects the p buffer when the x update is deferred: p_k lives in p for even k, in q for odd k)
    int ring = 2;       // form 1: search-direction buffers in rotation (2: p, q; K > 2: p, q, more[0..K-3]; p_k lives in buffer k % K), fixed at cg_setup
    lqcd_spinor_s* more[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    lqcd_spinor_s* buf(int j) const { return j == 0 ? p : j == 1 ? q : more[j - 2]; }
};
int cg_work_get(lqcd_ctx_s* c, int kind, CgWork& w);      // the four work vectors of a solve from the context's scratch pool (the ring's extra buffers follow in cg_setup)
void cg_work_put(CgWork& w);
int cg_enqueue_iteration(lqcd_op_s* op, lqcd_spinor_s* x, CgWork& w);
int cg_flush_x(lqcd_op_s* op, lqcd_spinor_s* x, CgWork& w);    // applies a pending deferred x update (end of a window that stopped on an even iteration)
int cg_setup(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, CgWork& w, double eps, double* rr0);
// cg_persist.hip: a staggered CG on a launch-bound lattice as ONE launch (initial residual included; the work vectors lend their storage)
bool cg_persist_ok(lqcd_op_s* op);
int cg_persist_run(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, CgWork& w, double eps, int maxiter, int* iters, double* rr, bool* converged, bool* gave_up);
typedef std::function<int(double2* out, const double2* in)> ApplyFn;
// multi-shift coefficient step (zeta recurrences next to the CG scalars): d_ms = [sigma | zeta_{n-1} | zeta_n | a | b | z] (ns doubles each),
// alpha_{n-1}, beta_{n-1}; stop_when_frozen raises S_DONE once every shift has converged (no unshifted solution wanted)
int ms_zeta_launch(lqcd_ctx_s* c, double* d_ms, int ns, int stop_when_frozen);

// ---- pieces shared by the fp64 (solvers.hip) and the mixed-precision (mixed.hip) even-odd BiCGStab chains
#ifdef __HIPCC__
template <int NV>
__device__ inline void block_reduce_nv(double (&a)[NV], double* partial) {
    __shared__ double red[NV][UB / 64];
#pragma unroll
    for (int v = 0; v < NV; v++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a[v] += __shfl_down(a[v], off, 64);
        if ((threadIdx.x & 63) == 0) red[v][threadIdx.x >> 6] = a[v];
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double t = 0;
#pragma unroll
        for (int w = 0; w < UB / 64; w++) t += red[threadIdx.x][w];
        partial[blockIdx.x * NV + threadIdx.x] = t;
    }
}
// the NV sums of a producer's partials, the same bits in every thread of the workgroup: ONE wave loads and adds them (every wave doing so made the
// prologue 5-9 us of texture-path time for 1024 workgroups), the others take the result from LDS
template <int NV>
__device__ inline void block_sum_partials(const double* __restrict__ partial, int n, double (&out)[NV], bool soa = false) {
    __shared__ double sh[NV];
    if (threadIdx.x < 64) {
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const double t = sum_partials_small_nv(partial, n, NV, v, soa);
            if (threadIdx.x == 0) sh[v] = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; v++) out[v] = sh[v];
}
struct BicgF {
    double* sc;             // the context's device scalar block
    int rho_in, rho_out;    // slots of rho for this iteration and the next (equal when the scalar kernels do the steps)
    int fold;
    const double* pin;      // fold: the producer's partials ...
    int pin_n;              // ... of that many workgroups
    int pin_soa = 0;        // ... in the [value][workgroup] layout (the dot partials of the hops; the streaming kernels' own partials stay [workgroup][value])
    const double* pin2;     // bicgf_xr: the |s|^2 partials of bicgf_s
    int pin2_n;
    double* pout;           // this kernel's partials
    const double* pin3 = nullptr;   // bicg_fused = 4, bicgf_s: the |r'|^2 partials the merged update launch left when its recurrence was not to be trusted (B_UNSURE)
    int pin3_n = 0;
    double guard = 1e-6;    // bicgf_xrp_rec: trust the recurrence for |r'|^2 while it exceeds guard * |s|^2
    int cont = 0;           // bicgf32_p after a reliable update (mixed.hip): no stopping test, the chain is re-armed (B_DONE = 0, B_EPS = cont_eps)
    double cont_eps = 0.0;
};
#endif
int schur_wilson(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, lqcd_spinor_s* to, int dg, const double2* Ai = nullptr);      // out = (1 - k^2 [A^-1] H_eo [A^-1] H_oe) in on the even sites (fp64)
int bicgstab_eo_wilson(lqcd_op_s* op, lqcd_spinor_s& xe, lqcd_spinor_s* rhs, lqcd_spinor_s* const w[6], lqcd_spinor_s* to, int dg, double eps,
                       int maxiter, int* iters, double* final_rr, const double2* Ai = nullptr);
// mixed.hip: the same solve with an fp32 inner chain and fp64 defect correction (plain Wilson, 12-real links); outer: correction steps
int bicgstab_eo_wilson_mixed(lqcd_op_s* op, lqcd_spinor_s& xe, lqcd_spinor_s* rhs, lqcd_spinor_s* const w[6], lqcd_spinor_s* to, int dg, double eps,
                             int maxiter, int* iters, double* final_rr, const double2* Ai = nullptr);

// CG for a Hermitian positive operator given as an enqueue function (solvers.hip); x holds the initial guess, r, p, q: work of n elements
int cg_generic(lqcd_ctx_s* c, const ApplyFn& A, size_t n, double2* x, const double2* b, double2* r, double2* p, double2* q, double eps,
               int maxiter, int* iters, double* final_rr);
// solvers.hip: the fused tail of a CG iteration, x += alpha p ; p = r + beta p (device scalars; a no-op behind the converging iteration), on raw buffers
int cg_launch_update_xp(lqcd_ctx_s* c, double2* x, double2* p, const double2* r, size_t n);
// domainwall.hip: the five-dimensional operator behind the entry points of the four-dimensional ones
int dw_op_apply(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in, int dagger);
int dw_op_apply_DdagD(lqcd_op_s* op, lqcd_spinor_s* out, lqcd_spinor_s* in);
int dw_solve_cg(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, double eps, int maxiter, int* iters, double* rr);
int dw_sample(lqcd_op_s* op, lqcd_spinor_s* phi, lqcd_spinor_s* xi, double eps, int maxiter);
int dw_action(lqcd_op_s* op, lqcd_spinor_s* phi, lqcd_spinor_s* X, lqcd_spinor_s* Y, double eps, int maxiter, double* Sf, int* iters);
int dw_force(lqcd_op_s* op, lqcd_gauge_s* out, lqcd_spinor_s* phi, double eps, int maxiter, double* Sf, int* iters);

// scratch fields of one call: returned to the context's pool on every exit path
struct ScratchScope {
    lqcd_ctx_s* c;
    std::vector<lqcd_spinor_s*> held;
    explicit ScratchScope(lqcd_ctx_s* c_) : c(c_) {}
    ScratchScope(const ScratchScope&) = delete;
    ScratchScope& operator=(const ScratchScope&) = delete;
    lqcd_spinor_s* get(int kind, int subset) {
        lqcd_spinor_s* s = scratch_get(c, kind, subset);
        if (s) held.push_back(s);
        return s;
    }
    ~ScratchScope() { for (lqcd_spinor_s* s : held) scratch_put(s); }
};

// mdom.hip
int mdom_check(int n, lqcd_ctx_s* c0);

}  // namespace lqcd
