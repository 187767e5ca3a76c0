// cg_persist.hip -- launch-bound lattices (BASELINE configs[1]: 8^4 staggered CG to 1e-10): the whole CG on D^+D in ONE launch.
//
// Replaces, for lattices of at most 256 chunks of 64 sites, the launch chain of solvers.hip cg_enqueue_iteration (cg_small: three dependent
// ~5 us launches per iteration) behind the same entry point -- lqcd_solve_cg_DdagD, i.e. LatticeDiracOperators.jl's
// solve_DinvX!(y, DdagD, x) -> cg (reference call site /root/reference/src/md/AbstractMD.jl:129 through calc_UdSfdU!).
//
// Structure.  One 64-lane workgroup per chunk, one site per lane, every workgroup resident (grid <= 256 = one per CU).  What never changes
// during a solve stays on the CU: the eight 3x3 matrices a site applies to its neighbours -- W_f = (eta bc / 2) U_mu(n), W_b = -(eta bc / 2)
// U_mu(n-mu)^+ -- in 72 KiB of LDS, and the site's own x, r, p, t = D p in registers.  What neighbours need travels through global memory
// with agent-scope (write-through / coherent) accesses: p, t and r of every site.  An iteration has TWO grid-wide synchronisations, each of
// which also carries one all-reduce:
//   phase A   p' = r + beta p for the site itself and -- recomputed from the neighbours' r and old p, same fma, same bits -- for its
//             eight neighbours;  t = m p' + H p';  publish p', t;  |t|^2 partial                         -> barrier, alpha = rr / sum
//   phase B   q = m t - H t (D^+ = m - H) from the neighbours' t;  r -= alpha q;  x += alpha p';  publish r;  |r|^2 partial
//                                                                                                        -> barrier, beta = rr' / rr, test
// (forming the neighbours' search direction on the fly removes the third synchronisation a separate p update would need; p alternates
// between two buffers because a site's old p is still being read while its new one is written).
// Synchronisation is placement-independent (cdna_hip_programming.md, Guideline 16, the "sc1 payload -> vmcnt(0) -> flag" hand-off): data is
// published with relaxed agent-scope atomic stores (global_store ... sc1, write-through), every wave drains its stores (s_waitcnt vmcnt(0)),
// lane 0 adds 1 to one of eight monotonic counters with a relaxed agent-scope atomic and lanes 0..7 poll them with relaxed loads; consumers read published data
// with relaxed agent-scope atomic loads (global_load ... sc1).  No fences, no dependence on which XCD a workgroup runs on.  Every spin is
// bounded (wall clock, 50 ms): a workgroup that gives up leaves -- then every workgroup does --, x stays untouched and the host repeats the
// solve with the launch chain (and stops using this form on the context).
// Reductions: every workgroup sums the same <= 256 partials in the same order, so all of them take the same decisions.
#include "ops_internal.h"
#include "stencil_common.h"


namespace lqcd {

namespace {

struct PersistArgs {
    Geom g;
    const double2* gauge;
    double2* x[2];
    const double2* b[2];
    double2* r[2];
    double2* pa[2];      // search direction: two buffers, written alternately
    double2* pb[2];
    double2* t[2];
    double mass, eps;    // eps < 0: never converged (fixed-length window)
    double* scal;        // the solver's scalar block: S_RR, S_ITERS, S_DONE on exit
    double* part;        // [0, 256): |t|^2 partials, [256, 512): |r|^2 partials
    unsigned* ctr;       // ctr[32 k], k < 8: barrier counters (monotonic across launches)
    int nwg, nch, maxiter;
    int nexp;            // arrivals a barrier waits for: nwg (test hook cg_persist = 2: nwg + 1, so that every workgroup gives up)
    unsigned epoch0;
};

__device__ inline double ldc(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void stc(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline cd ld_c(const double2* p) { return mk(ldc(&p->x), ldc(&p->y)); }
__device__ inline void st_c(double2* p, cd v) { stc(&p->x, v.re); stc(&p->y, v.im); }

constexpr unsigned long long kSpinLimit = 5000000ull;      // wall_clock64 ticks of 10 ns: 50 ms
constexpr int kGaveUpWord = 8 * 32;      // ninth 128-B slot of the context's counter block (cgp_ctr): set by any workgroup whose wait ran out, zeroed by the host before a launch

// All workgroups of the launch.  Above 16 workgroups the arrivals are spread over EIGHT counters 128 bytes apart (workgroup b adds to counter
// b % 8: one memory-side atomic unit serialises ~12 ns per arrival -- 256 arrivals on one word cost 3 us per barrier); lanes 0..7 poll one
// counter each and the wave votes.  Up to 16 workgroups use counter 0 alone (one polling load instead of eight).  The counters are never
// reset: `epoch` = number of this barrier counted over ALL launches of this context (the host carries the count from launch to launch).
// false: gave up (error reported through the done flag).
template <bool SHARD>
__device__ inline bool grid_sync(unsigned* ctr, unsigned epoch, int nwg) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's published stores are acknowledged, its loads have landed
    const int lane = threadIdx.x;
    const int nc = SHARD ? 8 : 1;
    if (lane == 0) __hip_atomic_fetch_add(ctr + (SHARD ? 32 * (blockIdx.x & 7) : 0), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned mine = lane < nc ? (unsigned)(SHARD ? (nwg - lane + 7) / 8 : nwg) * epoch : 0u;      // arrivals this lane's counter must show
    const unsigned long long t0 = wall_clock64();
    int ok = 1;
    for (;;) {
        const unsigned v = lane < nc ? __hip_atomic_load(ctr + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__all((int)(v - mine) >= 0)) break;      // wrap-safe comparison
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kSpinLimit) { ok = 0; break; }
    }
    asm volatile("" ::: "memory");
    // giving up is GLOBAL: the workgroups that were late still pass this barrier (this one did arrive) and may finish the solve, so the word the
    // host reads is one any workgroup can raise, not one workgroup's view
    if (!__builtin_amdgcn_readfirstlane(ok) && lane == 0) __hip_atomic_store(ctr + kGaveUpWord, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// (wave_sum: DPP row exchanges + v_readlane, lqcd_internal.h)

// sum of the nwg (<= 256) published partials, identical in every workgroup
__device__ inline double sum_partials(const double* part, int nwg) {
    const int lane = threadIdx.x;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = lane + 64 * k;
        if (j < nwg) v += ldc(part + j);
    }
    return wave_sum(v);
}

// acc += W psi for one hop; W = 9 elements at w[e * 64] (this lane's column of the LDS block)
__device__ inline void hop_acc(cd (&acc)[3], const double2* w, const cd (&psi)[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        cd t = acc[a];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const double2 m = w[(a * 3 + b) * 64];
            t.re = fma(m.x, psi[b].re, t.re); t.re = fma(-m.y, psi[b].im, t.re);
            t.im = fma(m.x, psi[b].im, t.im); t.im = fma(m.y, psi[b].re, t.im);
        }
        acc[a] = t;
    }
}

template <bool SHARD>
__global__ __launch_bounds__(64) void cg_persist_staggered(PersistArgs a) {
    __shared__ double2 W[8 * 9 * 64];      // [hop = 2 mu + (0 forward | 1 backward)][element][lane]
    const int lane = threadIdx.x;
    const int p = (int)blockIdx.x / a.nch, chunk = (int)blockIdx.x % a.nch;
    const int i = chunk * 64 + lane;
    const Geom& g = a.g;
    Nbr n;
    int c[4];
    neighbours(g, p, i, n, c);
    const int Us = glink_stride(g), Ss = sp_stride(g);
    int noff[8];                           // element offset of component 0 of the neighbour, inside the other parity's block
    {
        int e = 0;
#pragma unroll
        for (int mu = 0; mu < 4; mu++) {
            const double eta = (e & 1) ? -1.0 : 1.0;      // eta_mu(n) = (-1)^(x_0 + .. + x_{mu-1})  (stencil.hip stag_eta)
            e += c[mu];
            cd u[9];
            load_link(u, a.gauge + glink_off(g, p, mu, i), Us);
            const double cf = 0.5 * eta * n.sf[mu];
#pragma unroll
            for (int k = 0; k < 9; k++) { double2 v; v.x = cf * u[k].re; v.y = cf * u[k].im; W[((2 * mu) * 9 + k) * 64 + lane] = v; }
            load_link(u, a.gauge + glink_off(g, 1 - p, mu, n.bwd[mu]), Us);
            const double cb = -0.5 * eta * n.sb[mu];
#pragma unroll
            for (int r = 0; r < 3; r++)
#pragma unroll
                for (int s = 0; s < 3; s++) {      // (U^+)[r][s] = conj(U[s][r])
                    double2 v; v.x = cb * u[s * 3 + r].re; v.y = -cb * u[s * 3 + r].im;
                    W[((2 * mu + 1) * 9 + r * 3 + s) * 64 + lane] = v;
                }
            noff[2 * mu] = (int)sp_off(3, n.fwd[mu]);
            noff[2 * mu + 1] = (int)sp_off(3, n.bwd[mu]);
        }
    }
    const size_t own = sp_off(3, i);
    const double2* W_l = W + lane;
    const double eps = a.eps;
    unsigned nbar = a.epoch0;      // barriers this context has executed in earlier launches
    bool ok = true, done = false;
    cd x[3], r[3], pv[3], t[3];
    // ---- initial residual r = b - D^+ D x in two phases of the same shape as the iteration's (x comes from an earlier launch: plain loads)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        x[k] = ld(a.x[p] + own + (size_t)k * Ss);
        t[k] = mk(a.mass * x[k].re, a.mass * x[k].im);
    }
#pragma unroll
    for (int h = 0; h < 8; h++) {
        cd xn[3];
#pragma unroll
        for (int k = 0; k < 3; k++) xn[k] = ld(a.x[1 - p] + noff[h] + (size_t)k * Ss);
        hop_acc(t, W_l + h * 9 * 64, xn);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) st_c(a.t[p] + own + (size_t)k * Ss, t[k]);
    ok = grid_sync<SHARD>(a.ctr, ++nbar, a.nexp);
    {
        cd tn[8][3], q[3];
#pragma unroll
        for (int h = 0; h < 8; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) tn[h][k] = ld_c(a.t[1 - p] + noff[h] + (size_t)k * Ss);
#pragma unroll
        for (int k = 0; k < 3; k++) q[k] = mk(0.0, 0.0);
#pragma unroll
        for (int h = 0; h < 8; h++) hop_acc(q, W_l + h * 9 * 64, tn[h]);
        double nr = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const cd bv = ld(a.b[p] + own + (size_t)k * Ss);
            r[k].re = bv.re - fma(a.mass, t[k].re, -q[k].re);
            r[k].im = bv.im - fma(a.mass, t[k].im, -q[k].im);
            pv[k] = r[k];
            st_c(a.r[p] + own + (size_t)k * Ss, r[k]);
            st_c(a.pa[p] + own + (size_t)k * Ss, r[k]);       // a finite "old p" for the first iteration's beta = 0
            nr = fma(r[k].re, r[k].re, nr); nr = fma(r[k].im, r[k].im, nr);
        }
        nr = wave_sum(nr);
        if (lane == 0) stc(a.part + 256 + blockIdx.x, nr);
    }
    if (ok) ok = grid_sync<SHARD>(a.ctr, ++nbar, a.nexp);
    double rr = 0.0, beta = 0.0;
    int it = -1;          // the initial residual plays the part of "iteration -1": its |r|^2 partials are summed at the top of the loop
    while (ok) {
        double2* const* pold = (it & 1) ? a.pa : a.pb;      // it = -1, 1, 3 ...: the new p of iteration it + 1 goes to pb
        double2* const* pnew = (it & 1) ? a.pb : a.pa;
        // ---- phase A: the neighbours' r and old p are requested before the reduction of the previous phase is read
        cd rn[4][3], po[4][3];
#pragma unroll
        for (int h = 0; h < 4; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                rn[h][k] = ld_c(a.r[1 - p] + noff[h] + (size_t)k * Ss);
                po[h][k] = ld_c(pold[1 - p] + noff[h] + (size_t)k * Ss);
            }
        {
            const double rrn = sum_partials(a.part + 256, a.nwg);
            beta = it < 0 ? 0.0 : rrn / rr;
            rr = rrn;
            it++;
            if (rr < eps) { done = true; break; }
            if (it >= a.maxiter) break;
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { pv[k].re = fma(beta, pv[k].re, r[k].re); pv[k].im = fma(beta, pv[k].im, r[k].im); }
#pragma unroll
        for (int k = 0; k < 3; k++) t[k] = mk(a.mass * pv[k].re, a.mass * pv[k].im);
#pragma unroll
        for (int h = 0; h < 4; h++) {
            cd pn[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { pn[k].re = fma(beta, po[h][k].re, rn[h][k].re); pn[k].im = fma(beta, po[h][k].im, rn[h][k].im); }
            hop_acc(t, W_l + h * 9 * 64, pn);
        }
#pragma unroll
        for (int h = 4; h < 8; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                rn[h - 4][k] = ld_c(a.r[1 - p] + noff[h] + (size_t)k * Ss);
                po[h - 4][k] = ld_c(pold[1 - p] + noff[h] + (size_t)k * Ss);
            }
#pragma unroll
        for (int h = 4; h < 8; h++) {
            cd pn[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { pn[k].re = fma(beta, po[h - 4][k].re, rn[h - 4][k].re); pn[k].im = fma(beta, po[h - 4][k].im, rn[h - 4][k].im); }
            hop_acc(t, W_l + h * 9 * 64, pn);
        }
        double nt = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            st_c(pnew[p] + own + (size_t)k * Ss, pv[k]);
            st_c(a.t[p] + own + (size_t)k * Ss, t[k]);
            nt = fma(t[k].re, t[k].re, nt); nt = fma(t[k].im, t[k].im, nt);
        }
        nt = wave_sum(nt);
        if (lane == 0) stc(a.part + blockIdx.x, nt);
        ok = grid_sync<SHARD>(a.ctr, ++nbar, a.nexp);
        if (!ok) break;
        // ---- phase B: the neighbours' t requested before the |t|^2 partials are read
        cd tn[8][3], q[3];
#pragma unroll
        for (int h = 0; h < 8; h++)
#pragma unroll
            for (int k = 0; k < 3; k++) tn[h][k] = ld_c(a.t[1 - p] + noff[h] + (size_t)k * Ss);
        const double alpha = rr / sum_partials(a.part, a.nwg);
#pragma unroll
        for (int k = 0; k < 3; k++) q[k] = mk(0.0, 0.0);
#pragma unroll
        for (int h = 0; h < 8; h++) hop_acc(q, W_l + h * 9 * 64, tn[h]);
        double nr = 0.0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double qre = fma(a.mass, t[k].re, -q[k].re), qim = fma(a.mass, t[k].im, -q[k].im);      // D^+ t = m t - H t
            r[k].re = fma(-alpha, qre, r[k].re); r[k].im = fma(-alpha, qim, r[k].im);
            x[k].re = fma(alpha, pv[k].re, x[k].re); x[k].im = fma(alpha, pv[k].im, x[k].im);
            st_c(a.r[p] + own + (size_t)k * Ss, r[k]);
            nr = fma(r[k].re, r[k].re, nr); nr = fma(r[k].im, r[k].im, nr);
        }
        nr = wave_sum(nr);
        if (lane == 0) stc(a.part + 256 + blockIdx.x, nr);
        ok = grid_sync<SHARD>(a.ctr, ++nbar, a.nexp);
    }
    // The solution goes to the t vector, not to x: whether the solve stands is decided for the whole grid by the host (kGaveUpWord), which then copies
    // it -- a solve that gave up anywhere leaves x exactly as it found it.  Nobody reads t any more: its last readers (phase B) are behind the last barrier.
    if (ok) {
#pragma unroll
        for (int k = 0; k < 3; k++) st(a.t[p] + own + (size_t)k * Ss, x[k]);
    }
    if (blockIdx.x == 0 && lane == 0) {
        a.scal[S_RR] = rr;
        a.scal[S_ITERS] = (double)(it < 0 ? 0 : it);
        a.scal[S_DONE] = !ok ? -1.0 : (done ? 1.0 : 0.0);      // -1: a synchronisation gave up (every workgroup then does)
        a.scal[S_PQ] = (double)(nbar - a.epoch0);               // barriers executed: the host's epoch count follows
    }
}

}  // namespace

// Does the one-launch CG apply?  Staggered operator on an unpartitioned lattice without a communicator, whole chunks, at most 256 of them (one
// workgroup per CU), and the tunables select the small-lattice form of the fused iteration (cg_fused = 2, cg_small = 1: what it replaces).
bool cg_persist_ok(lqcd_op_s* op) {
    lqcd_ctx_s* c = op->ctx;
    if (!c->tun.cg_persist || c->tun.cg_fused < 2 || !c->tun.cg_small || op->kind != LQCD_STAGGERED || any_partitioned(c) || c->has_comm || !c->local_peers.empty()) return false;
    const Geom& g = c->geom;
    if (g.Vh % 64 != 0) return false;
    const int nwg = 2 * (g.Vh / 64);
    if (nwg > std::min(256, c->num_cu)) return false;
    // every workgroup must be resident at once: the kernel's 72 KiB of LDS and its registers, asked of the runtime (once per process and form)
    static int per_cu[2] = {-1, -1};
    const int form = nwg > 16 ? 1 : 0;
    if (per_cu[form] < 0) {
        int nb = 0;
        hipError_t e = form ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cg_persist_staggered<true>, 64, 0)
                            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, cg_persist_staggered<false>, 64, 0);
        if (e != hipSuccess) { (void)hipGetLastError(); nb = 0; }
        per_cu[form] = nb;
    }
    return (long long)per_cu[form] * c->num_cu >= nwg;
}

// The whole solve: r = b - D^+D x, then the iterations, until r.r < eps (eps < 0: exactly maxiter iterations) or maxiter.  The work vectors only
// lend their storage (r, two p buffers, t); on return x is complete and rr / iteration count / done flag have been read back.
int cg_persist_run(lqcd_op_s* op, lqcd_spinor_s* x, lqcd_spinor_s* b, CgWork& w, double eps, int maxiter, int* iters, double* rr, bool* converged, bool* gave_up) {
    *gave_up = false;
    lqcd_ctx_s* c = op->ctx;
    apply_bc(c, op->bc);
    PersistArgs a;
    a.g = c->geom;
    a.gauge = op->gauge->data;
    for (int p = 0; p < 2; p++) {
        a.x[p] = spinor_block(x, p); a.b[p] = spinor_block(b, p); a.r[p] = spinor_block(w.r, p);
        a.pa[p] = spinor_block(w.p, p); a.pb[p] = spinor_block(w.q, p); a.t[p] = spinor_block(w.tmp, p);
    }
    a.mass = op->km;
    a.eps = eps;
    a.scal = c->d_scal;
    a.part = c->d_partial;
    a.ctr = c->cgp_ctr;
    a.nch = c->geom.Vh / 64;
    a.nwg = 2 * a.nch;
    a.maxiter = maxiter;
    a.nexp = a.nwg + (c->tun.cg_persist == 2 ? 1 : 0);
    // a launch whose grid is not the one the counters have counted so far (another lattice cannot share a context, but the flat / sharded
    // forms differ) or an epoch count near the wrap starts from zeroed counters
    if (c->cgp_nwg != a.nwg || c->cgp_epoch > 100000000u) {
        HIPCHK(hipMemsetAsync(a.ctr, 0, 9 * 32 * sizeof(unsigned), c->stream));
        c->cgp_epoch = 0; c->cgp_nwg = a.nwg;
    }
    a.epoch0 = c->cgp_epoch;
    HIPCHK(hipMemsetAsync(a.ctr + kGaveUpWord, 0, sizeof(unsigned), c->stream));
    if (a.nwg > 16) hipLaunchKernelGGL(cg_persist_staggered<true>, dim3(a.nwg), dim3(64), 0, c->stream, a);
    else hipLaunchKernelGGL(cg_persist_staggered<false>, dim3(a.nwg), dim3(64), 0, c->stream, a);
    HIPCHK(hipGetLastError());
    c->cgp_nwg = -1;          // until the barrier count of this launch is known the counters cannot be trusted
    unsigned gave = 1;
    HIPCHK(hipMemcpyAsync(c->h_scal, c->d_scal + S_RR, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&gave, a.ctr + kGaveUpWord, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (gave != 0 || c->h_scal[S_DONE - S_RR] < 0.0) {      // SOME workgroup waited 50 ms at a synchronisation: not all of them were resident (a busy GPU)
        *gave_up = true;                       // x is untouched (the result of the workgroups that did finish sits in the t vector and is dropped);
        return LQCD_OK;                        // the caller falls back to the launch chain and stops asking for this form
    }
    HIPCHK(hipMemcpyAsync(x->data, w.tmp->data, x->elems * sizeof(double2), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->cgp_epoch += (unsigned)c->h_scal[S_PQ - S_RR];
    c->cgp_nwg = a.nwg;
    *rr = c->h_scal[0];
    *iters = (int)c->h_scal[S_ITERS - S_RR];
    *converged = c->h_scal[S_DONE - S_RR] > 0.0;
    return LQCD_OK;
}

}  // namespace lqcd
