// force.hip -- pseudofermion force "U dS_f/dU" on the device (reference call site calc_UdSfdU!,
// /root/reference/src/md/AbstractMD.jl:129, wrapped by P_update_fermion! :120-135).
//
// S_f = eta^+ (D^+D)^-1 eta, X = (D^+D)^-1 eta, Y = D X  =>  delta S_f = -2 Re(Y^+ (delta D) X).  The output G_mu(n) is the
// general 3x3 matrix defined by   d/d eps S_f[U_mu(n) -> exp(i eps T) U_mu(n)] = -2 Im tr(T G_mu(n))   (Hermitian T):
//   Wilson:    G = kappa s [ sum_spin (U X(n+mu))_s ((r - g_mu) Y(n))_s^+  -  sum_spin X(n)_s (U (r + g_mu) Y(n+mu))_s^+ ]
//   staggered: G = -1/2 eta_mu(n) s [ (U X(n+mu)) Y(n)^+ + X(n) (U Y(n+mu))^+ ]
// s = boundary sign of the hop n -> n+mu.  X and Y stay resident after the solve; nothing goes back to the host.
//
// Mapping: workgroup = 64 consecutive checkerboard sites of one parity x 4 waves, wave = direction mu (wave-uniform gamma
// algebra, every load/store 64 lanes x 16 B contiguous in the chunk-blocked layouts).  HBM-bound: reads X, Y at n and n+mu
// (neighbour re-use through L2), the four links, writes four link-shaped matrices: 2*192 + 576 + 576 = 1536 B/site compulsory
// (Wilson).  Runs once per MD step, against hundreds of Dslash applications in the solve that precedes it.
//
// Partitioned lattice: the links of the upper face need X(n+mu), Y(n+mu) from the +mu neighbour.  One exchange step: the
// FULL X and Y spinors of the lower face (24 complex per site; any r) are packed, sent to the -mu neighbour (RCCL on the
// communication stream, or hipMemcpy in the in-process PE-grid emulation) and read by the sweep in place of the local
// neighbour; the rank on the global upper boundary applies the boundary sign.
#include "lqcd_internal.h"

#include <algorithm>

namespace lqcd {

struct FArgs {
    Geom g;
    const double2* gauge;
    double2* out;
    const double2* X[2];
    const double2* Y[2];
    const double2* ghost[4];   // received lower-face X|Y of the +mu neighbour: [parity of that site][X comps, Y comps][Fh]
    double2* send[4];
    double sign_fwd[4];        // partitioned directions: bc sign if this rank sits on the global upper boundary, else 1
    double coef;               // kappa (Wilson) or -1/2 (staggered)
    double r;
    int nc;                    // components per spinor: 12 | 3
    double scale;              // out = (acc ? out : 0) + scale * G   (sums over the poles of a rational action)
    int acc;
    BlockMap bm;               // workgroup -> (chunk, parity): the XCD tile sweep of the Dslash kernels (tunable md_remap), or plain order
};

// (g_MU psi)[S][c] for a full 4-spinor held in registers
template <int MU, int S>
__device__ __forceinline__ cd gamma_elem(const cd (&psi)[4][3], int c) {
    if constexpr (MU < 3) return mul_ipow<GK[MU][S]>(psi[PERM[MU][S]][c]);
    else return (S < 2) ? psi[S][c] : mk(-psi[S][c].re, -psi[S][c].im);
}

// out[s][c] = r psi[s][c] + SG (g_MU psi)[s][c]
template <int MU, int SG>
__device__ __forceinline__ void r_plus_gamma(cd (&out)[4][3], const cd (&psi)[4][3], double r) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const cd g0 = gamma_elem<MU, 0>(psi, c), g1 = gamma_elem<MU, 1>(psi, c), g2 = gamma_elem<MU, 2>(psi, c), g3 = gamma_elem<MU, 3>(psi, c);
        out[0][c] = mk(fma(r, psi[0][c].re, SG * g0.re), fma(r, psi[0][c].im, SG * g0.im));
        out[1][c] = mk(fma(r, psi[1][c].re, SG * g1.re), fma(r, psi[1][c].im, SG * g1.im));
        out[2][c] = mk(fma(r, psi[2][c].re, SG * g2.re), fma(r, psi[2][c].im, SG * g2.im));
        out[3][c] = mk(fma(r, psi[3][c].re, SG * g3.re), fma(r, psi[3][c].im, SG * g3.im));
    }
}

__device__ __forceinline__ void mv3(cd (&o)[3], const cd (&u)[9], const cd (&v)[3]) {
#pragma unroll
    for (int a = 0; a < 3; a++) {
        cd t = mk(0.0, 0.0);
        cfma(t, u[a * 3 + 0], v[0]);
        cfma(t, u[a * 3 + 1], v[1]);
        cfma(t, u[a * 3 + 2], v[2]);
        o[a] = t;
    }
}
// C[a][b] += sg * v[a] conj(w[b])
__device__ __forceinline__ void outer_acc(cd (&C)[9], const cd (&v)[3], const cd (&w)[3], double sg) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) {
            cd& t = C[a * 3 + b];
            t.re = fma(sg * v[a].re, w[b].re, t.re); t.re = fma(sg * v[a].im, w[b].im, t.re);
            t.im = fma(sg * v[a].im, w[b].re, t.im); t.im = fma(-sg * v[a].re, w[b].im, t.im);
        }
}

__device__ __forceinline__ void load_spinor4(cd (&psi)[4][3], const double2* __restrict__ base, int stride) {
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int c = 0; c < 3; c++) psi[s][c] = ld(base + (size_t)(s * 3 + c) * stride);
}

// where X(n+mu), Y(n+mu) of local site (p, c) live: the local neighbour, or the ghost buffer on a partitioned upper face
struct Nbr {
    const double2* X;
    const double2* Y;
    int stride;
    double sign;
};
__device__ __forceinline__ Nbr fwd_neighbour(const FArgs& k, int p, const int (&c)[4], int mu) {
    const Geom& g = k.g;
    Nbr n;
    if (c[mu] == g.L[mu] - 1 && g.part[mu]) {
        const int Fh = g.Vh / g.L[mu], f = coords_to_face(g, mu, c);
        n.X = k.ghost[mu] + (size_t)(1 - p) * 2 * k.nc * Fh + f;
        n.Y = n.X + (size_t)k.nc * Fh;
        n.stride = Fh;
        n.sign = k.sign_fwd[mu];
        return n;
    }
    int d[4] = {c[0], c[1], c[2], c[3]};
    n.sign = 1.0;
    if (++d[mu] == g.L[mu]) { d[mu] = 0; n.sign = g.bc_fwd[mu]; }
    const size_t off = sp_off(k.nc, coords_to_cb(g, d));
    n.X = k.X[1 - p] + off;
    n.Y = k.Y[1 - p] + off;
    n.stride = sp_stride(g);
    return n;
}

template <int MU, bool ACC>
__device__ __forceinline__ void wilson_force_site(const FArgs& k, int p, int i, const int (&c)[4]) {
    const Geom& g = k.g;
    const int Vs = sp_stride(g), Gs = glink_stride(g);
    const Nbr nb = fwd_neighbour(k, p, c, MU);
    const double2* __restrict__ Xn = k.X[p] + sp_off(12, i);
    const double2* __restrict__ Yn = k.Y[p] + sp_off(12, i);
    const size_t go = glink_off(g, p, MU, i);
    cd u[9], C[9];
#pragma unroll
    for (int e = 0; e < 9; e++) { u[e] = ld(k.gauge + go + (size_t)e * Gs); C[e] = mk(0.0, 0.0); }
    {   // + sum_s (U X(n+mu))_s ((r - g) Y(n))_s^+
        cd y[4][3], z[4][3];
        load_spinor4(y, Yn, Vs);
        r_plus_gamma<MU, -1>(z, y, k.r);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            cd x[3], w[3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) x[cc] = ld(nb.X + (size_t)(s * 3 + cc) * nb.stride);
            mv3(w, u, x);
            outer_acc(C, w, z[s], 1.0);
        }
    }
    {   // - sum_s X(n)_s (U (r + g) Y(n+mu))_s^+
        cd y[4][3], q[4][3];
        load_spinor4(y, nb.Y, nb.stride);
        r_plus_gamma<MU, 1>(q, y, k.r);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            cd x[3], w[3];
#pragma unroll
            for (int cc = 0; cc < 3; cc++) x[cc] = ld(Xn + (size_t)(s * 3 + cc) * Vs);
            mv3(w, u, q[s]);
            outer_acc(C, x, w, -1.0);
        }
    }
    const double f = k.coef * nb.sign * k.scale;
#pragma unroll
    for (int e = 0; e < 9; e++) {
        cd o = mk(f * C[e].re, f * C[e].im);
        if constexpr (ACC) { const cd old = ld(k.out + go + (size_t)e * Gs); o = mk(o.re + old.re, o.im + old.im); }
        st(k.out + go + (size_t)e * Gs, o);
    }
}

// ACC is a template parameter: as a run-time branch the nine read-modify-write pairs cost 90 VGPRs (occupancy 3 -> 1)
template <bool ACC>
__global__ __launch_bounds__(256) void wilson_force_kernel(FArgs k) {
    const Geom& g = k.g;
    int chunk, p;
    block_map(k.bm, blockIdx.x, chunk, p);
    const int i = chunk * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    switch (mu) {
    case 0: wilson_force_site<0, ACC>(k, p, i, c); break;
    case 1: wilson_force_site<1, ACC>(k, p, i, c); break;
    case 2: wilson_force_site<2, ACC>(k, p, i, c); break;
    default: wilson_force_site<3, ACC>(k, p, i, c); break;
    }
}

template <bool ACC>
__global__ __launch_bounds__(256) void staggered_force_kernel(FArgs k) {
    const Geom& g = k.g;
    const int p = blockIdx.x & 1, i = (blockIdx.x >> 1) * 64 + (threadIdx.x & 63), mu = threadIdx.x >> 6;
    if (i >= g.Vh) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    const int Vs = sp_stride(g), Gs = glink_stride(g);
    const Nbr nb = fwd_neighbour(k, p, c, mu);
    int e = 0;                                   // eta_mu(n) = (-1)^(x_0 + ... + x_{mu-1}), GLOBAL coordinates
    for (int nu = 0; nu < mu; nu++) e += c[nu] + g.origin[nu];
    const double f = k.coef * nb.sign * k.scale * ((e & 1) ? -1.0 : 1.0);
    const size_t go = glink_off(g, p, mu, i);
    cd u[9], C[9], xn[3], yn[3], xp[3], yp[3], ux[3], uy[3];
#pragma unroll
    for (int q = 0; q < 9; q++) { u[q] = ld(k.gauge + go + (size_t)q * Gs); C[q] = mk(0.0, 0.0); }
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
        xn[cc] = ld(k.X[p] + sp_off(3, i) + (size_t)cc * Vs);
        yn[cc] = ld(k.Y[p] + sp_off(3, i) + (size_t)cc * Vs);
        xp[cc] = ld(nb.X + (size_t)cc * nb.stride);
        yp[cc] = ld(nb.Y + (size_t)cc * nb.stride);
    }
    mv3(ux, u, xp);
    mv3(uy, u, yp);
    outer_acc(C, ux, yn, 1.0);
    outer_acc(C, xn, uy, 1.0);
#pragma unroll
    for (int q = 0; q < 9; q++) {
        cd o = mk(f * C[q].re, f * C[q].im);
        if constexpr (ACC) { const cd old = ld(k.out + go + (size_t)q * Gs); o = mk(o.re + old.re, o.im + old.im); }
        st(k.out + go + (size_t)q * Gs, o);
    }
}

// lower-face sites (x_mu = 0) of both parities: copy the full X and Y spinors into the send buffer of direction mu = blockIdx.y
__global__ __launch_bounds__(128) void force_pack_kernel(FArgs k) {
    const Geom& g = k.g;
    const int mu = blockIdx.y;
    if (!g.part[mu]) return;
    const int Fh = g.Vh / g.L[mu], Vs = sp_stride(g), nc = k.nc;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * Fh) return;
    const int ps = t / Fh, f = t - ps * Fh;
    int c[4];
    face_to_coords(g, mu, 0, ps, f, c);
    const size_t off = sp_off(nc, coords_to_cb(g, c));
    const double2* __restrict__ X = (ps ? k.X[1] : k.X[0]) + off;
    const double2* __restrict__ Y = (ps ? k.Y[1] : k.Y[0]) + off;
    double2* dst = k.send[mu] + (size_t)ps * 2 * nc * Fh + f;
    for (int q = 0; q < nc; q++) {
        dst[(size_t)q * Fh] = X[(size_t)q * Vs];
        dst[(size_t)(nc + q) * Fh] = Y[(size_t)q * Vs];
    }
}

static int force_buffers(lqcd_ctx_s* c) {
    if (c->force_ncomp) return LQCD_OK;
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        const size_t bytes = (size_t)2 * 24 * face_half_sites(c->geom, mu) * sizeof(double2);   // sized for Wilson (12 + 12 components)
        HIPCHK(hipMalloc((void**)&c->force_send[mu], bytes));
        HIPCHK(hipMalloc((void**)&c->force_recv[mu], bytes));
    }
    c->force_ncomp = 24;
    return LQCD_OK;
}
static size_t force_halo_elems(lqcd_ctx_s* c, int mu, int nc) { return (size_t)2 * 2 * nc * face_half_sites(c->geom, mu); }

static FArgs make_fargs(lqcd_ctx_s* c, int kind, const lqcd_gauge_s* U, lqcd_gauge_s* out, lqcd_spinor_s* X, lqcd_spinor_s* Y, double km, double r) {
    FArgs k;
    k.g = c->geom;
    k.gauge = U ? U->data : nullptr;
    k.out = out ? out->data : nullptr;
    for (int p = 0; p < 2; p++) { k.X[p] = spinor_block(X, p); k.Y[p] = spinor_block(Y, p); }
    for (int mu = 0; mu < 4; mu++) {
        k.ghost[mu] = c->force_recv[mu];
        k.send[mu] = c->force_send[mu];
        k.sign_fwd[mu] = (c->coord[mu] == c->pe[mu] - 1) ? c->geom.bc_fwd[mu] : 1.0;
    }
    k.coef = kind == LQCD_WILSON ? km : -0.5;
    k.r = r;
    k.nc = kind == LQCD_WILSON ? 12 : 3;
    k.scale = 1.0;
    k.acc = 0;
    k.bm = make_block_map(c->geom, c->tun.md_remap ? c->tun.xcd_remap : 0, c->tun.xcd_nsub, c->tun.xcd_ysplit);
    return k;
}

int launch_force_pack(lqcd_ctx_s* c, int kind, lqcd_spinor_s* X, lqcd_spinor_s* Y) {
    int maxf = 0;
    for (int mu = 0; mu < 4; mu++)
        if (c->geom.part[mu]) maxf = std::max(maxf, face_half_sites(c->geom, mu));
    if (!maxf) return LQCD_OK;
    LQCHK(force_buffers(c));
    FArgs k = make_fargs(c, kind, nullptr, nullptr, X, Y, 0.0, 1.0);
    hipLaunchKernelGGL(force_pack_kernel, dim3((2 * maxf + 127) / 128, 4), dim3(128), 0, c->stream, k);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

// lower face -> the -mu neighbour (which reads it as the ghost of its upper face)
int force_halo_exchange_rccl(lqcd_ctx_s* c, int kind) {
    const int nc = kind == LQCD_WILSON ? 12 : 3;
    ARGCHK(c->has_comm, "fermion force halo exchange: communicator not initialised (call lqcd_ctx_comm_init or lqcd_ctx_peer_init)");
    HIPCHK(hipEventRecord(c->ev_pack, c->stream));
    HIPCHK(hipStreamWaitEvent(c->comm_stream, c->ev_pack, 0));
    CommXfer x[4];
    int n = 0;
    for (int mu = 0; mu < 4; mu++) {
        if (!c->geom.part[mu]) continue;
        x[n++] = CommXfer{c->force_send[mu], c->force_recv[mu], force_halo_elems(c, mu, nc) * sizeof(double2), mu, 1};      // travels backward
    }
    LQCHK(comm_sendrecv(c, x, n, c->comm_stream, true));
    HIPCHK(hipEventRecord(c->ev_comm, c->comm_stream));
    HIPCHK(hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    return LQCD_OK;
}

int force_halo_exchange_local_all(lqcd_ctx_s** ctxs, int n, int kind) {
    const int nc = kind == LQCD_WILSON ? 12 : 3;
    for (int r = 0; r < n; r++) HIPCHK(hipStreamSynchronize(ctxs[r]->stream));
    for (int r = 0; r < n; r++) {
        lqcd_ctx_s* c = ctxs[r];
        for (int mu = 0; mu < 4; mu++) {
            if (!c->geom.part[mu]) continue;
            HIPCHK(hipMemcpy(ctxs[c->nbr_bwd[mu]]->force_recv[mu], c->force_send[mu], force_halo_elems(c, mu, nc) * sizeof(double2),
                             hipMemcpyDeviceToDevice));
        }
    }
    HIPCHK(hipDeviceSynchronize());
    return LQCD_OK;
}

int launch_fermion_force(lqcd_ctx_s* c, int kind, const lqcd_gauge_s* U, lqcd_gauge_s* out, lqcd_spinor_s* X, lqcd_spinor_s* Y, double km,
                         double r, double scale, int accumulate) {
    FArgs k = make_fargs(c, kind, U, out, X, Y, km, r);
    k.scale = scale;
    k.acc = accumulate;
    out->version++;
    const int nb = 2 * c->geom.nch;
    if (kind == LQCD_WILSON) {
        if (accumulate) hipLaunchKernelGGL(wilson_force_kernel<true>, dim3(nb), dim3(256), 0, c->stream, k);
        else hipLaunchKernelGGL(wilson_force_kernel<false>, dim3(nb), dim3(256), 0, c->stream, k);
    } else {
        if (accumulate) hipLaunchKernelGGL(staggered_force_kernel<true>, dim3(nb), dim3(256), 0, c->stream, k);
        else hipLaunchKernelGGL(staggered_force_kernel<false>, dim3(nb), dim3(256), 0, c->stream, k);
    }
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}

}  // namespace lqcd
