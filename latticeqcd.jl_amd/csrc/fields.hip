// fields.hip -- layout conversion between the reference's host arrays and the device checkerboard layout,
// seeded field generators, plaquette.  (Gaugefields.jl territory: Initialize_Gaugefields, substitute_U!,
// calculate_Plaquette -- call sites /root/reference/src/system/universe.jl:41-77, src/system/lqcd.jl:187-193.)
#include "lqcd_internal.h"

#include <atomic>

#include <cstring>

namespace lqcd {

// ------------------------------------------------------------------ gauge reorder
// host reference layout: a + 3*(b + 3*(site + V*mu));  disk layout: ((site*4 + mu)*3 + a)*3 + b
__device__ inline size_t host_gauge_index(int layout, size_t V, int mu, size_t site, int a, int b) {
    return layout == LQCD_LAYOUT_REFERENCE ? (size_t)a + 3 * ((size_t)b + 3 * (site + V * mu))
                                           : ((site * 4 + mu) * 3 + a) * 3 + b;
}

// host arrays may carry a wing of width w in every direction (the reference's Nwing: extents L + 2w, interior at offset w,
// Initialize_Gaugefields(NC, Nwing, ...) universe.jl:41-49): only the interior is read / written
__device__ inline size_t host_site(const Geom& g, const int c[4], int w) {
    return (size_t)(c[0] + w) + (size_t)(g.L[0] + 2 * w) * ((size_t)(c[1] + w) + (size_t)(g.L[1] + 2 * w) * ((size_t)(c[2] + w) + (size_t)(g.L[2] + 2 * w) * (size_t)(c[3] + w)));
}
__host__ __device__ inline size_t host_volume(const Geom& g, int w) {
    return (size_t)(g.L[0] + 2 * w) * (g.L[1] + 2 * w) * (g.L[2] + 2 * w) * (g.L[3] + 2 * w);
}

__global__ void gauge_reorder(Geom g, double2* dev, double2* host_img, int layout, int to_device, int w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.Vh) return;
    const int p = t / g.Vh, i = t % g.Vh;
    int c[4];
    cb_to_coords(g, p, i, c);
    const size_t V = host_volume(g, w);
    const size_t site = host_site(g, c, w);
    for (int mu = 0; mu < 4; mu++)
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                double2* d = dev + glink_off(g, p, mu, i) + (size_t)(a * 3 + b) * glink_stride(g);
                double2* h = host_img + host_gauge_index(layout, V, mu, site, a, b);
                if (to_device) *d = *h; else *h = *d;
            }
}

// ------------------------------------------------------------------ spinor reorder
// host: ic + 3*(site + V*is)  (Wilson, is = 0..3)  /  ic + 3*site (staggered)
__global__ void spinor_reorder(Geom g, double2* dev0, double2* dev1, double2* host_img, int nspin, int to_device, int w) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.Vh) return;
    const int p = t / g.Vh, i = t % g.Vh;
    double2* dev = p ? dev1 : dev0;
    if (!dev) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    const size_t V = host_volume(g, w);
    const size_t site = host_site(g, c, w);
    for (int s = 0; s < nspin; s++)
        for (int ic = 0; ic < 3; ic++) {
            double2* d = dev + sp_off(nspin * 3, i) + (size_t)(s * 3 + ic) * sp_stride(g);
            double2* h = host_img + ic + 3 * (site + V * s);
            if (to_device) *d = *h; else *h = *d;
        }
}

// ------------------------------------------------------------------ generators
__device__ inline void gauss_pair(uint64_t key, double& a, double& b) {
    const double u1 = u01(splitmix64(key)), u2 = u01(splitmix64(key ^ 0x5851F42D4C957F2Dull));
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586476925286766559 * u2;
    a = rad * cos(ang);
    b = rad * sin(ang);
}

__device__ inline uint64_t global_site(const Geom& g, const int c[4]) {
    const uint64_t x = c[0] + g.origin[0], y = c[1] + g.origin[1], z = c[2] + g.origin[2], t = c[3] + g.origin[3];
    return x + (uint64_t)g.gL[0] * (y + (uint64_t)g.gL[1] * (z + (uint64_t)g.gL[2] * t));
}

// hot start: Gaussian 3x3, rows 1-2 Gram-Schmidt, row 3 = conj(row1 x row2)  
__global__ void gauge_hot(Geom g, double2* dev, uint64_t seed) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.Vh * 4) return;
    const int mu = t & 3, s = t >> 2;
    const int p = s / g.Vh, i = s % g.Vh;
    int c[4];
    cb_to_coords(g, p, i, c);
    const uint64_t gs = global_site(g, c);
    double m[3][3][2];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) gauss_pair(rng_key(seed, gs * 4 + mu, a, b), m[a][b][0], m[a][b][1]);
    // normalise row 0
    double n0 = 0;
    for (int b = 0; b < 3; b++) n0 += m[0][b][0] * m[0][b][0] + m[0][b][1] * m[0][b][1];
    n0 = 1.0 / sqrt(n0);
    for (int b = 0; b < 3; b++) { m[0][b][0] *= n0; m[0][b][1] *= n0; }
    // row1 -= <row0,row1> row0
    double dr = 0, di = 0;
    for (int b = 0; b < 3; b++) {
        dr += m[0][b][0] * m[1][b][0] + m[0][b][1] * m[1][b][1];
        di += m[0][b][0] * m[1][b][1] - m[0][b][1] * m[1][b][0];
    }
    for (int b = 0; b < 3; b++) {
        m[1][b][0] -= dr * m[0][b][0] - di * m[0][b][1];
        m[1][b][1] -= dr * m[0][b][1] + di * m[0][b][0];
    }
    double n1 = 0;
    for (int b = 0; b < 3; b++) n1 += m[1][b][0] * m[1][b][0] + m[1][b][1] * m[1][b][1];
    n1 = 1.0 / sqrt(n1);
    for (int b = 0; b < 3; b++) { m[1][b][0] *= n1; m[1][b][1] *= n1; }
    // row2 = conj(row0 x row1)
    for (int b = 0; b < 3; b++) {
        const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
        const double xr = m[0][b1][0] * m[1][b2][0] - m[0][b1][1] * m[1][b2][1] - (m[0][b2][0] * m[1][b1][0] - m[0][b2][1] * m[1][b1][1]);
        const double xi = m[0][b1][0] * m[1][b2][1] + m[0][b1][1] * m[1][b2][0] - (m[0][b2][0] * m[1][b1][1] + m[0][b2][1] * m[1][b1][0]);
        m[2][b][0] = xr;
        m[2][b][1] = -xi;
    }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            dev[glink_off(g, p, mu, i) + (size_t)(a * 3 + b) * glink_stride(g)] = make_double2(m[a][b][0], m[a][b][1]);
}

__global__ void gauge_unit(Geom g, double2* dev) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.Vh * 4) return;
    const int mu = t & 3, s = t >> 2;
    const int p = s / g.Vh, i = s % g.Vh;
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            dev[glink_off(g, p, mu, i) + (size_t)(a * 3 + b) * glink_stride(g)] = make_double2(a == b ? 1.0 : 0.0, 0.0);
}

// mode 0: gaussian re,im ~ N(0,1); mode 1: Z4 noise (+-1, +-i)
__global__ void spinor_fill(Geom g, double2* dev0, double2* dev1, int ncomp, uint64_t seed, int mode) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * g.Vh) return;
    const int p = t / g.Vh, i = t % g.Vh;
    double2* dev = p ? dev1 : dev0;
    if (!dev) return;
    int c[4];
    cb_to_coords(g, p, i, c);
    const uint64_t gs = global_site(g, c);
    for (int k = 0; k < ncomp; k++) {
        const uint64_t key = rng_key(seed, gs, 77, k);
        double2 v;
        if (mode == 0) {
            gauss_pair(key, v.x, v.y);
        } else {
            const int z = (int)(splitmix64(key) >> 62);
            v = make_double2(z == 0 ? 1.0 : (z == 2 ? -1.0 : 0.0), z == 1 ? 1.0 : (z == 3 ? -1.0 : 0.0));
        }
        dev[sp_off(ncomp, i) + (size_t)k * sp_stride(g)] = v;
    }
}

// ------------------------------------------------------------------ plaquette
__device__ inline void load_link_at(cd (&u)[9], const double2* gauge, const Geom& g, int mu, const int c[4]) {
    const int p = (c[0] + c[1] + c[2] + c[3]) & 1;
    const int i = coords_to_cb(g, c);
    const double2* U = gauge + glink_off(g, p, mu, i);
    for (int k = 0; k < 9; k++) u[k] = ld(U + (size_t)k * glink_stride(g));
}
__device__ inline void mm(cd (&C)[9], const cd (&A)[9], const cd (&B)[9], bool adjB) {
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            cd t = mk(0, 0);
            for (int k = 0; k < 3; k++) {
                cd bb = adjB ? mk(B[b * 3 + k].re, -B[b * 3 + k].im) : B[k * 3 + b];
                cfma(t, A[a * 3 + k], bb);
            }
            C[a * 3 + b] = t;
        }
}

// ghost[mu]: links of the +mu neighbour's x_mu = 0 slice, layout [parity][nu][9][Fh(mu)] (only used when part[mu])
struct PlaqArgs {
    Geom g;
    const double2* gauge;
    const double2* ghost[4];
    double* partial;
};

__device__ inline void link_shifted(cd (&u)[9], const PlaqArgs& k, int nu, const int c[4], int mu) {
    // link U_nu at site c + mu_hat
    int d[4] = {c[0], c[1], c[2], c[3]};
    d[mu] += 1;
    if (d[mu] == k.g.L[mu]) {
        d[mu] = 0;
        if (k.g.part[mu]) {
            const int p = (d[0] + d[1] + d[2] + d[3]) & 1;
            const int Fh = face_half_sites(k.g, mu);
            const int f = coords_to_face(k.g, mu, d);
            const double2* U = k.ghost[mu] + ((size_t)(p * 4 + nu) * 9) * Fh + f;
            for (int j = 0; j < 9; j++) u[j] = ld(U + (size_t)j * Fh);
            return;
        }
    }
    load_link_at(u, k.gauge, k.g, nu, d);
}

__global__ __launch_bounds__(128) void plaquette_kernel(PlaqArgs k) {
    __shared__ double red[2];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double sum = 0;
    if (t < 2 * k.g.Vh) {
        const int p = t / k.g.Vh, i = t % k.g.Vh;
        int c[4];
        cb_to_coords(k.g, p, i, c);
        for (int mu = 0; mu < 4; mu++)
            for (int nu = mu + 1; nu < 4; nu++) {
                cd A[9], B[9], C[9], D[9], T1[9], T2[9], T3[9];
                load_link_at(A, k.gauge, k.g, mu, c);
                link_shifted(B, k, nu, c, mu);
                link_shifted(C, k, mu, c, nu);
                load_link_at(D, k.gauge, k.g, nu, c);
                mm(T1, A, B, false);
                mm(T2, T1, C, true);
                mm(T3, T2, D, true);
                sum += T3[0].re + T3[4].re + T3[8].re;
            }
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) k.partial[blockIdx.x] = red[0] + red[1];
}

// pack the x_mu = 0 slice of all links (both parities) for the -mu neighbour
__global__ void gauge_face_pack(Geom g, const double2* gauge, double2* dst, int mu) {
    const int Fh = face_half_sites(g, mu);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 2 * Fh) return;
    const int p = t / Fh, f = t % Fh;
    int c[4];
    face_to_coords(g, mu, 0, p, f, c);
    const int i = coords_to_cb(g, c);
    for (int nu = 0; nu < 4; nu++)
        for (int j = 0; j < 9; j++)
            dst[((size_t)(p * 4 + nu) * 9 + j) * Fh + f] = gauge[glink_off(g, p, nu, i) + (size_t)j * glink_stride(g)];
}

}  // namespace lqcd

using namespace lqcd;

// ------------------------------------------------------------------ C API: gauge
extern "C" int lqcd_gauge_create(lqcd_ctx_t ctx, lqcd_gauge_t* g) {
    ARGCHK(ctx && g, "lqcd_gauge_create: null argument");
    HIPCHK(hipSetDevice(ctx->device));
    LQCHK(lqcd::ctx_drain_parked(ctx));      // fields whose destroy call came from another thread (a finalizer): their storage goes before new storage is asked for
    lqcd_gauge_s* x = new lqcd_gauge_s;
    x->ctx = ctx;
    // versions are unique across handles (handle epoch in the upper 32 bits, writes counted in the lower): caches keyed on a
    // version (12-real copy, clover term, fp32 copies of the mixed-precision solver) can never match a recycled handle address
    static std::atomic<uint64_t> epoch{0};
    x->version = (++epoch << 32) | 1u;
    x->elems = gauge_elems(ctx->geom);
    x->data = nullptr;
    hipError_t e = hipMalloc((void**)&x->data, x->elems * sizeof(double2));
    if (e != hipSuccess && std::this_thread::get_id() == ctx->home_thread) {      // fields that finalizer threads parked still hold device memory: free them and try once more
        (void)hipGetLastError();
        (void)lqcd::ctx_drain_parked(ctx);
        e = hipMalloc((void**)&x->data, x->elems * sizeof(double2));
    }
    if (e != hipSuccess) { delete x; return hip_fail(e, "hipMalloc(gauge)", __FILE__, __LINE__); }
    e = hipMemsetAsync(x->data, 0, x->elems * sizeof(double2), ctx->stream);  // stride padding stays zero
    if (e != hipSuccess) { (void)hipFree(x->data); delete x; return hip_fail(e, "memset(gauge)", __FILE__, __LINE__); }
    *g = x;
    return LQCD_OK;
}

namespace lqcd {
std::mutex& view_mutex() {
    static std::mutex m;
    return m;
}
int gauge_destroy_now(lqcd_gauge_s* g) {
    // (no hipSetDevice: the context may already be gone -- finalizers run in any order -- and hipFree does not need it)
    if (ctx_is_live(g->ctx)) (void)links_flush_of(g);      // recorded link operations may name this field: they run before its storage goes
    (void)hipFree(g->data);
    (void)hipFree(g->data12);
    (void)hipFree(g->data12d);
    delete g;
    return LQCD_OK;
}
}   // namespace lqcd

extern "C" int lqcd_gauge_destroy(lqcd_gauge_t g) {
    if (!g) return LQCD_OK;
    // A destroy call from another thread than the context's own (a finalizer thread, while the context's thread may be inside a library call) must not run the
    // recorded link operations or touch their record: the field is parked and the context's thread frees it (capi.hip ctx_park_gauge; only while lazy_links is on)
    if (lqcd::ctx_park_gauge(g)) return LQCD_OK;
    return lqcd::gauge_destroy_now(g);
}

static int gauge_xfer(lqcd_gauge_t g, double* host, int layout, int to_device, int wing = 0) {
    ARGCHK(g && host, "gauge upload/download: null argument");
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(layout == LQCD_LAYOUT_REFERENCE || layout == LQCD_LAYOUT_DISK, "gauge upload/download: bad layout tag");
    ARGCHK(wing >= 0 && wing <= 4 && (wing == 0 || layout == LQCD_LAYOUT_REFERENCE), "gauge upload/download: wings exist in the reference layout only (width 0..4)");
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    double2* img = nullptr;
    const size_t bytes = (size_t)4 * 9 * host_volume(c->geom, wing) * sizeof(double2);  // host image: no stride padding, wings if any
    HIPCHK(hipMalloc((void**)&img, bytes));
    int st = LQCD_OK;
    const int nt = 2 * c->geom.Vh;
    if (to_device || wing) {      // a download into a winged array starts from the caller's image so that the wings survive
        if (to_device) g->version++;
        hipError_t e = hipMemcpyAsync(img, host, bytes, hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) st = hip_fail(e, "H2D gauge", __FILE__, __LINE__);
    }
    if (st == LQCD_OK) {
        hipLaunchKernelGGL(gauge_reorder, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->geom, g->data, img, layout, to_device, wing);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) st = hip_fail(e, "gauge_reorder", __FILE__, __LINE__);
    }
    if (st == LQCD_OK && !to_device) {
        hipError_t e = hipMemcpyAsync(host, img, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) st = hip_fail(e, "D2H gauge", __FILE__, __LINE__);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync gauge xfer", __FILE__, __LINE__);
    (void)hipFree(img);
    return st;
}

extern "C" int lqcd_gauge_upload(lqcd_gauge_t g, const double* host, int layout) {
    return gauge_xfer(g, const_cast<double*>(host), layout, 1);
}
extern "C" int lqcd_gauge_download(lqcd_gauge_t g, double* host, int layout) { return gauge_xfer(g, host, layout, 0); }
// the reference's arrays with Nwing > 0: (NC,NC,NX+2w,NY+2w,NZ+2w,NT+2w) per direction; only the interior is transferred
extern "C" int lqcd_gauge_upload_wing(lqcd_gauge_t g, const double* host, int nwing) {
    return gauge_xfer(g, const_cast<double*>(host), LQCD_LAYOUT_REFERENCE, 1, nwing);
}
extern "C" int lqcd_gauge_download_wing(lqcd_gauge_t g, double* host, int nwing) { return gauge_xfer(g, host, LQCD_LAYOUT_REFERENCE, 0, nwing); }

extern "C" int lqcd_gauge_unit(lqcd_gauge_t g) {
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(g, "lqcd_gauge_unit: null");
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int nt = 2 * c->geom.Vh * 4;
    g->version++;
    g->unitary_version = g->version;
    hipLaunchKernelGGL(gauge_unit, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->geom, g->data);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

extern "C" int lqcd_gauge_hot_start(lqcd_gauge_t g, uint64_t seed) {
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(g, "lqcd_gauge_hot_start: null");
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int nt = 2 * c->geom.Vh * 4;
    g->version++;
    g->unitary_version = g->version;      // rows 0, 1 by Gram-Schmidt, row 2 = conj(row 0 x row 1)
    hipLaunchKernelGGL(gauge_hot, dim3((nt + 127) / 128), dim3(128), 0, c->stream, c->geom, g->data, seed);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}

namespace lqcd {
// local (un-normalised) plaquette sum of one rank; ghost links must already be in place when partitioned
int plaquette_local_sum(lqcd_gauge_s* g, const double2* const ghost[4], double* sum) {
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    PlaqArgs k;
    k.g = c->geom;
    k.gauge = g->data;
    for (int mu = 0; mu < 4; mu++) k.ghost[mu] = ghost ? ghost[mu] : nullptr;
    k.partial = c->d_partial;
    const int nt = 2 * c->geom.Vh, nb = (nt + 127) / 128;
    hipLaunchKernelGGL(plaquette_kernel, dim3(nb), dim3(128), 0, c->stream, k);
    HIPCHK(hipGetLastError());
    std::vector<double> h(nb);
    HIPCHK(hipMemcpyAsync(h.data(), c->d_partial, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double s = 0;
    for (int i = 0; i < nb; i++) s += h[i];
    *sum = s;
    return LQCD_OK;
}

int gauge_pack_face(lqcd_gauge_s* g, int mu, double2* dst) {
    lqcd_ctx_s* c = g->ctx;
    const int nt = 2 * face_half_sites(c->geom, mu);
    hipLaunchKernelGGL(gauge_face_pack, dim3((nt + 127) / 128), dim3(128), 0, c->stream, c->geom, g->data, dst, mu);
    HIPCHK(hipGetLastError());
    return LQCD_OK;
}
}  // namespace lqcd

// ------------------------------------------------------------------ C API: spinors
extern "C" int lqcd_spinor_create(lqcd_ctx_t ctx, lqcd_spinor_t* s, int kind, int subset) {
    ARGCHK(ctx && s, "lqcd_spinor_create: null argument");
    ARGCHK(kind == LQCD_WILSON || kind == LQCD_STAGGERED, "lqcd_spinor_create: bad kind");
    ARGCHK(subset == LQCD_FULL || subset == LQCD_EVEN || subset == LQCD_ODD, "lqcd_spinor_create: bad subset");
    HIPCHK(hipSetDevice(ctx->device));
    lqcd_spinor_s* x = new lqcd_spinor_s;
    x->ctx = ctx;
    x->kind = kind;
    x->subset = subset;
    x->ncomp = kind == LQCD_WILSON ? 12 : 3;
    x->elems = (size_t)x->ncomp * ctx->geom.Vs * (subset == LQCD_FULL ? 2 : 1);
    x->data = nullptr;
    hipError_t e = hipMalloc((void**)&x->data, x->elems * sizeof(double2));
    if (e != hipSuccess && std::this_thread::get_id() == ctx->home_thread) {      // (as for gauge fields: parked storage is released and the allocation repeated once)
        (void)hipGetLastError();
        (void)lqcd::ctx_drain_parked(ctx);
        e = hipMalloc((void**)&x->data, x->elems * sizeof(double2));
    }
    if (e != hipSuccess) { delete x; return hip_fail(e, "hipMalloc(spinor)", __FILE__, __LINE__); }
    e = hipMemsetAsync(x->data, 0, x->elems * sizeof(double2), ctx->stream);
    if (e != hipSuccess) { (void)hipFree(x->data); delete x; return hip_fail(e, "memset(spinor)", __FILE__, __LINE__); }
    *s = x;
    return LQCD_OK;
}

extern "C" int lqcd_spinor_destroy(lqcd_spinor_t s) {
    if (!s) return LQCD_OK;
    // (no hipSetDevice: see lqcd_gauge_destroy)
    std::lock_guard<std::mutex> lk(lqcd::view_mutex());      // the view count of a five-dimensional field: destroy calls may come from finalizer threads
    if (s->view_of) {                      // a slice view: the last one of a destroyed parent takes the parent's storage with it
        lqcd_spinor_s* parent = s->view_of;
        delete s;
        if (--parent->nviews == 0 && parent->zombie) { (void)hipFree(parent->data); delete parent; }
        return LQCD_OK;
    }
    if (s->nviews > 0) { s->zombie = true; return LQCD_OK; }      // views are still out: the storage stays until the last of them is destroyed
    if (s->owner) (void)hipFree(s->data);
    delete s;
    return LQCD_OK;
}

extern "C" int lqcd_gauge_unitarity_deviation(lqcd_gauge_t g, double* maxdev) {
    LQCHK(lqcd::links_flush_of(g));      // recorded single-direction link operations run first (md.hip)
    ARGCHK(g && maxdev, "lqcd_gauge_unitarity_deviation: null argument");
    LQCHK(lqcd::gauge_ensure_recon12(g));      // measured by the pass that builds the 12-real copy (once per version of the field)
    *maxdev = g->recon_dev;
    return LQCD_OK;
}

namespace lqcd {
// device pointer of the parity block p of a spinor (nullptr if the spinor does not hold that parity)
// 12-real copy of the links: rows 0 and 1; *maxdev receives max |row2 - conj(row0 x row1)| (as the bit pattern of a
// non-negative double, which orders like an unsigned integer)
__global__ __launch_bounds__(256) void gauge_compress12(Geom g, const double2* __restrict__ src, double2* __restrict__ dst, unsigned long long* maxdev) {
    // lane = site (coalesced 1 KiB rows in both layouts), blockIdx.y = (parity, mu); one atomic per wave
    const int i = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y >> 2, mu = blockIdx.y & 3;
    double dev = 0.0;
    if (i < g.Vh) {
        const size_t so = glink_off(g, p, mu, i), d_o = glink12_off(g, p, mu, i);
        const int Gs = glink_stride(g);
        cd u[9];
#pragma unroll
        for (int e = 0; e < 9; e++) u[e] = ld(src + so + (size_t)e * Gs);
#pragma unroll
        for (int e = 0; e < 6; e++) st(dst + d_o + (size_t)e * 64, u[e]);
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
            const cd x = cmul(u[b1], u[3 + b2]) - cmul(u[b2], u[3 + b1]);      // (row0 x row1)_b
            const double d1 = fabs(u[6 + b].re - x.re), d2 = fabs(u[6 + b].im + x.im);
            dev = (d1 <= 1e300 && d2 <= 1e300) ? fmax(dev, fmax(d1, d2)) : 1e300;      // NaN / inf links are not unitary
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dev = fmax(dev, __shfl_down(dev, off, 64));
    if ((threadIdx.x & 63) == 0 && dev > 0.0) atomicMax(maxdev, (unsigned long long)__double_as_longlong(dev));
}

int gauge_ensure_recon12(lqcd_gauge_s* g) {
    if (g->version12 == g->version && g->data12) return LQCD_OK;
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!g->data12) HIPCHK(hipMalloc((void**)&g->data12, gauge12_elems(c->geom) * sizeof(double2)));
    unsigned long long* d_dev = (unsigned long long*)(c->d_scal + SCAL_DOUBLES - 9);
    HIPCHK(hipMemsetAsync(d_dev, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(gauge_compress12, dim3((c->geom.Vh + 255) / 256, 8), dim3(256), 0, c->stream, c->geom, g->data, g->data12, d_dev);
    HIPCHK(hipGetLastError());
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, d_dev, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double dev;
    memcpy(&dev, &bits, sizeof(dev));
    g->recon_ok = dev <= 1e-14;
    g->recon_dev = dev;
    if (g->recon_ok) g->unitary_version = g->version;
    g->version12 = g->version;
    return LQCD_OK;
}

// "12 + delta" copy: words 0..5 = rows 0, 1; word 6 = (delta_0, delta_1), word 7 = (delta_2, 0) as fp32 complex pairs, delta = row 2 - conj(row 0 x row 1)
// with the cross product formed by the SAME fma sequence the kernels use (stencil_common.h recon_row2); *maxdev: max |delta| (bit pattern, see above)
__global__ __launch_bounds__(256) void gauge_compress12d(Geom g, const double2* __restrict__ src, double2* __restrict__ dst, unsigned long long* maxdev) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, p = blockIdx.y >> 2, mu = blockIdx.y & 3;
    double dev = 0.0;
    if (i < g.Vh) {
        const size_t so = glink_off(g, p, mu, i);
        const size_t d_o = ((((size_t)p * g.nch + (size_t)(i >> 6)) * 4 + mu) * 8) * 64 + (i & 63);
        const int Gs = glink_stride(g);
        cd u[9];
#pragma unroll
        for (int e = 0; e < 9; e++) u[e] = ld(src + so + (size_t)e * Gs);
#pragma unroll
        for (int e = 0; e < 6; e++) st(dst + d_o + (size_t)e * 64, u[e]);
        float dl[6];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            const int b1 = (b + 1) % 3, b2 = (b + 2) % 3;
            const cd a = u[b1], bb = u[3 + b2], c = u[b2], d = u[3 + b1];
            double wr = a.re * bb.re;
            wr = fma(-a.im, bb.im, wr); wr = fma(-c.re, d.re, wr); wr = fma(c.im, d.im, wr);
            double wi = a.re * bb.im;
            wi = fma(a.im, bb.re, wi); wi = fma(-c.re, d.im, wi); wi = fma(-c.im, d.re, wi);
            const double dr = u[6 + b].re - wr, di = u[6 + b].im + wi;      // row 2 - conj(row 0 x row 1)
            dl[2 * b] = (float)dr; dl[2 * b + 1] = (float)di;
            const double d1 = fabs(dr), d2 = fabs(di);
            dev = (d1 <= 1e300 && d2 <= 1e300) ? fmax(dev, fmax(d1, d2)) : 1e300;
        }
        double2 w6, w7;
        w6.x = __hiloint2double(__float_as_int(dl[1]), __float_as_int(dl[0]));
        w6.y = __hiloint2double(__float_as_int(dl[3]), __float_as_int(dl[2]));
        w7.x = __hiloint2double(__float_as_int(dl[5]), __float_as_int(dl[4]));
        w7.y = 0.0;
        dst[d_o + 6 * 64] = w6;
        dst[d_o + 7 * 64] = w7;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dev = fmax(dev, __shfl_down(dev, off, 64));
    if ((threadIdx.x & 63) == 0 && dev > 0.0) atomicMax(maxdev, (unsigned long long)__double_as_longlong(dev));
}

int gauge_ensure_recon12d(lqcd_gauge_s* g) {
    if (g->version12d == g->version && g->data12d) return LQCD_OK;
    lqcd_ctx_s* c = g->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!g->data12d) HIPCHK(hipMalloc((void**)&g->data12d, (size_t)2 * c->geom.nch * 4 * 8 * 64 * sizeof(double2)));
    unsigned long long* d_dev = (unsigned long long*)(c->d_scal + SCAL_DOUBLES - 9);
    HIPCHK(hipMemsetAsync(d_dev, 0, sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(gauge_compress12d, dim3((c->geom.Vh + 255) / 256, 8), dim3(256), 0, c->stream, c->geom, g->data, g->data12d, d_dev);
    HIPCHK(hipGetLastError());
    unsigned long long bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, d_dev, sizeof(bits), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    double dev;
    memcpy(&dev, &bits, sizeof(dev));
    g->delta_ok = dev <= 1e-9;
    g->version12d = g->version;
    return LQCD_OK;
}

double2* spinor_block(lqcd_spinor_s* s, int p) {
    const size_t blk = (size_t)s->ncomp * s->ctx->geom.Vs;
    if (s->subset == LQCD_FULL) return s->data + p * blk;
    if (s->subset == LQCD_EVEN) return p == 0 ? s->data : nullptr;
    return p == 1 ? s->data : nullptr;
}
}  // namespace lqcd

static int spinor_xfer(lqcd_spinor_t s, double* host, int to_device, int wing = 0) {
    ARGCHK(s && host, "spinor upload/download: null argument");
    ARGCHK(s->ls == 1, "spinor upload/download: a five-dimensional field moves slice by slice (lqcd_spinor_slice)");
    ARGCHK(wing >= 0 && wing <= 4, "spinor upload/download: wing width 0..4");
    lqcd_ctx_s* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    const size_t full = (size_t)s->ncomp * host_volume(c->geom, wing);
    const size_t bytes = full * sizeof(double2);
    double2* img = nullptr;
    HIPCHK(hipMalloc((void**)&img, bytes));
    int st = LQCD_OK;
    hipError_t e = hipSuccess;
    // download of a half field: start from the caller's array so the other parity is preserved
    if (to_device || s->subset != LQCD_FULL || wing) e = hipMemcpyAsync(img, host, bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) st = hip_fail(e, "H2D spinor", __FILE__, __LINE__);
    if (st == LQCD_OK) {
        const int nt = 2 * c->geom.Vh;
        hipLaunchKernelGGL(spinor_reorder, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->geom, spinor_block(s, 0),
                           spinor_block(s, 1), img, s->ncomp / 3, to_device, wing);
        e = hipGetLastError();
        if (e != hipSuccess) st = hip_fail(e, "spinor_reorder", __FILE__, __LINE__);
    }
    if (st == LQCD_OK && !to_device) {
        e = hipMemcpyAsync(host, img, bytes, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) st = hip_fail(e, "D2H spinor", __FILE__, __LINE__);
    }
    e = hipStreamSynchronize(c->stream);
    if (st == LQCD_OK && e != hipSuccess) st = hip_fail(e, "sync spinor xfer", __FILE__, __LINE__);
    (void)hipFree(img);
    return st;
}

extern "C" int lqcd_spinor_upload(lqcd_spinor_t s, const double* host) { return spinor_xfer(s, const_cast<double*>(host), 1); }
extern "C" int lqcd_spinor_download(lqcd_spinor_t s, double* host) { return spinor_xfer(s, host, 0); }
// fields created without nowing = true carry a wing (the staggered fields of universe.jl:107): (NC,NX+2w,...,NT+2w,NG)
extern "C" int lqcd_spinor_upload_wing(lqcd_spinor_t s, const double* host, int nwing) { return spinor_xfer(s, const_cast<double*>(host), 1, nwing); }
extern "C" int lqcd_spinor_download_wing(lqcd_spinor_t s, double* host, int nwing) { return spinor_xfer(s, host, 0, nwing); }

extern "C" int lqcd_spinor_zero(lqcd_spinor_t s) {
    ARGCHK(s, "lqcd_spinor_zero: null");
    HIPCHK(hipSetDevice(s->ctx->device));
    HIPCHK(hipMemsetAsync(s->data, 0, s->elems * sizeof(double2), s->ctx->stream));
    HIPCHK(hipStreamSynchronize(s->ctx->stream));
    return LQCD_OK;
}

extern "C" int lqcd_spinor_copy(lqcd_spinor_t dst, lqcd_spinor_t src) {
    ARGCHK(dst && src, "lqcd_spinor_copy: null");
    ARGCHK(dst->ctx == src->ctx && dst->kind == src->kind && dst->subset == src->subset && dst->elems == src->elems, "lqcd_spinor_copy: shape mismatch");
    HIPCHK(hipSetDevice(dst->ctx->device));
    HIPCHK(hipMemcpyAsync(dst->data, src->data, dst->elems * sizeof(double2), hipMemcpyDeviceToDevice, dst->ctx->stream));
    HIPCHK(hipStreamSynchronize(dst->ctx->stream));
    return LQCD_OK;
}

static int spinor_fill_mode(lqcd_spinor_t s, uint64_t seed, int mode) {
    ARGCHK(s, "spinor fill: null");
    lqcd_ctx_s* c = s->ctx;
    HIPCHK(hipSetDevice(c->device));
    const int nt = 2 * c->geom.Vh;
    if (s->ls == 1) {
        // a half-lattice field is ONE parity block: the absent parity is a null pointer the kernel skips (spinor_block's rule), so an EVEN / ODD
        // field receives exactly the numbers the same parity of a FULL fill with this seed would
        hipLaunchKernelGGL(spinor_fill, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->geom, spinor_block(s, 0), spinor_block(s, 1), s->ncomp, seed, mode);
        HIPCHK(hipGetLastError());
    } else {
        // a five-dimensional field (always FULL): slice i5 is the four-dimensional fill with a seed hashed from (seed, i5) -- one stream per slice, keyed by
        // the global site as ever, and no collision with a field filled with a neighbouring seed
        ARGCHK(s->subset == LQCD_FULL, "spinor fill: a five-dimensional field is a FULL field");
        const size_t slice = s->elems / s->ls;
        for (int i5 = 0; i5 < s->ls; i5++) {
            double2* base = s->data + (size_t)i5 * slice;
            const uint64_t seed5 = splitmix64(seed ^ splitmix64(0x5D5D5D5D00000000ull + (uint64_t)i5));
            hipLaunchKernelGGL(spinor_fill, dim3((nt + 255) / 256), dim3(256), 0, c->stream, c->geom, base, base + slice / 2, s->ncomp, seed5, mode);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    return LQCD_OK;
}
extern "C" int lqcd_spinor_gaussian(lqcd_spinor_t s, uint64_t seed) { return spinor_fill_mode(s, seed, 0); }
extern "C" int lqcd_spinor_z4(lqcd_spinor_t s, uint64_t seed) { return spinor_fill_mode(s, seed, 1); }

extern "C" int lqcd_spinor_point_source(lqcd_spinor_t s, const int gx[4], int ic, int is) {
    ARGCHK(s && gx, "lqcd_spinor_point_source: null");
    lqcd_ctx_s* c = s->ctx;
    const int nspin = s->ncomp / 3;
    ARGCHK(ic >= 0 && ic < 3 && is >= 0 && is < nspin, "lqcd_spinor_point_source: bad colour/spin index");
    for (int mu = 0; mu < 4; mu++) ARGCHK(gx[mu] >= 0 && gx[mu] < c->gL[mu], "lqcd_spinor_point_source: site out of range");
    LQCHK(lqcd_spinor_zero(s));
    int lc[4];
    for (int mu = 0; mu < 4; mu++) {
        lc[mu] = gx[mu] - c->geom.origin[mu];
        if (lc[mu] < 0 || lc[mu] >= c->geom.L[mu]) return LQCD_OK;  // source lives on another rank
    }
    const int p = (lc[0] + lc[1] + lc[2] + lc[3]) & 1;
    double2* blk = spinor_block(s, p);
    if (!blk) return LQCD_OK;
    const double2 one = make_double2(1.0, 0.0);
    HIPCHK(hipMemcpy(blk + sp_off(s->ncomp, coords_to_cb(c->geom, lc)) + (size_t)(is * 3 + ic) * sp_stride(c->geom), &one, sizeof(one), hipMemcpyHostToDevice));
    return LQCD_OK;
}

extern "C" int lqcd_spinor_extract(lqcd_spinor_t half, lqcd_spinor_t full) {
    ARGCHK(half && full && half->ctx == full->ctx && half->kind == full->kind, "lqcd_spinor_extract: mismatch");
    ARGCHK(full->subset == LQCD_FULL && half->subset != LQCD_FULL, "lqcd_spinor_extract: need (half, full)");
    const int p = half->subset == LQCD_EVEN ? 0 : 1;
    HIPCHK(hipSetDevice(full->ctx->device));
    HIPCHK(hipMemcpyAsync(half->data, spinor_block(full, p), half->elems * sizeof(double2), hipMemcpyDeviceToDevice, full->ctx->stream));
    HIPCHK(hipStreamSynchronize(full->ctx->stream));
    return LQCD_OK;
}

extern "C" int lqcd_spinor_insert(lqcd_spinor_t full, lqcd_spinor_t half) {
    ARGCHK(half && full && half->ctx == full->ctx && half->kind == full->kind, "lqcd_spinor_insert: mismatch");
    ARGCHK(full->subset == LQCD_FULL && half->subset != LQCD_FULL, "lqcd_spinor_insert: need (full, half)");
    const int p = half->subset == LQCD_EVEN ? 0 : 1;
    HIPCHK(hipSetDevice(full->ctx->device));
    HIPCHK(hipMemcpyAsync(spinor_block(full, p), half->data, half->elems * sizeof(double2), hipMemcpyDeviceToDevice, full->ctx->stream));
    HIPCHK(hipStreamSynchronize(full->ctx->stream));
    return LQCD_OK;
}
