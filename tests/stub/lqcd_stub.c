/* lqcd_stub.c -- TEST INFRASTRUCTURE ONLY: a CPU stand-in for liblqcd_hip.so with the same C ABI, so that the N > 1 control path of
 * bench.py (rendezvous over torch.distributed/gloo, broadcast of the communicator id, PE-grid decomposition, barriers, max-over-ranks
 * reductions, the single JSON line of rank 0) can be EXECUTED by two real processes on a machine without GPUs
 * (tests/test_host_logic.py::test_bench_control_path_two_processes).  Nothing here computes physics: field handles are dummies, the
 * timing entry points return rank-dependent constants so the test can check that the max over ranks is what gets reported, and
 * lqcd_ctx_comm_init records the id each rank received in $LQCD_STUB_DIR/rank<r>.id so the test can check the broadcast.
 * Host-only entry points of the real library (lqcd_decompose, lqcd_version) are forwarded to it through dlopen.
 * Selected with LQCD_HIP_LIB=<this .so>; the product never loads it. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int gL[4], pe[4], rank, nranks, device; int p_variant, p_remap, p_nsub, p_ysplit, p_fused, p_recon, p_mode; } ctx_t;
static void* real_lib(void) {
    static void* h = NULL;
    if (!h) {
        const char* p = getenv("LQCD_STUB_REAL_LIB");
        h = dlopen(p ? p : "liblqcd_hip.so", RTLD_NOW | RTLD_LOCAL);
    }
    return h;
}
#define OK 0
static int rank_of_env_early(void) { const char* r = getenv("RANK"); return r ? atoi(r) : 0; }
const char* lqcd_last_error(void) { return "lqcd_stub: no error text"; }
int lqcd_version(void) { int (*f)(void) = real_lib() ? (int (*)(void))dlsym(real_lib(), "lqcd_version") : NULL; return f ? f() : 0; }
int lqcd_device_count(void) { return 8; }
int lqcd_decompose(const int* gL, const int* pe, int rank, int* L, int* org, int* nf, int* nb) {
    int (*f)(const int*, const int*, int, int*, int*, int*, int*) =
        real_lib() ? (int (*)(const int*, const int*, int, int*, int*, int*, int*))dlsym(real_lib(), "lqcd_decompose") : NULL;
    if (!f) { fprintf(stderr, "lqcd_stub: the real library (LQCD_STUB_REAL_LIB) is needed for lqcd_decompose\n"); return 1; }
    return f(gL, pe, rank, L, org, nf, nb);
}
int lqcd_ctx_create(void** out, int device, const int* gL, const int* pe, int rank) {
    ctx_t* c = (ctx_t*)calloc(1, sizeof(ctx_t));
    memcpy(c->gL, gL, 16); memcpy(c->pe, pe, 16);
    c->rank = rank; c->device = device; c->nranks = pe[0] * pe[1] * pe[2] * pe[3];
    c->p_variant = 1; c->p_remap = 2; c->p_nsub = 16; c->p_ysplit = 4; c->p_fused = 2; c->p_recon = 12; c->p_mode = 1;
    *out = c;
    return OK;
}
int lqcd_ctx_destroy(void* c) { free(c); return OK; }
int lqcd_ctx_sync(void* c) { (void)c; return OK; }
int lqcd_ctx_set_param(void* c, const char* k, int v) { (void)c; (void)k; (void)v; return OK; }
int lqcd_ctx_get_param(void* cc, const char* k, int* v) {
    ctx_t* c = (ctx_t*)cc;
    *v = !strcmp(k, "dslash_variant") ? c->p_variant : !strcmp(k, "xcd_remap") ? c->p_remap : !strcmp(k, "xcd_nsub") ? c->p_nsub :
         !strcmp(k, "xcd_ysplit") ? c->p_ysplit : !strcmp(k, "cg_fused") ? c->p_fused : !strcmp(k, "gauge_recon") ? c->p_recon :
         !strcmp(k, "recon_active") ? 1 : !strcmp(k, "halo_stream_mode") ? c->p_mode : 0;
    return OK;
}
int lqcd_comm_unique_id(unsigned char* id) { for (int i = 0; i < 256; i++) id[i] = (unsigned char)(37 * i + 11); return OK; }
static int g_backend[64];
int lqcd_ctx_comm_init(void* cc, const unsigned char* id, int nranks) {
    ctx_t* c = (ctx_t*)cc;
    if (nranks != c->nranks) return 1;
    g_backend[c->rank & 63] = 1;
    const char* d = getenv("LQCD_STUB_DIR");
    if (d) {
        char path[1024];
        snprintf(path, sizeof path, "%s/rank%d.id", d, c->rank);
        FILE* f = fopen(path, "wb");
        if (f) { fwrite(id, 1, 256, f); fprintf(f, "\npe=%d,%d,%d,%d nranks=%d device=%d\n", c->pe[0], c->pe[1], c->pe[2], c->pe[3], nranks, c->device); fclose(f); }
    }
    return OK;
}
/* the peer-mapped backend's bootstrap: the blob carries the rank, peer_init checks that the gathered blobs are in rank order and records them like comm_init does */
int lqcd_ctx_peer_export(void* cc, unsigned char* blob) {
    ctx_t* c = (ctx_t*)cc;
    if (getenv("LQCD_STUB_PEER_FAILS")) return 4;
    memset(blob, 0, 256);
    blob[0] = 'S'; blob[1] = (unsigned char)c->rank; blob[2] = (unsigned char)c->nranks;
    return OK;
}
int lqcd_ctx_peer_init(void* cc, const unsigned char* blobs, int nranks) {
    ctx_t* c = (ctx_t*)cc;
    if (nranks != c->nranks) return 1;
    for (int r = 0; r < nranks; r++) if (blobs[256 * r] != 'S' || blobs[256 * r + 1] != (unsigned char)r) return 4;
    const char* d = getenv("LQCD_STUB_DIR");
    if (d) {
        char path[1024];
        snprintf(path, sizeof path, "%s/rank%d.peer", d, c->rank);
        FILE* f = fopen(path, "wb");
        if (f) { fprintf(f, "pe=%d,%d,%d,%d nranks=%d device=%d\n", c->pe[0], c->pe[1], c->pe[2], c->pe[3], nranks, c->device); fclose(f); }
    }
    g_backend[c->rank & 63] = 2;
    return OK;
}
int lqcd_ctx_comm_backend(void* cc, int* b) { ctx_t* c = (ctx_t*)cc; *b = g_backend[c->rank & 63]; return OK; }
/* the halo self-check of the N > 1 line: |D b|^2 of the synthetic 32^3x64 problem; LQCD_STUB_BAD_NORM makes the peer "backend" return a wrong one (bench.py must then
 * fall back to RCCL on every rank) */
int lqcd_op_apply(void* op, void* out, void* in, int dagger) { (void)op; (void)out; (void)in; (void)dagger; return OK; }
int lqcd_dot(void* a, void* b, double* re, double* im) {
    (void)a; (void)b;
    const int bad = getenv("LQCD_STUB_BAD_NORM") && g_backend[rank_of_env_early() & 63] == 2;
    *re = bad ? 1.0 : 6.63734359571458697e+07; *im = 0.0;
    return OK;
}
static int handle(void** h) { *h = malloc(8); return OK; }
int lqcd_gauge_create(void* c, void** h) { (void)c; return handle(h); }
int lqcd_gauge_destroy(void* h) { free(h); return OK; }
int lqcd_gauge_hot_start(void* h, uint64_t s) { (void)h; (void)s; return OK; }
int lqcd_spinor_create(void* c, void** h, int kind, int subset) { (void)c; (void)kind; (void)subset; return handle(h); }
int lqcd_spinor_destroy(void* h) { free(h); return OK; }
int lqcd_spinor_gaussian(void* h, uint64_t s) { (void)h; (void)s; return OK; }
int lqcd_op_create(void* c, void** h, int kind, void* g, double km, double r, const int* bc) { (void)c; (void)kind; (void)g; (void)km; (void)r; (void)bc; return handle(h); }
int lqcd_op_destroy(void* h) { free(h); return OK; }
/* timings depend on the rank: rank r reports (r + 1) x the base value, so the job-wide figure must be the LAST rank's */
static int rank_of_env(void) { const char* r = getenv("RANK"); return r ? atoi(r) : 0; }
int lqcd_bench_dslash(void* op, void* o, void* i, int dag, int warm, int reps, double* ms) { (void)op; (void)o; (void)i; (void)dag; (void)warm; (void)reps; *ms = 0.05 * (rank_of_env() + 1); return OK; }
int lqcd_bench_dslash_median(void* op, void* o, void* i, int dag, int warm, int reps, double* med, double* mean) {
    (void)op; (void)o; (void)i; (void)dag; (void)warm; (void)reps; *med = 0.05 * (rank_of_env() + 1); *mean = *med; return OK; }
int lqcd_cg_session_begin(void* op, void* x, void* b) { (void)op; (void)x; (void)b; return OK; }
int lqcd_cg_session_iterate(void* op, int n) { (void)op; (void)n; return OK; }
int lqcd_cg_session_end(void* op) { (void)op; return OK; }
int lqcd_bench_halo_phases(void* op, void* o, void* i, int dag, int reps, double* ms) {
    (void)op; (void)o; (void)i; (void)dag; (void)reps; for (int k = 0; k < 6; k++) ms[k] = 0.01 * (k + 1) * (rank_of_env() + 1); return OK; }
int lqcd_bench_allreduce(void* c, int reps, double* us) { (void)c; (void)reps; *us = 10.0 * (rank_of_env() + 1); return OK; }
int64_t lqcd_index_lex(const int* L, int x, int y, int z, int t) { return x + (int64_t)L[0] * (y + (int64_t)L[1] * (z + (int64_t)L[2] * t)); }
