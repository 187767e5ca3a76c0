// ipc_probe.hip -- can two PROCESSES on ONE MI355X share device memory (hipIpc) and order kernels through flag words?
// Round-6 groundwork for the peer-mapped halo backend.  Build: hipcc --offload-arch=gfx950 -O2 ipc_probe.hip -o ipc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <sys/wait.h>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s failed: %s (line %d)\n", g_rank, #x, hipGetErrorString(e_), __LINE__); exit(3); } } while (0)
static int g_rank = -1;

struct Blob { hipIpcMemHandle_t h; size_t bytes; int pid; };

// 1 wave: lane 0 raises the peer's flag to `v` (after a system fence) and waits until its own flag reaches `v`; gives up after `limit` ticks of the 100 MHz clock
__global__ void sync_kernel(unsigned* peer_flag, unsigned* my_flag, unsigned v, unsigned long long limit, unsigned* status) {
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(peer_flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(my_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < v) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > limit) { *status = v; break; }
        }
    }
}
__global__ void fill_kernel(double* dst, size_t n, double v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v + (double)i;
}
__global__ void check_kernel(const double* src, size_t n, double v, unsigned* bad) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (src[i] != v + (double)i) atomicAdd(bad, 1u);
}
__global__ void busy_kernel(double* x, size_t n, int reps) {      // fills the machine: every CU busy for a while
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        double a = x[i];
        for (int r = 0; r < reps; r++) a = fma(a, 1.0000001, 1e-9);
        x[i] = a;
    }
}

static void xchg(int wfd, int rfd, const void* mine, void* theirs, size_t n) {
    if (write(wfd, mine, n) != (ssize_t)n) { perror("write"); exit(4); }
    size_t got = 0;
    while (got < n) { ssize_t r = read(rfd, (char*)theirs + got, n - got); if (r <= 0) { perror("read"); exit(4); } got += r; }
}

static int run(int rank, int wfd, int rfd, int finegrained) {
    g_rank = rank;
    CK(hipSetDevice(0));
    const size_t nd = (size_t)3 << 17;      // 3 MB of doubles = one face message
    const size_t bytes = 4096 + 2 * nd * sizeof(double);
    char* win = nullptr;
    if (finegrained) CK(hipExtMallocWithFlags((void**)&win, bytes, hipDeviceMallocFinegrained));
    else CK(hipMalloc((void**)&win, bytes));
    CK(hipMemset(win, 0, bytes));
    CK(hipDeviceSynchronize());
    Blob mine, theirs;
    memset(&mine, 0, sizeof mine);
    CK(hipIpcGetMemHandle(&mine.h, win));
    mine.bytes = bytes; mine.pid = getpid();
    xchg(wfd, rfd, &mine, &theirs, sizeof mine);
    char* peer = nullptr;
    CK(hipIpcOpenMemHandle((void**)&peer, theirs.h, hipIpcMemLazyEnablePeerAccess));
    printf("[%d] finegrained=%d window %p, peer (pid %d) mapped at %p\n", rank, finegrained, win, theirs.pid, peer); fflush(stdout);
    unsigned* my_flag = (unsigned*)win; unsigned* peer_flag = (unsigned*)peer;
    unsigned* status = nullptr;
    CK(hipHostMalloc((void**)&status, 64, hipHostMallocDefault));
    status[0] = 0; status[1] = 0;
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const unsigned long long limit = 100000000ull * 5;      // 5 s
    unsigned v = 0;
    // (1) correctness: I fill the peer's buffer[v & 1], sync, check mine
    for (int it = 0; it < 20; it++) {
        v++;
        double* dst = (double*)(peer + 4096) + (v & 1) * nd;
        const double* src = (const double*)(win + 4096) + (v & 1) * nd;
        fill_kernel<<<256, 256, 0, s>>>(dst, nd, 1000.0 * v + (1 - rank));     // the value the RECEIVER (rank 1 - rank) expects
        sync_kernel<<<1, 64, 0, s>>>(peer_flag, my_flag, v, limit, status);
        check_kernel<<<256, 256, 0, s>>>(src, nd, 1000.0 * v + rank, status + 1);
    }
    CK(hipStreamSynchronize(s));
    printf("[%d] correctness: timeouts=%u bad=%u\n", rank, status[0], status[1]); fflush(stdout);
    if (status[0] || status[1]) return 5;
    // (2) latency of an exchange step: back-to-back sync kernels
    const int N = 2000;
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < N; it++) { v++; sync_kernel<<<1, 64, 0, s>>>(peer_flag, my_flag, v, limit, status); }
    CK(hipStreamSynchronize(s));
    auto t1 = std::chrono::steady_clock::now();
    printf("[%d] sync kernel alone: %.2f us per step (timeouts %u)\n", rank, std::chrono::duration<double, std::micro>(t1 - t0).count() / N, status[0]); fflush(stdout);
    // (3) pack (3 MB into the peer) + sync + consumer, per step
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < N; it++) {
        v++;
        double* dst = (double*)(peer + 4096) + (v & 1) * nd;
        const double* src = (const double*)(win + 4096) + (v & 1) * nd;
        fill_kernel<<<256, 256, 0, s>>>(dst, nd, 1000.0 * v + (1 - rank));
        sync_kernel<<<1, 64, 0, s>>>(peer_flag, my_flag, v, limit, status);
        check_kernel<<<256, 256, 0, s>>>(src, nd, 1000.0 * v + rank, status + 1);
    }
    CK(hipStreamSynchronize(s));
    t1 = std::chrono::steady_clock::now();
    printf("[%d] fill 3 MB -> peer, sync, check 3 MB: %.2f us per step (timeouts %u bad %u)\n", rank, std::chrono::duration<double, std::micro>(t1 - t0).count() / N, status[0], status[1]); fflush(stdout);
    // the same three kernels without the peer (local buffer, no sync): what the kernels themselves cost
    {
        double* loc = nullptr;
        CK(hipMalloc((void**)&loc, nd * sizeof(double)));
        t0 = std::chrono::steady_clock::now();
        for (int it = 0; it < N; it++) {
            fill_kernel<<<256, 256, 0, s>>>(loc, nd, 1.0);
            check_kernel<<<256, 256, 0, s>>>(loc, nd, 1.0, status + 1);
        }
        CK(hipStreamSynchronize(s));
        t1 = std::chrono::steady_clock::now();
        printf("[%d] fill + check on a local hipMalloc buffer, no sync: %.2f us per step\n", rank, std::chrono::duration<double, std::micro>(t1 - t0).count() / N); fflush(stdout);
        CK(hipFree(loc));
    }
    // (4) no deadlock when both processes fill the machine in front of the sync
    {
        double* big = nullptr;
        const size_t nb = (size_t)1 << 24;
        CK(hipMalloc((void**)&big, nb * sizeof(double)));
        CK(hipMemset(big, 0, nb * sizeof(double)));
        for (int it = 0; it < 20; it++) {
            v++;
            busy_kernel<<<4096, 256, 0, s>>>(big, nb, 200 + 400 * rank);
            sync_kernel<<<1, 64, 0, s>>>(peer_flag, my_flag, v, limit, status);
        }
        CK(hipStreamSynchronize(s));
        printf("[%d] busy + sync x20: timeouts=%u\n", rank, status[0]); fflush(stdout);
        CK(hipFree(big));
    }
    // (5) interprocess events, for the record
    {
        hipEvent_t ev;
        hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventInterprocess);
        hipIpcEventHandle_t eh, peh;
        memset(&eh, 0, sizeof eh);
        if (e == hipSuccess) e = hipIpcGetEventHandle(&eh, ev);
        int ok = e == hipSuccess, pok = 0;
        xchg(wfd, rfd, &ok, &pok, sizeof ok);
        xchg(wfd, rfd, &eh, &peh, sizeof eh);
        printf("[%d] interprocess event: create/get handle %s\n", rank, ok ? "ok" : hipGetErrorString(e)); fflush(stdout);
        if (ok && pok) {
            hipEvent_t pev;
            e = hipIpcOpenEventHandle(&pev, peh);
            printf("[%d] hipIpcOpenEventHandle: %s\n", rank, hipGetErrorString(e)); fflush(stdout);
            if (e == hipSuccess) {
                const int M = 200;
                t0 = std::chrono::steady_clock::now();
                for (int it = 0; it < M; it++) {
                    CK(hipEventRecord(ev, s));
                    int one = 1, two = 0;
                    xchg(wfd, rfd, &one, &two, sizeof one);      // host rendezvous: the peer has recorded
                    CK(hipStreamWaitEvent(s, pev, 0));
                    fill_kernel<<<1, 64, 0, s>>>((double*)(win + 4096), 64, 1.0);
                }
                CK(hipStreamSynchronize(s));
                t1 = std::chrono::steady_clock::now();
                printf("[%d] record + host rendezvous + wait on the peer's event: %.2f us per step\n", rank, std::chrono::duration<double, std::micro>(t1 - t0).count() / M); fflush(stdout);
            }
        }
        (void)hipGetLastError();
    }
    int done = 1, pd = 0;
    xchg(wfd, rfd, &done, &pd, sizeof done);
    CK(hipIpcCloseMemHandle(peer));
    CK(hipFree(win));
    return 0;
}

int main(int argc, char** argv) {
    const int finegrained = argc > 1 ? atoi(argv[1]) : 1;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) { perror("pipe"); return 2; }
    const pid_t pid = fork();      // BEFORE any HIP call
    if (pid == 0) { close(p2c[1]); close(c2p[0]); return run(1, c2p[1], p2c[0], finegrained); }
    close(p2c[0]); close(c2p[1]);
    const int rc = run(0, p2c[1], c2p[0], finegrained);
    int st = 0;
    waitpid(pid, &st, 0);
    printf("parent rc %d, child rc %d\n", rc, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return rc || !WIFEXITED(st) || WEXITSTATUS(st);
}
