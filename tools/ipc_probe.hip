// ipc_probe.hip -- can N processes on ONE device exchange data inside kernels through hipIpc-mapped buffers?
// (The all-pairs gradient exchange of SURVEY.md section 8e needs peer-mapped arenas and cross-process flags; a
// 1-GPU box can exercise both functionally: every "peer" is another process on the same device.)
//
//   * interior pointers: does hipIpcGetMemHandle / hipIpcOpenMemHandle of (base + offset) return base or base + offset?
//   * do kernels of different processes run CONCURRENTLY (a kernel of process A spins on a flag that a kernel of
//     process B writes)?  flag round trip time
//   * the exchange itself: signal "my data is final" -> wait for every peer -> owner sums its slice over the ranks
//     in rank order (remote reads) -> pushes the result into every peer's buffer (remote writes) -> last workgroup
//     signals "done" and waits for the peers' -- one launch; every word checked, buffers reused every round
//
// build: hipcc --offload-arch=gfx950 -O3 -o ipc_probe tools/ipc_probe.hip ; run: ./ipc_probe [nproc=2] [rounds=50]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("[%d] %s: %s\n", g_rank, #x, hipGetErrorString(e_)); fflush(stdout); _exit(2); } } while (0)
static int g_rank = 0;

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kMaxRanks = 8;

struct Shared {                       // host shared memory (mmap before fork)
    hipIpcMemHandle_t h[kMaxRanks][3];
    long long off[kMaxRanks][3];
    volatile int arrive[16];
};
static void host_barrier(Shared* s, int slot, int n) {
    __sync_fetch_and_add(&s->arrive[slot], 1);
    while (s->arrive[slot] < n) usleep(50);
}

struct Peers {
    float* grads[kMaxRanks];          // each rank's "gradient" buffer (mine = local pointer)
    float* params[kMaxRanks];
    unsigned* flags[kMaxRanks];       // [0..8) ready[src], [8..16) done[src], [16] ticket, [17] error
};

__device__ inline unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline v4f load_sys(const float* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// four system-scope loads `step` bytes apart, issued together, ONE wait (an asm load's result must not be touched
// before its wait, and hipcc tracks nothing inside asm statements: issue and wait live in one statement)
__device__ inline void load_sys4(const float* p, long long step, v4f (&v)[4]) {
    const char* b = reinterpret_cast<const char*>(p);
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc0 sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc0 sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc0 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(b), "v"(b + step), "v"(b + 2 * step), "v"(b + 3 * step)
        : "memory");
}
__device__ inline void store_sys(float* p, const v4f& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// bounded wait: flag >= epoch, or give up after ~2 s and raise the error word
__device__ inline bool wait_ge(const unsigned* flag, unsigned epoch, unsigned* err) {
    const long long t0 = wall_clock64();
    while ((int)(ld_sys(flag) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000ll) { atomicAdd(err, 1u); return false; }
    }
    return true;
}

// flag ping: every rank writes `epoch` into slot `me` of every peer's flag block, then waits for all peers' writes
__global__ void ping_k(Peers P, int me, int n, unsigned epoch, unsigned long long* t) {
    const int q = threadIdx.x;
    const long long t0 = wall_clock64();
    if (q < n && q != me) st_sys(P.flags[q] + me, epoch);
    if (q < n && q != me) wait_ge(P.flags[me] + q, epoch, P.flags[me] + 17);
    if (q == 0) { t[0] += wall_clock64() - t0; }
}

__global__ void fill_k(float* g, long long n4, float val) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll)
        reinterpret_cast<v4f*>(g)[i] = v4f{val, val + 1.f, val + 2.f, val + 3.f};
}

// one launch: reduce-scatter (pull) + all-gather (push) of `n4` float4 over n ranks
__global__ void __launch_bounds__(256) exchange_k(Peers P, int me, int n, long long n4, unsigned epoch) {
    unsigned* mine = P.flags[me];
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < n && tid != me) {      // my gradients are final (the producing kernels precede this launch)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        st_sys(P.flags[tid] + me, epoch);
    }
    if (tid < n && tid != me) wait_ge(mine + tid, epoch, mine + 17);
    __syncthreads();
    const long long S = (n4 + n - 1) / n, lo = me * S, hi = lo + S < n4 ? lo + S : n4;
    const long long stride = gridDim.x * 256ll;
    long long i = lo + blockIdx.x * 256ll + tid;
    for (; i + 3 * stride < hi; i += 4 * stride) {       // four float4 per thread in flight per peer
        v4f s[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) s[u] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < n; ++q) {                    // rank order: every element is summed the same way everywhere
            v4f g[4];
            if (q == me) {
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = reinterpret_cast<const v4f*>(P.grads[me])[i + u * stride];
            } else {
                load_sys4(P.grads[q] + 4 * i, 16 * stride, g);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += g[u];
        }
        for (int q = 0; q < n; ++q)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q == me) reinterpret_cast<v4f*>(P.params[me])[i + u * stride] = s[u];
                else store_sys(P.params[q] + 4 * (i + u * stride), s[u]);
            }
    }
    for (; i < hi; i += stride) {
        v4f s = v4f{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < n; ++q)
            s += q == me ? reinterpret_cast<const v4f*>(P.grads[me])[i] : load_sys(P.grads[q] + 4 * i);
        for (int q = 0; q < n; ++q) {
            if (q == me) reinterpret_cast<v4f*>(P.params[me])[i] = s;
            else store_sys(P.params[q] + 4 * i, s);
        }
    }
    // last workgroup out: everything this rank pushed has been issued by all workgroups -> fence, tell the peers,
    // wait until every peer has told me (my buffers are then complete, and my gradients free for re-use)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        last = atomicAdd(mine + 16, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (tid == 0) mine[16] = 0;
    if (tid < n && tid != me) {
        st_sys(P.flags[tid] + 8 + me, epoch);
        wait_ge(mine + 8 + tid, epoch, mine + 17);
    }
}

__global__ void check4_k(const float* p, long long n4, v4f want, unsigned* bad) {
    unsigned b = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        const v4f v = reinterpret_cast<const v4f*>(p)[i];
        b += (v[0] != want[0]) + (v[1] != want[1]) + (v[2] != want[2]) + (v[3] != want[3]);
    }
    if (b) atomicAdd(bad, b);
}

static int run(int rank, int n, int rounds, Shared* sh) {
    g_rank = rank;
    CK(hipSetDevice(0));
    const long long n4 = 1ll << 20;                       // 16 MB per buffer
    char* block;                                          // grads and params live INSIDE one allocation (interior pointers)
    CK(hipMalloc(&block, 2 * n4 * 16 + (1 << 20)));
    float* grads = (float*)(block + (1 << 20));
    float* params = grads + 4 * n4;
    unsigned* flags;
    CK(hipExtMallocWithFlags((void**)&flags, 4096, hipDeviceMallocUncached));
    CK(hipMemset(flags, 0, 4096));
    CK(hipMemset(block, 0, 2 * n4 * 16 + (1 << 20)));
    void* ptrs[3] = {grads, params, flags};
    for (int k = 0; k < 3; ++k) {
        void* base = nullptr; size_t sz = 0;
        CK(hipMemGetAddressRange((hipDeviceptr_t*)&base, &sz, ptrs[k]));
        CK(hipIpcGetMemHandle(&sh->h[rank][k], base));
        sh->off[rank][k] = (char*)ptrs[k] - (char*)base;
        if (rank == 0) printf("buffer %d: allocation %zu bytes, pointer at +%lld\n", k, sz, sh->off[rank][k]);
    }
    if (rank == 0) {                                      // what does a handle of an interior pointer open to?
        hipIpcMemHandle_t hi;
        hipError_t e = hipIpcGetMemHandle(&hi, grads);
        printf("hipIpcGetMemHandle(interior pointer): %s; handle %s the base allocation's\n", hipGetErrorString(e),
               memcmp(&hi, &sh->h[0][0], sizeof(hi)) == 0 ? "EQUALS" : "differs from");
    }
    host_barrier(sh, 0, n);
    Peers P;
    memset(&P, 0, sizeof(P));
    for (int q = 0; q < n; ++q) {
        void* m[3];
        for (int k = 0; k < 3; ++k) {
            if (q == rank) { m[k] = ptrs[k]; continue; }
            void* base = nullptr;
            CK(hipIpcOpenMemHandle(&base, sh->h[q][k], hipIpcMemLazyEnablePeerAccess));
            m[k] = (char*)base + sh->off[q][k];
        }
        P.grads[q] = (float*)m[0]; P.params[q] = (float*)m[1]; P.flags[q] = (unsigned*)m[2];
    }
    host_barrier(sh, 1, n);
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned long long* t; CK(hipMalloc(&t, 8)); CK(hipMemset(t, 0, 8));
    unsigned* bad; CK(hipMalloc(&bad, 4)); CK(hipMemset(bad, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // 1. flag ping between kernels of different processes
    unsigned epoch = 0;
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(ping_k, dim3(1), dim3(64), 0, st, P, rank, n, ++epoch, t);
    CK(hipStreamSynchronize(st));
    unsigned long long ht; CK(hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost));
    unsigned herr; CK(hipMemcpy(&herr, flags + 17, 4, hipMemcpyDeviceToHost));
    printf("[%d] flag ping: %.2f us per round inside the kernel, time-outs %u\n", rank, ht / 100.0 / 200, herr);
    host_barrier(sh, 2, n);
    CK(hipMemset(flags, 0, 4096));
    host_barrier(sh, 3, n);

    // 2. the exchange, `rounds` times over the same buffers
    float ms_total = 0;
    epoch = 0;
    for (int r = 1; r <= rounds; ++r) {
        const float val = (float)(r * 16 + rank);        // rank q's gradient = (val_q, val_q + 1, ...)
        hipLaunchKernelGGL(fill_k, dim3(256), dim3(256), 0, st, grads, n4, val);
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(exchange_k, dim3(256), dim3(256), 0, st, P, rank, n, n4, ++epoch);
        CK(hipEventRecord(e1, st));
        v4f want = v4f{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < n; ++q) { const float vq = (float)(r * 16 + q); want += v4f{vq, vq + 1.f, vq + 2.f, vq + 3.f}; }
        hipLaunchKernelGGL(check4_k, dim3(256), dim3(256), 0, st, params, n4, want, bad);
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 5) ms_total += ms;
    }
    unsigned hbad; CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, flags + 17, 4, hipMemcpyDeviceToHost));
    printf("[%d] exchange of 16 MB over %d ranks: %.1f us per launch, mismatching words %u, time-outs %u\n", rank, n,
           ms_total * 1000 / (rounds - 5), hbad, herr);
    host_barrier(sh, 4, n);
    return (hbad || herr) ? 1 : 0;
}

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2, rounds = argc > 2 ? atoi(argv[2]) : 50;
    if (n < 2 || n > kMaxRanks) return 1;
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset(sh, 0, sizeof(Shared));
    pid_t kids[kMaxRanks];
    for (int r = 1; r < n; ++r) {
        kids[r] = fork();
        if (kids[r] == 0) _exit(run(r, n, rounds, sh));   // HIP is first touched after the fork
    }
    int rc = run(0, n, rounds, sh);
    for (int r = 1; r < n; ++r) { int s = 0; waitpid(kids[r], &s, 0); rc |= WEXITSTATUS(s); }
    printf("ipc_probe: %s\n", rc ? "FAILED" : "ok");
    return rc;
}
