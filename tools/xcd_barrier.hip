// What does a barrier among the 32 workgroups of ONE XCD cost (all 8 XCDs doing the same at once), when it is
// built on that XCD's own L2: arrive = atomic add WITHOUT sc1 (executes in the L2), poll = sc1 load (bypasses
// L1, served by the L2), payload = plain stores (write-through L1 -> L2) drained with vmcnt(0) before arriving
// and read back with sc1 loads after leaving?  Checked: every workgroup verifies the words its 31 neighbours
// wrote before the barrier.  One 256-thread workgroup per CU; workgroup i runs on XCD i % 8 (census first).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/xcd_barrier.hip -o ab_libs/xcd_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
}

// payload: each workgroup publishes `words` floats per round into slab[xcd][rank][words]
__global__ void __launch_bounds__(256)
barrier_kernel(unsigned* counters, float* slab, int rounds, int words, unsigned* bad, unsigned long long* t, int mode,
               const float4* stream, size_t stream_n4) {
    __shared__ float sink[256];
    __shared__ int s_spun;
    const int xcd = blockIdx.x & 7, rank = blockIdx.x >> 3;            // 32 ranks per XCD
    unsigned* ctr = counters + xcd * 64;                               // one 256-byte line per XCD
    unsigned errors = 0;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 1; r <= rounds; ++r) {
        if (mode & 2) {
            // what a layer's main loop does to the memory system: 256 KB of operand loads per workgroup and round
            // (mode & 4: the same loads WITHOUT the barrier = the control)
            size_t base = ((size_t)blockIdx.x * 16384 + (size_t)r * 1048576 * 3) % (stream_n4 - 16384);
#pragma unroll 8
            for (int i = 0; i < 64; ++i) {
                const float4 v = stream[base + (size_t)i * 256 + threadIdx.x];
                sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
            }
        }
        if (mode & 4) continue;
        if (mode & 1) {
            // two slabs by round parity: a neighbour that is already a round ahead writes the OTHER one
            float* mine = slab + (size_t)(r & 1) * 8 * 32 * 4096 + ((size_t)xcd * 32 + rank) * words;
            for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(r * 1000 + rank);      // plain stores
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);       // no sc1: in the L2
            const unsigned want = 32u * r;
            int spins = 0;
            while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {    // sc1 load
                if (++spins > (1 << 20)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            s_spun = spins;
        }
        __syncthreads();
        if (s_spun > (1 << 20)) { if (threadIdx.x == 0) atomicOr(bad, 0x80000000u); break; }
        if (mode & 1) {
            // read one word of every neighbour's payload (sc1: past the L1)
            if (threadIdx.x < 32) {
                const float* theirs = slab + (size_t)(r & 1) * 8 * 32 * 4096 + ((size_t)xcd * 32 + threadIdx.x) * words + (r % words);
                const float v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v != (float)(r * 1000 + (int)threadIdx.x)) { ++errors; if (v < (float)(r * 1000)) errors += 0x10000; }   // high half: words from an EARLIER round
                acc += v;
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    sink[threadIdx.x] = acc + sacc.x + sacc.y + sacc.z + sacc.w;
    if (sink[threadIdx.x] == 12345.678f) bad[0] = 1;
    if (errors) atomicAdd(bad, errors);
    if (threadIdx.x == 0) t[blockIdx.x] = t1 - t0;
}

int main() {
    unsigned *d_census, *counters, *bad; float* slab; unsigned long long* t;
    CK(hipMalloc(&d_census, 256 * 4)); CK(hipMalloc(&counters, 8 * 256)); CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&slab, (size_t)2 * 8 * 32 * 4096 * 4)); CK(hipMalloc(&t, 256 * 8));
    hipLaunchKernelGGL(census, dim3(256), dim3(64), 0, 0, d_census);
    unsigned h[256];
    CK(hipMemcpy(h, d_census, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 1;
    for (int i = 0; i < 256; ++i) if (h[i] != h[i & 7]) ok = 0;
    printf("workgroup i on XCD i %% 8: %s (XCC ids of workgroups 0..7: %u %u %u %u %u %u %u %u)\n", ok ? "yes" : "NO",
           h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    if (!ok) return 0;
    const int rounds = 2000;
    float4* stream; const size_t stream_n4 = (size_t)64 * 1024 * 1024 / 16;
    CK(hipMalloc(&stream, stream_n4 * 16)); CK(hipMemset(stream, 0, stream_n4 * 16));
    for (int mode : {0, 1, 6, 2, 3})
        for (int words : {256, 1024, 4096}) {
            if (mode != 1 && mode != 3 && words != 1024) continue;
            CK(hipMemset(counters, 0, 8 * 256)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(slab, 0, (size_t)2 * 8 * 32 * 4096 * 4));
            hipLaunchKernelGGL(barrier_kernel, dim3(256), dim3(256), 0, 0, counters, slab, rounds, words, bad, t, mode, stream, stream_n4);
            CK(hipDeviceSynchronize());
            unsigned long long ht[256]; unsigned hb;
            CK(hipMemcpy(ht, t, sizeof(ht), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (int i = 0; i < 256; ++i) mx = ht[i] > mx ? ht[i] : mx;
            const char* what[8] = {"barrier alone", "barrier + payload", "256 KB of loads per workgroup + barrier", "256 KB of loads + barrier + payload", "", "",
                                   "256 KB of loads per workgroup, NO barrier (control)", ""};
            printf("%-52s payload %5d B: ", what[mode], (mode & 1) ? words * 4 : 0);
            printf("%.2f us per round, %s\n", mx / 100.0 / rounds, hb == 0 ? "all words fresh" : (hb & 0x80000000u ? "TIMED OUT" : "STALE WORDS SEEN"));
            if (hb) printf("   (error word 0x%x)\n", hb);
        }
    return 0;
}
