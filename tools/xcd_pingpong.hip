// How long does ONE tagged 8-byte word take from a workgroup to a workgroup that polls it -- inside an XCD (same L2) and
// across XCDs (through the fabric)?  Two workgroups play ping-pong: A writes {round, value}, B polls until it sees the
// round, answers in its own word, A polls that.  Reported: microseconds per HOP (half a round trip), by where B runs
// (block 8 = the XCD of block 0; block 1 = the next XCD) and by the scope of the store / load pair (agent: sc1;
// system: sc0 sc1).  Decides whether a chip-wide rollout server (stacks too big for one XCD's LDS) can hand activations
// across XCDs at a price worth paying.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/xcd_pingpong.hip -o ab_libs/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int SCOPE>
__device__ inline void put(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE>
__device__ inline unsigned long long get(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }

template <int SCOPE>
__global__ void pingpong(unsigned long long* words, int partner_block, int rounds, unsigned long long* ticks, unsigned* xcc, int nread) {
    const int b = blockIdx.x;
    if (b != 0 && b != partner_block) return;
    if (threadIdx.x == 0) xcc[b == 0 ? 0 : 1] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
    unsigned long long* mine = words + (b == 0 ? 0 : 64);          // 512 bytes apart
    unsigned long long* theirs = words + (b == 0 ? 64 : 0);
    const int lane = threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        if (b == 0) {
            if (lane < nread) put<SCOPE>(mine + lane, ((unsigned long long)r << 32) | (unsigned)lane);
            if (lane < nread) { unsigned spins = 0; while ((unsigned)(get<SCOPE>(theirs + lane) >> 32) != (unsigned)r) if (++spins > (1u << 24)) break; }
        } else {
            if (lane < nread) { unsigned spins = 0; while ((unsigned)(get<SCOPE>(theirs + lane) >> 32) != (unsigned)r) if (++spins > (1u << 24)) break; }
            if (lane < nread) put<SCOPE>(mine + lane, ((unsigned long long)r << 32) | (unsigned)lane);
        }
    }
    if (b == 0 && lane == 0) ticks[0] = wall_clock64() - t0;
}

int main() {
    unsigned long long *words, *ticks; unsigned* xcc;
    CK(hipMalloc(&words, 4096)); CK(hipMalloc(&ticks, 8)); CK(hipMalloc(&xcc, 8));
    const int rounds = 20000;
    for (int nread : {1, 32, 64})
        for (int partner : {8, 1, 4}) {
            for (int scope = 0; scope < 2; ++scope) {
                CK(hipMemset(words, 0, 4096));
                if (scope == 0) hipLaunchKernelGGL((pingpong<__HIP_MEMORY_SCOPE_AGENT>), dim3(16), dim3(64), 0, 0, words, partner, rounds, ticks, xcc, nread);
                else hipLaunchKernelGGL((pingpong<__HIP_MEMORY_SCOPE_SYSTEM>), dim3(16), dim3(64), 0, 0, words, partner, rounds, ticks, xcc, nread);
                CK(hipDeviceSynchronize());
                unsigned long long t; unsigned x[2];
                CK(hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
                printf("%2d word(s) per hop, block 0 (XCC %u) <-> block %d (XCC %u), %s scope: %.3f us per hop\n", nread, x[0], partner, x[1],
                       scope == 0 ? "agent " : "system", t / 100.0 / rounds / 2.0);
            }
        }
    return 0;
}
