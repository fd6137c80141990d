// Is the workgroup -> XCD assignment of a launch static (block b -> XCD b % 8, whatever is free there) or does the
// dispatcher route around a full XCD?  Decides what a persistent multi-layer kernel may assume when something else --
// the resident rollout server, another process's launches -- holds CUs of one XCD.
// An "occupant" launch keeps 150 KB of LDS on the 32 CUs of XCD 0 for ~20 ms; on a second stream a 256-workgroup
// launch with the forward kernel's footprint (512 threads, 96 KB LDS) follows at once.  Reported: where (XCC_ID) and
// when (wall clock relative to the occupant's end) each of its workgroups started.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xcd_dispatch_probe.hip -o ab_libs/xcd_dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u; }

__global__ void __launch_bounds__(256) occupant(unsigned long long* t, unsigned long long ticks, unsigned* xcc_of) {
    extern __shared__ float lds[];
    if ((blockIdx.x & 7) != 0) return;
    if (threadIdx.x == 0) {
        lds[0] = 1.f;
        xcc_of[blockIdx.x >> 3] = xcc_id();
        const unsigned long long t0 = wall_clock64();
        if (blockIdx.x == 0) t[0] = t0;
        while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
        if (blockIdx.x == 0) t[1] = wall_clock64();
    }
    __syncthreads();
}
__global__ void __launch_bounds__(512) census(unsigned long long* start, unsigned* xcc) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1.f;
        start[blockIdx.x] = wall_clock64();
        xcc[blockIdx.x] = xcc_id();
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 500) __builtin_amdgcn_s_sleep(8);        // 5 us of "work"
    }
    __syncthreads();
}

int main() {
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    unsigned long long *t, *start; unsigned *xcc, *oxcc;
    CK(hipMalloc(&t, 16)); CK(hipMalloc(&start, 256 * 8)); CK(hipMalloc(&xcc, 256 * 4)); CK(hipMalloc(&oxcc, 32 * 4));
    CK(hipFuncSetAttribute((const void*)occupant, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    CK(hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    for (int with_occupant = 0; with_occupant < 2; ++with_occupant) {
        CK(hipMemset(t, 0, 16));
        if (with_occupant) hipLaunchKernelGGL(occupant, dim3(256), dim3(256), 150 * 1024, s0, t, 2000000ull, oxcc);   // 20 ms
        // let the occupant get there first
        if (with_occupant) { hipEvent_t e; CK(hipEventCreate(&e)); for (volatile int i = 0; i < 2000000; ++i) {} }
        hipLaunchKernelGGL(census, dim3(256), dim3(512), 96 * 1024, s1, start, xcc);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> hs(256); std::vector<unsigned> hx(256), ho(32); unsigned long long ht[2];
        CK(hipMemcpy(hs.data(), start, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), xcc, 256 * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(ho.data(), oxcc, 32 * 4, hipMemcpyDeviceToHost));
        int per_xcc[8] = {0}, moved = 0, late = 0;
        unsigned long long first = ~0ull;
        for (int b = 0; b < 256; ++b) first = hs[b] < first ? hs[b] : first;
        for (int b = 0; b < 256; ++b) {
            per_xcc[hx[b]]++;
            if (hx[b] != (unsigned)(b & 7)) ++moved;
            if (with_occupant && hs[b] >= ht[1]) ++late;
        }
        printf("%s: census workgroups per XCC: %d %d %d %d %d %d %d %d; %d ran on an XCC other than b %% 8", with_occupant ? "with XCD 0's LDS held for 20 ms" : "alone",
               per_xcc[0], per_xcc[1], per_xcc[2], per_xcc[3], per_xcc[4], per_xcc[5], per_xcc[6], per_xcc[7], moved);
        if (with_occupant) {
            int occ_ok = 0; for (int i = 0; i < 32; ++i) occ_ok += ho[i] == ho[0];
            printf("; %d started only after the occupant ended (occupant on XCC %u, %d of 32 there; first census workgroup %.1f us after the occupant's start, occupant ran %.1f us)",
                   late, ho[0], occ_ok, (double)((long long)(first - ht[0])) * 0.01, (double)(ht[1] - ht[0]) * 0.01);
            double worst = 0; for (int b = 0; b < 256; ++b) { const double d = (double)((long long)(hs[b] - first)) * 0.01; worst = d > worst ? d : worst; }
            printf("; last census workgroup started %.1f us after the first", worst);
        }
        printf("\n");
    }
    return 0;
}
