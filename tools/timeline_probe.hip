// Where does a layer launch spend its time?  Runs the PRODUCTION forward kernel
// (gemm_splitk_ws_kernel<P_ROW, EpiBiasAct>, rows x 1024 x K) with wall-clock marks (PVAE_MARK,
// 100 MHz) and ablation switches (PVAE_PROBE) compiled in, and prints per configuration: the launch
// period of back-to-back dependent launches (HIP events) and the per-workgroup timeline relative to
// the first workgroup's entry.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/timeline_probe.hip -o tools/timeline_probe.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define PVAE_TIMELINE 1
__device__ unsigned long long* g_timeline;
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int M = 256, N = 1024;
    hipStream_t st; CK(hipStreamCreate(&st));
    float *X, *W, *B, *O; unsigned long long* T;
    CK(hipMalloc(&X, (size_t)M * 1024 * 4)); CK(hipMalloc(&W, (size_t)N * 1024 * 4));
    CK(hipMalloc(&B, N * 4)); CK(hipMalloc(&O, (size_t)M * N * 4));
    CK(hipMalloc(&T, (size_t)2048 * 8 * 8));
    CK(hipMemset(X, 0, (size_t)M * 1024 * 4)); CK(hipMemset(W, 0, (size_t)N * 1024 * 4)); CK(hipMemset(B, 0, N * 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &T, sizeof(T)));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    // mode: 0 normal, 2 every step re-reads tile 0 (cache-resident source), 4 no MFMAs, 6 no DMA (the
    // loader waves only keep the barriers).  NOTE: the hooks change the compiler's wait-count
    // placement in the compute loop (lgkmcnt(0) instead of the library build's lgkmcnt(4)), so
    // the probe's per-tile time is ~80 ns above the library's (272 ns, from rocprof K=256 vs 1024).
    struct Cfg { int K, rows, mode; };
    const Cfg cfgs[] = {{64, 256, 0}, {256, 256, 0}, {1024, 256, 0}, {1024, 32, 0}, {1024, 256, 2},
                        {1024, 256, 4}, {1024, 256, 6}};
    for (const Cfg& c : cfgs) {
        const int K = c.K;
        const GemmGrid g = make_grid(c.rows, N, 32, 32);
        EpiBiasAct e{O, N, B, 1};
        GemmArgs ga{X, K, W, K, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
        ga.krot = c.mode;
        auto go = [&]() {
            hipLaunchKernelGGL((gemm_splitk_ws_kernel<true, EpiBiasAct>), dim3(g.grid), dim3(512), 0, st, PVAE_GA_PASS(ga), e);
        };
        for (int i = 0; i < 20; ++i) go();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        const int iters = 200;
        for (int i = 0; i < iters; ++i) go();
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h((size_t)g.grid * 8);
        CK(hipMemcpy(h.data(), T, h.size() * 8, hipMemcpyDeviceToHost));     // marks of the last launch
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < g.grid; ++w) t0 = std::min(t0, h[(size_t)w * 8 + 0]);
        const char* names[6] = {"entered", "tile 0 seen by compute", "main loop done", "epilogue issued",
                                "prologue loads issued", "tile 0 landed (loader)"};
        printf("K=%4d  %3d workgroups  mode %d  launch period %.2f us (back-to-back, same stream)\n", K, g.grid, c.mode,
               ms * 1e3 / iters);
        for (int id : {0, 4, 5, 1, 2, 3}) {
            std::vector<double> v;
            for (int w = 0; w < g.grid; ++w) v.push_back((double)(h[(size_t)w * 8 + id] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            printf("   %-24s min %6.2f  median %6.2f  max %6.2f us after the first workgroup entered\n", names[id],
                   v.front(), v[v.size() / 2], v.back());
        }
    }
    return 0;
}
