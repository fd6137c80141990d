// Who finishes when inside the production fused backward launch (bwd_pair_kernel: 256 input-gradient workgroups
// on 32x32 tiles || 256 weight-gradient workgroups on 64x64 tiles, gradient stored || 32 bias-gradient workgroups
// || 256 deferred-Adam workgroups) at 256 x 1024 x 1024 on random operands: 100 MHz wall-clock marks per
// workgroup (PVAE_TIMELINE), relative to the first workgroup's entry.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pair_timeline.hip -o ab_libs/pair_timeline
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define PVAE_TIMELINE 1
__device__ unsigned long long* g_timeline;
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fill_kernel(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += gridDim.x * 256ull) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = ((int)(x & 0xffff) - 32768) * (1.0f / 32768.0f);
    }
}

int main() {
    const int M = 256, N = 1024, K = 1024;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t nw = (size_t)N * K + N;
    float *dZ, *W, *X, *act, *dX, *G, *p2, *g2, *m2, *v2; unsigned long long* T;
    CK(hipMalloc(&dZ, (size_t)M * N * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&X, (size_t)M * K * 4));
    CK(hipMalloc(&act, (size_t)M * K * 4)); CK(hipMalloc(&dX, (size_t)M * K * 4)); CK(hipMalloc(&G, nw * 4));
    CK(hipMalloc(&p2, nw * 4)); CK(hipMalloc(&g2, nw * 4)); CK(hipMalloc(&m2, nw * 4)); CK(hipMalloc(&v2, nw * 4));
    CK(hipMalloc(&T, (size_t)2048 * 8 * 8)); CK(hipMemset(T, 0, (size_t)2048 * 8 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &T, sizeof(T)));
    float* bufs[] = {dZ, W, X, act, p2, g2, m2, v2};
    const size_t lens[] = {(size_t)M * N, nw, (size_t)M * K, (size_t)M * K, nw, nw, nw, nw};
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, bufs[i], lens[i], 17u * (i + 1));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, v2, nw, 99u);
    CK(hipStreamSynchronize(st));
    AdamSeg ad;
    ad.p = p2; ad.g = g2; ad.m = m2; ad.v = v2; ad.n4 = (long long)(nw / 4);
    ad.s = AdamScalars{5e-6f, 1.f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.001f};
    const EpiMask ed{dX, K, act, K};
    EpiGradStore es{G, K};
    es.gb = G + (size_t)N * K;
    const GemmGrid g1 = make_grid(M, K, 32, 32), gg2 = make_grid(N, K, 64, 64);
    GemmArgs gw{dZ, N, X, K, M, gg2.tiles_q, gg2.tiles_p, gg2.p_per_xcd};
    const int nd = g1.grid, nwg = gg2.grid, nb = bias_tiles(gw), na = adam_blocks(&ad), grid = nd + nwg + nb + na;
    auto launch = [&]() {
        hipLaunchKernelGGL((bwd_pair_kernel<EpiMask, EpiGradStore, 0>), dim3(grid), dim3(256), 0, st,
                           PVAE_GA2_PASS((GemmArgs{dZ, N, W, K, N, g1.tiles_q, g1.tiles_p, g1.p_per_xcd}), gw), ed, es, ad);
    };
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < 200; ++i) launch();
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)grid * 8);
    CK(hipMemcpy(h.data(), T, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < grid; ++w) if (h[(size_t)w * 8]) t0 = std::min(t0, h[(size_t)w * 8]);
    printf("launch period %.2f us back to back; %d + %d + %d + %d workgroups\n", ms * 1e3 / 200, nd, nwg, nb, na);
    auto stat = [&](const char* name, int lo, int hi, int id) {
        std::vector<double> x;
        for (int w = lo; w < hi; ++w) if (h[(size_t)w * 8 + id]) x.push_back((double)((long long)(h[(size_t)w * 8 + id] - t0)) * 0.01);
        if (x.empty()) { printf("   %-46s (no marks)\n", name); return; }
        std::sort(x.begin(), x.end());
        printf("   %-46s min %6.2f  median %6.2f  max %6.2f us\n", name, x.front(), x[x.size() / 2], x.back());
    };
    stat("input-gradient workgroups: entered", 0, nd, 0);
    stat("input-gradient workgroups: main loop done", 0, nd, 2);
    stat("input-gradient workgroups: finished", 0, nd, 3);
    stat("weight-gradient workgroups: entered", nd, nd + nwg, 0);
    stat("weight-gradient workgroups: contraction done", nd, nd + nwg, 2);
    stat("weight-gradient workgroups: finished", nd, nd + nwg, 3);
    stat("bias-gradient workgroups: finished", nd + nwg, nd + nwg + nb, 3);
    stat("deferred-Adam workgroups: finished", nd + nwg + nb, grid, 3);
    return 0;
}
