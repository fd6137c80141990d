// Timeline of the horizontally fused backward launch (bwd_pair_kernel: blocks [0,256) = input
// gradient 256x1024x1024 on 32x32 tiles, blocks [256,512) = weight gradient 1024x1024 (K = 256
// rows) on 64x64 tiles with Adam in the epilogue): when does each kind of workgroup start and end,
// and on which CU does it run (do a dgrad and a wgrad workgroup really share every CU?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pair_probe.hip -o tools/pair_probe.out
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <type_traits>
#include <vector>
#define PVAE_TIMELINE 1
__device__ unsigned long long* g_timeline;
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
// bwd_pair with gradient store, plus `na` extra workgroups that apply Adam to ANOTHER layer's
// stored gradient (p, m, v, g streamed from memory) while the contractions run
template <class EpiD, class EpiW>
__global__ void __launch_bounds__(256)
pair_plus_adam(GemmArgs gd, EpiD ed, int nd, GemmArgs gw, EpiW ew, int nw, float* p, const float* g, float* m, float* v,
               long long n4, AdamScalars as, int na, int adam_first) {
    __shared__ __attribute__((aligned(16))) float lds[kRegRingFloats];
    int b = blockIdx.x;
    if (adam_first) b = b < na ? nd + nw + bias_tiles(gw) + b : b - na;
    if (b < nd) splitk_reg_body<false, EpiD, 0>(lds, b, gd, ed);
    else if (b < nd + nw) wgrad_body<EpiW, 0>(lds, b - nd, gw, ew);
    else if (b < nd + nw + bias_tiles(gw)) bias_grad_body(lds, b - nd - nw, gw, ew);
    else {
        const int a = b - nd - nw - bias_tiles(gw);
        for (long long i = a * 256ll + threadIdx.x; i < n4; i += na * 256ll) {
            v4f pp = reinterpret_cast<v4f*>(p)[i];
            const v4f gg = reinterpret_cast<const v4f*>(g)[i];
            v4f mm = reinterpret_cast<v4f*>(m)[i];
            v4f vv = reinterpret_cast<v4f*>(v)[i];
            adam_update4(gg, pp, mm, vv, as);
            reinterpret_cast<v4f*>(p)[i] = pp;
            reinterpret_cast<v4f*>(m)[i] = mm;
            reinterpret_cast<v4f*>(v)[i] = vv;
        }
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define CK2(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) printf("%s: %s\n", #x, hipGetErrorString(e_)); } while (0)
int main() {
    const int M = 256, N = 1024, K = 1024;
    hipStream_t st; CK(hipStreamCreate(&st));
    float *dZ, *W, *X, *dX, *act, *m, *v; unsigned long long* T;
    const size_t nw = (size_t)N * K;
    CK(hipMalloc(&dZ, (size_t)M * N * 4)); CK(hipMalloc(&W, nw * 4 + N * 4)); CK(hipMalloc(&X, (size_t)M * K * 4));
    CK(hipMalloc(&dX, (size_t)M * K * 4)); CK(hipMalloc(&act, (size_t)M * K * 4));
    CK(hipMalloc(&m, nw * 4 + N * 4)); CK(hipMalloc(&v, nw * 4 + N * 4));
    CK(hipMalloc(&T, (size_t)2048 * 8 * 8));
    CK(hipMemset(dZ, 0, (size_t)M * N * 4)); CK(hipMemset(W, 0, nw * 4 + N * 4)); CK(hipMemset(X, 0, (size_t)M * K * 4));
    CK(hipMemset(act, 0, (size_t)M * K * 4)); CK(hipMemset(m, 0, nw * 4 + N * 4)); CK(hipMemset(v, 0, nw * 4 + N * 4));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &T, sizeof(T)));
    AdamScalars as{5e-4f, 1.f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.001f};
    EpiGradAdam e{W, m, v, K, as};
    e.b = W + nw; e.bm = m + nw; e.bv = v + nw;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto go = [&]() { return gemm_bwd_pair(dZ, N, W, K, act, K, dX, K, M, K, N, dZ, N, X, K, N, K, M, e, st); };
    for (int i = 0; i < 20; ++i) CK(go());
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    const int iters = 200;
    for (int i = 0; i < iters; ++i) CK(go());
    CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const int grid = 512;
    std::vector<unsigned long long> h((size_t)grid * 8);
    CK(hipMemcpy(h.data(), T, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int w = 0; w < grid; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
    printf("bwd_pair launch period %.2f us (back-to-back)\n", ms * 1e3 / iters);
    auto stat = [&](const char* name, int lo, int hi, int id) {
        std::vector<double> x;
        for (int w = lo; w < hi; ++w) x.push_back((double)(h[(size_t)w * 8 + id] - t0) * 0.01);
        std::sort(x.begin(), x.end());
        printf("   %-44s min %6.2f  median %6.2f  max %6.2f us\n", name, x.front(), x[x.size() / 2], x.back());
    };
    stat("dgrad workgroups: entered", 0, 256, 0);
    stat("dgrad workgroups: finished", 0, 256, 3);
    stat("wgrad workgroups: entered", 256, 512, 0);
    stat("wgrad workgroups: contraction done", 256, 512, 2);
    stat("wgrad workgroups: finished (Adam stored)", 256, 512, 3);
    // co-residency: (xcc, hw_id CU/SH/SE bits) of every workgroup
    std::map<unsigned long long, std::pair<int, int>> cu;
    for (int w = 0; w < grid; ++w) {
        const unsigned long long key = (h[(size_t)w * 8 + 7] << 32) | ((h[(size_t)w * 8 + 6] >> 8) & 0xff);
        if (w < 256) cu[key].first++; else cu[key].second++;
    }
    int both = 0, two_d = 0, two_w = 0;
    for (auto& kv : cu) {
        if (kv.second.first == 1 && kv.second.second == 1) ++both;
        if (kv.second.first >= 2) ++two_d;
        if (kv.second.second >= 2) ++two_w;
    }
    printf("   %zu distinct CUs used; %d hold one dgrad + one wgrad workgroup, %d hold >= 2 dgrad, %d hold >= 2 wgrad\n",
           cu.size(), both, two_d, two_w);
    // ablations of BOTH co-resident bodies: which resource do they fight over?
    int part = 3;                       // bit 0: dgrad workgroups present, bit 1: wgrad workgroups present
    auto run_abl = [&](auto tag, const char* name) {
        constexpr int A = decltype(tag)::value;
        const EpiMask ed{dX, K, act, K};
        const GemmGrid g1 = make_grid(M, K, 32, 32), g2 = make_grid(N, K, 64, 64);
        auto launch = [&]() {
            const int nd = (part & 1) ? g1.grid : 0, nw2 = (part & 2) ? g2.grid : 0;
            hipLaunchKernelGGL((bwd_pair_kernel<EpiMask, EpiGradAdam, A>), dim3(nd + nw2), dim3(256), 0, st,
                               PVAE_GA2_PASS((GemmArgs{dZ, N, W, K, N, nd ? g1.tiles_q : 0, g1.tiles_p, g1.p_per_xcd}),
                                             (GemmArgs{dZ, N, X, K, M, nw2 ? g2.tiles_q : 0, g2.tiles_p, g2.p_per_xcd})), ed, e, AdamSeg());
        };
        for (int i = 0; i < 10; ++i) launch();
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < 100; ++i) launch();
        hipEventRecord(b, st); hipEventSynchronize(b);
        float t; hipEventElapsedTime(&t, a, b);
        printf("   %-46s period %6.2f us\n", name, t * 10.0f);
        CK2(hipMemcpy(h.data(), T, h.size() * 8, hipMemcpyDeviceToHost));
        t0 = ~0ull;
        for (int w = 0; w < grid; ++w) t0 = std::min(t0, h[(size_t)w * 8]);
        const int nd = (part & 1) ? 256 : 0;
        t0 = ~0ull;
        for (int w = 0; w < nd + ((part & 2) ? 256 : 0); ++w) t0 = std::min(t0, h[(size_t)w * 8]);
        if (part & 1) stat("      dgrad finished", 0, 256, 3);
        if (part & 2) stat("      wgrad contraction done", nd, nd + 256, 2);
        if (part & 2) stat("      wgrad finished", nd, nd + 256, 3);
    };
    run_abl(std::integral_constant<int, 0>(), "both bodies complete");
    run_abl(std::integral_constant<int, 1>(), "no global loads / LDS writes");
    run_abl(std::integral_constant<int, 2>(), "no LDS fragment reads");
    run_abl(std::integral_constant<int, 4>(), "no MFMA");
    run_abl(std::integral_constant<int, 8>(), "no barriers");
    run_abl(std::integral_constant<int, 11>(), "MFMA only (no loads, LDS reads, barriers)");
    run_abl(std::integral_constant<int, 3>(), "MFMA + barriers (no loads, no LDS reads)");
    {
        // More waves per SIMD?  Same flops, same operand bytes, but every workgroup contracts over half
        // the length (twice as many workgroups, 4 co-resident per CU if registers allow).  Results
        // would need a cross-workgroup reduction; this only times the loops.
        float *gw, *big;
        CK(hipMalloc(&gw, (size_t)2048 * 1024 * 4)); CK(hipMalloc(&big, (size_t)2048 * 1024 * 4));
        CK(hipMemset(big, 0, (size_t)2048 * 1024 * 4));
        EpiGradStore es{gw, K};
        auto timeit = [&](const char* name, int dM, int dN, int wN, int wM) {
            const EpiMask ed{dX, K, act, K};
            const GemmGrid g1 = make_grid(dM, K, 32, 32), g2 = make_grid(wN, K, 64, 64);
            auto launch = [&]() {
                hipLaunchKernelGGL((bwd_pair_kernel<EpiMask, EpiGradStore, 0>), dim3(g1.grid + g2.grid), dim3(256), 0, st,
                                   PVAE_GA2_PASS((GemmArgs{big, dN, W, K, dN, g1.tiles_q, g1.tiles_p, g1.p_per_xcd}),
                                                 (GemmArgs{big, wN, X, K, wM, g2.tiles_q, g2.tiles_p, g2.p_per_xcd})), ed, es, AdamSeg());
            };
            for (int i = 0; i < 10; ++i) launch();
            hipStreamSynchronize(st);
            hipEventRecord(a, st);
            for (int i = 0; i < 100; ++i) launch();
            hipEventRecord(b, st); hipEventSynchronize(b);
            float t; hipEventElapsedTime(&t, a, b);
            printf("   %-60s %4d workgroups, period %6.2f us\n", name, g1.grid + g2.grid, t * 10.0f);
            std::vector<unsigned long long> hh((size_t)(g1.grid + g2.grid) * 8);
            CK2(hipMemcpy(hh.data(), T, hh.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long tt = ~0ull;
            for (int w = 0; w < g1.grid + g2.grid; ++w) tt = std::min(tt, hh[(size_t)w * 8]);
            auto st2 = [&](const char* nm, int lo, int hi, int id) {
                std::vector<double> x;
                for (int w = lo; w < hi; ++w) x.push_back((double)(hh[(size_t)w * 8 + id] - tt) * 0.01);
                std::sort(x.begin(), x.end());
                printf("         %-36s min %6.2f  median %6.2f  max %6.2f us\n", nm, x.front(), x[x.size() / 2], x.back());
            };
            st2("dgrad entered", 0, g1.grid, 0);
            st2("dgrad finished", 0, g1.grid, 3);
            st2("wgrad entered", g1.grid, g1.grid + g2.grid, 0);
            st2("wgrad contraction done", g1.grid, g1.grid + g2.grid, 2);
            st2("wgrad finished", g1.grid, g1.grid + g2.grid, 3);
        };
        {
            // deferred Adam: a second layer's p/m/v/g (same size), updated by extra workgroups
            float *p2, *g2, *m2, *v2;
            CK(hipMalloc(&p2, nw * 4)); CK(hipMalloc(&g2, nw * 4)); CK(hipMalloc(&m2, nw * 4)); CK(hipMalloc(&v2, nw * 4));
            CK(hipMemset(p2, 0, nw * 4)); CK(hipMemset(g2, 0, nw * 4)); CK(hipMemset(m2, 0, nw * 4)); CK(hipMemset(v2, 0, nw * 4));
            const EpiMask ed{dX, K, act, K};
            const GemmGrid g1 = make_grid(M, K, 32, 32), gg2 = make_grid(N, K, 64, 64);
            for (int first : {0, 1})
            for (int na : {0, 128, 256, 512}) {
                auto launch = [&]() {
                    hipLaunchKernelGGL((pair_plus_adam<EpiMask, EpiGradStore>), dim3(g1.grid + gg2.grid + gg2.tiles_q + na), dim3(256), 0, st,
                                       GemmArgs{dZ, N, W, K, N, g1.tiles_q, g1.tiles_p, g1.p_per_xcd}, ed, g1.grid,
                                       GemmArgs{dZ, N, X, K, M, gg2.tiles_q, gg2.tiles_p, gg2.p_per_xcd}, es, gg2.grid,
                                       p2, g2, m2, v2, (long long)(nw / 4), as, na, first);
                };
                for (int i = 0; i < 10; ++i) launch();
                hipStreamSynchronize(st);
                hipEventRecord(a, st);
                for (int i = 0; i < 100; ++i) launch();
                hipEventRecord(b, st); hipEventSynchronize(b);
                float t; hipEventElapsedTime(&t, a, b);
                printf("   gradient-store pair + %3d Adam workgroups on another layer (%s): period %6.2f us\n", na, first ? "dispatched first" : "dispatched last", t * 10.0f);
            }
        }
        float* dX2; CK(hipMalloc(&dX2, (size_t)512 * K * 4));
        timeit("gradient store: 256x1024 (k 1024) | 1024x1024 (k 256)", 256, 1024, 1024, 256);
        float* keep = dX; dX = dX2;
        float* keepa = act; act = big;
        timeit("half contraction: 512x1024 (k 512) | 2048x1024 (k 128)", 512, 512, 2048, 128);
        dX = keep; act = keepa;
    }
    for (int pp : {1, 2}) {
        part = pp;
        printf("--- only the %s workgroups\n", pp == 1 ? "dgrad" : "wgrad");
        run_abl(std::integral_constant<int, 0>(), "complete");
        run_abl(std::integral_constant<int, 4>(), "no MFMA");
        run_abl(std::integral_constant<int, 11>(), "MFMA only (no loads, LDS reads, barriers)");
        run_abl(std::integral_constant<int, 3>(), "MFMA + barriers (no loads, no LDS reads)");
        run_abl(std::integral_constant<int, 1>(), "no global loads / LDS writes");
    }
    return 0;
}
