// Ablation of the ring kernels in isolation (guide section 7: "ablate before optimizing").
// Each variant is launched back-to-back on one stream; time per launch from HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../physicsvae_amd/csrc/pvae_gemm.h"
#include "gemm_dma_ring.h"
using namespace pvae;
static int g_pad = 0;   // extra floats of row pitch
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct EpiNop { float* out; int ld; struct Pre {}; __device__ Pre preload(int, int) const { return Pre(); } __device__ void operator()(int q, int p, v4f v, const Pre&) const { if (v.x == 123.456f) out[(size_t)q * ld + p] = v.y; } __device__ void finish(float*, int, int) const {} };

template <int ABL> float run_fwd(bool prow, const float* X, const float* W, float* out, int M, int N, int K, hipStream_t st, int iters, bool real_epi) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const GemmGrid g = make_grid(M, N, 32, 32);
    auto go = [&]() {
        if (real_epi) {
            EpiBiasAct e{out, N, nullptr, 1};
            if (prow) hipLaunchKernelGGL((gemm_splitk_kernel<true, 8, EpiBiasAct, ABL>), dim3(g.grid), dim3(256), 0, st, X, K + g_pad, W, K + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd, e);
            else hipLaunchKernelGGL((gemm_splitk_kernel<false, 8, EpiBiasAct, ABL>), dim3(g.grid), dim3(256), 0, st, X, K + g_pad, W, N + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd, e);
        } else {
            EpiNop e{out, N};
            if (prow) hipLaunchKernelGGL((gemm_splitk_kernel<true, 8, EpiNop, ABL>), dim3(g.grid), dim3(256), 0, st, X, K + g_pad, W, K + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd, e);
            else hipLaunchKernelGGL((gemm_splitk_kernel<false, 8, EpiNop, ABL>), dim3(g.grid), dim3(256), 0, st, X, K + g_pad, W, N + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd, e);
        }
    };
    for (int i = 0; i < 20; ++i) go();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
template <int ABL> float run_reg(bool prow, const float* X, const float* W, float* out, int M, int N, int K, hipStream_t st, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const GemmGrid g = make_grid(M, N, 32, 32);
    EpiBiasAct e{out, N, nullptr, 1};
    auto go = [&]() {
        if (prow) hipLaunchKernelGGL((gemm_splitk_reg_kernel<true, EpiBiasAct, ABL>), dim3(g.grid), dim3(256), 0, st, GemmArgs{X, K + g_pad, W, K + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd}, e);
        else hipLaunchKernelGGL((gemm_splitk_reg_kernel<false, EpiBiasAct, ABL>), dim3(g.grid), dim3(256), 0, st, GemmArgs{X, K + g_pad, W, N + g_pad, K, g.tiles_q, g.tiles_p, g.p_per_xcd}, e);
    };
    for (int i = 0; i < 20; ++i) go();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
template <int ST> float run_ws(bool prow, const float* X, const float* W, float* out, int M, int N, int K, hipStream_t st, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const GemmGrid g = make_grid(M, N, 32, 32);
    EpiBiasAct e{out, N, nullptr, 1};
    auto go = [&]() {
        if (prow) hipLaunchKernelGGL((gemm_splitk_wsN_kernel<true, ST, EpiBiasAct>), dim3(g.grid), dim3(512), 0, st, GemmArgs{X, K, W, K, K, g.tiles_q, g.tiles_p, g.p_per_xcd}, e);
        else hipLaunchKernelGGL((gemm_splitk_wsN_kernel<false, ST, EpiBiasAct>), dim3(g.grid), dim3(512), 0, st, GemmArgs{X, K, W, N, K, g.tiles_q, g.tiles_p, g.p_per_xcd}, e);
    };
    for (int i = 0; i < 20; ++i) go();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
// max |a-b| between two device buffers (n floats)
static float max_diff(const float* a, const float* b, size_t n) {
    std::vector<float> ha(n), hb(n);
    hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost);
    float m = 0; for (size_t i = 0; i < n; ++i) { float d = fabsf(ha[i] - hb[i]); if (d > m) m = d; } return m;
}
template <int ABL> float run_wgrad(const float* dZ, const float* X, float* out, int M, int N, int K, hipStream_t st, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const GemmGrid g = make_grid(N, K, 64, 64);
    EpiGradStore e{out, K};
    auto go = [&]() { hipLaunchKernelGGL((gemm_wgrad_kernel<8, EpiGradStore, ABL>), dim3(g.grid), dim3(256), 0, st, dZ, N, X, K, M, g.tiles_q, g.tiles_p, g.p_per_xcd, e); };
    for (int i = 0; i < 20; ++i) go();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int M = 256, N = 1024;
    float *X, *W, *out, *big;
    CK(hipMalloc(&X, 4096 * 4096 * 4)); CK(hipMalloc(&W, 4096 * 4096 * 4)); CK(hipMalloc(&out, 4096 * 4096 * 4)); CK(hipMalloc(&big, 64 << 20));
    std::vector<float> h(4096 * 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    CK(hipMemcpy(X, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int it = 300;
    printf("forward 256x1024, K sweep (full kernel, real epilogue): ");
    for (int K : {64, 256, 512, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_fwd<0>(true, X, W, out, M, N, K, st, it, true));
    printf("\ndgrad   256x1024, K sweep: ");
    for (int K : {64, 256, 512, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_fwd<0>(false, X, W, out, M, N, K, st, it, true));
    printf("\nREG fwd 256x1024, K sweep: ");
    for (int K : {64, 256, 512, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_reg<0>(true, X, W, out, M, N, K, st, it));
    printf("\nREG dgrad 256x1024, K sweep: ");
    for (int K : {64, 256, 512, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_reg<0>(false, X, W, out, M, N, K, st, it));
    printf("\nREG fwd ablation K=4096: full %.2f  no-stage %.2f  no-LDSread %.2f  no-MFMA %.2f  MFMA-only %.2f",
           run_reg<0>(true, X, W, out, M, N, 4096, st, it), run_reg<1>(true, X, W, out, M, N, 4096, st, it),
           run_reg<2>(true, X, W, out, M, N, 4096, st, it), run_reg<4>(true, X, W, out, M, N, 4096, st, it),
           run_reg<11>(true, X, W, out, M, N, 4096, st, it));
    printf("\nWS(8 stages) fwd 256x1024, K sweep: ");
    for (int K : {256, 448, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_ws<8>(true, X, W, out, M, N, K, st, it));
    printf("\nWS(8 stages) dgrad 256x1024, K sweep: ");
    for (int K : {256, 448, 1024, 2048, 4096}) printf("K=%d %.2fus  ", K, run_ws<8>(false, X, W, out, M, N, K, st, it));
    printf("\nWS stage sweep, K=1024 fwd/dgrad, K=4096 fwd: ");
    printf(" S=3 %.2f/%.2f/%.2f", run_ws<3>(true, X, W, out, M, N, 1024, st, it), run_ws<3>(false, X, W, out, M, N, 1024, st, it), run_ws<3>(true, X, W, out, M, N, 4096, st, it));
    printf(" S=4 %.2f/%.2f/%.2f", run_ws<4>(true, X, W, out, M, N, 1024, st, it), run_ws<4>(false, X, W, out, M, N, 1024, st, it), run_ws<4>(true, X, W, out, M, N, 4096, st, it));
    printf(" S=5 %.2f/%.2f/%.2f", run_ws<5>(true, X, W, out, M, N, 1024, st, it), run_ws<5>(false, X, W, out, M, N, 1024, st, it), run_ws<5>(true, X, W, out, M, N, 4096, st, it));
    printf(" S=6 %.2f/%.2f/%.2f", run_ws<6>(true, X, W, out, M, N, 1024, st, it), run_ws<6>(false, X, W, out, M, N, 1024, st, it), run_ws<6>(true, X, W, out, M, N, 4096, st, it));
    {   // correctness of WS vs REG on the same data
        run_reg<0>(true, X, W, out, M, N, 1024, st, 1); 
        run_ws<8>(true, X, W, big, M, N, 1024, st, 1);
        hipDeviceSynchronize();
        printf("\nWS vs REG fwd max|diff| = %g", max_diff(out, big, (size_t)M * N));
        run_reg<0>(false, X, W, out, M, N, 1024, st, 1);
        run_ws<8>(false, X, W, big, M, N, 1024, st, 1);
        hipDeviceSynchronize();
        printf(" ; dgrad max|diff| = %g", max_diff(out, big, (size_t)M * N));
    }
    for (int pad : {0}) {
        g_pad = pad;
        printf("\npitch +%d floats:  DMA fwd K=1024 %.2f K=4096 %.2f | REG fwd K=1024 %.2f K=4096 %.2f | DMA dgrad K=1024 %.2f | REG dgrad K=1024 %.2f | DMA-only(no MFMA/LDS) K=4096 %.2f | REG stage-only %.2f",
               pad, run_fwd<0>(true, X, W, out, M, N, 1024, st, it, true), run_fwd<0>(true, X, W, out, M, N, 4096, st, it, true),
               run_reg<0>(true, X, W, out, M, N, 1024, st, it), run_reg<0>(true, X, W, out, M, N, 4096, st, it),
               run_fwd<0>(false, X, W, out, M, N, 1024, st, it, true), run_reg<0>(false, X, W, out, M, N, 1024, st, it),
               run_fwd<6>(true, X, W, out, M, N, 4096, st, it, false), run_reg<4>(true, X, W, out, M, N, 4096, st, it));
    }
    g_pad = 0;
    printf("\nwgrad 1024x1024, M(batch) sweep: ");
    for (int B : {32, 128, 256, 512, 1024, 2048}) printf("B=%d %.2fus  ", B, run_wgrad<0>(X, W, out, B, 1024, 1024, st, it));
    printf("\n\nablation at K=4096 (64 k-tiles; per-tile cost = (t - t(K=64)) / 63):\n");
    auto row = [&](const char* name, float t4096, float t64) { printf("  %-34s %8.2f us   per k-tile %6.1f ns\n", name, t4096, (t4096 - t64) / 63.f * 1e3f); };
    row("fwd full", run_fwd<0>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<0>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-DMA", run_fwd<1>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<1>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-LDSread", run_fwd<2>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<2>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-MFMA", run_fwd<4>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<4>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-DMA no-LDSread", run_fwd<3>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<3>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-DMA no-LDS no-barrier (MFMA)", run_fwd<11>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<11>(true, X, W, out, M, N, 64, st, it, false));
    row("fwd no-MFMA no-LDSread (DMA+bar)", run_fwd<6>(true, X, W, out, M, N, 4096, st, it, false), run_fwd<6>(true, X, W, out, M, N, 64, st, it, false));
    row("dgrad full", run_fwd<0>(false, X, W, out, M, N, 4096, st, it, false), run_fwd<0>(false, X, W, out, M, N, 64, st, it, false));
    row("dgrad no-DMA", run_fwd<1>(false, X, W, out, M, N, 4096, st, it, false), run_fwd<1>(false, X, W, out, M, N, 64, st, it, false));
    row("dgrad no-LDSread", run_fwd<2>(false, X, W, out, M, N, 4096, st, it, false), run_fwd<2>(false, X, W, out, M, N, 64, st, it, false));
    printf("wgrad ablation at batch 2048 (64 k-tiles):\n");
    row("wgrad full", run_wgrad<0>(X, W, out, 2048, 1024, 1024, st, it), run_wgrad<0>(X, W, out, 32, 1024, 1024, st, it));
    row("wgrad no-DMA", run_wgrad<1>(X, W, out, 2048, 1024, 1024, st, it), run_wgrad<1>(X, W, out, 32, 1024, 1024, st, it));
    row("wgrad no-LDSread", run_wgrad<2>(X, W, out, 2048, 1024, 1024, st, it), run_wgrad<2>(X, W, out, 32, 1024, 1024, st, it));
    row("wgrad no-MFMA", run_wgrad<4>(X, W, out, 2048, 1024, 1024, st, it), run_wgrad<4>(X, W, out, 32, 1024, 1024, st, it));
    row("wgrad MFMA only", run_wgrad<11>(X, W, out, 2048, 1024, 1024, st, it), run_wgrad<11>(X, W, out, 32, 1024, 1024, st, it));
    printf("ideal per k-tile at 2.4 GHz: fwd/dgrad 16 MFMA x 32 cyc = 213 ns ; wgrad 32 MFMA x 32 cyc = 427 ns\n");
    return 0;
}
