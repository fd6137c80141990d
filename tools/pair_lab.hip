// Ablations of the fused backward launch (bwd_pair_kernel: 256 input-gradient workgroups on 32x32 tiles ||
// 256 weight-gradient workgroups on 64x64 tiles || bias-gradient workgroups || 256 deferred-Adam workgroups)
// at the BASELINE sizes (256 x 1024 x 1024), on RANDOM operands (zero-filled buffers clock ~19 % higher,
// MI355X_MICROARCH.md DVFS note), timed as the period of back-to-back launches on one stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pair_lab.hip -o ab_libs/pair_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
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

template <int ABL>
static float time_pair(hipStream_t st, const float* dZ, const float* W, const float* X, const float* act, float* dX,
                       float* G, int M, int N, int K, int parts, const AdamSeg& ad, hipEvent_t a, hipEvent_t b) {
    const EpiMask ed{dX, K, act, K};
    EpiGradStore es{G, K};
    es.gb = G + (size_t)N * K;
    const GemmGrid g1 = make_grid(M, K, 32, 32), g2 = make_grid(N, K, 64, 64);
    const int nd = (parts & 1) ? g1.grid : 0, nw = (parts & 2) ? g2.grid : 0;
    GemmArgs gw{dZ, N, X, K, M, g2.tiles_q, g2.tiles_p, g2.p_per_xcd};
    const int nb = (parts & 4) ? bias_tiles(gw) : 0;
    if (!(parts & 4)) es.gb = nullptr;
    const AdamSeg adv = (parts & 8) ? ad : AdamSeg();
    GemmArgs gw_ = gw;
    if (!nw) gw_.tiles_q = 0;                     // (an absent body: no tiles; bias_tiles() reads tiles_q ... see nb above)
    auto launch = [&]() {
        hipLaunchKernelGGL((bwd_pair_kernel<EpiMask, EpiGradStore, ABL>), dim3(nd + nw + nb + adam_blocks(&adv)), dim3(256), 0, st,
                           PVAE_GA2_PASS((GemmArgs{dZ, N, W, K, N, nd ? g1.tiles_q : 0, g1.tiles_p, g1.p_per_xcd}), gw_), ed, es, adv);
    };
    for (int i = 0; i < 20; ++i) launch();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    const int iters = 300;
    for (int i = 0; i < iters; ++i) launch();
    hipEventRecord(b, st);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

__global__ void empty_kernel(int) {}

// Shader clock actually sustained under a given load: every wave of a 1024-workgroup grid issues `n`
// dependent-free MFMAs on `live` operands (random: realistic switching power; zeros: the optimistic case)
// and workgroup 0 reports shader cycles (s_memtime) against the 100 MHz wall clock.
__global__ void __launch_bounds__(256) clock_kernel(const float* __restrict__ src, float* __restrict__ sink, int n,
                                                    unsigned long long* out) {
    float a[8], b[8];              // eight operand pairs per lane, cycled (consecutive MFMAs see different values)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = src[(threadIdx.x + 256 * (blockIdx.x & 255) + 4099 * u) & 65535];
        b[u] = src[(7919 + threadIdx.x + 6151 * u) & 65535];
    }
    v4f acc[4] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; i += 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[u & 3], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    const v4f s4 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (s4[0] == 123.456f) sink[threadIdx.x] = s4[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

int main() {
    const int M = 256, N = 1024, K = 1024;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t nw = (size_t)N * K + N;
    float *dZ, *W, *X, *act, *dX, *G, *p2, *g2, *m2, *v2;
    CK(hipMalloc(&dZ, (size_t)M * N * 4)); CK(hipMalloc(&W, nw * 4)); CK(hipMalloc(&X, (size_t)M * K * 4));
    CK(hipMalloc(&act, (size_t)M * K * 4)); CK(hipMalloc(&dX, (size_t)M * K * 4)); CK(hipMalloc(&G, nw * 4));
    CK(hipMalloc(&p2, nw * 4)); CK(hipMalloc(&g2, nw * 4)); CK(hipMalloc(&m2, nw * 4)); CK(hipMalloc(&v2, nw * 4));
    float* bufs[] = {dZ, W, X, act, p2, g2, m2, v2};
    const size_t lens[] = {(size_t)M * N, nw, (size_t)M * K, (size_t)M * K, nw, nw, nw, nw};
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, bufs[i], lens[i], 17u * (i + 1));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, v2, nw, 99u);      // second moment: positive
    CK(hipStreamSynchronize(st));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    AdamSeg ad;
    ad.p = p2; ad.g = g2; ad.m = m2; ad.v = v2; ad.n4 = (long long)(nw / 4);
    ad.s = AdamScalars{5e-6f, 1.f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.001f};
    {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, 0);
        hipStreamSynchronize(st); hipEventRecord(a, st);
        for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, st, 0);
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("empty 256-workgroup kernel, back to back: %.2f us\n", ms * 1e3f / 300);
    }
    {
        unsigned long long* d_out; CK(hipMalloc(&d_out, 16));
        float* zeros; CK(hipMalloc(&zeros, 65536 * 4)); CK(hipMemset(zeros, 0, 65536 * 4));
        for (int pass = 0; pass < 2; ++pass) {
            const float* src = pass ? zeros : W;
            for (int rep = 0; rep < 12; ++rep) hipLaunchKernelGGL(clock_kernel, dim3(1024), dim3(256), 0, st, src, dX, 4096, d_out);
            CK(hipStreamSynchronize(st));
            unsigned long long h[2]; CK(hipMemcpy(h, d_out, 16, hipMemcpyDeviceToHost));
            const double us = h[1] / 100.0, ghz = h[0] / (us * 1e3), per = (double)h[0] / (4096.0 * 4);
            printf("MFMA-saturated chip (1024 workgroups, 4 waves/SIMD... 16384 MFMAs per wave), %s operands: %.0f us, shader clock "
                   "%.3f GHz, %.1f cycles per MFMA per wave -> fp32 MFMA peak at this clock %.1f TFLOP/s\n",
                   pass ? "ZERO" : "random", us, ghz, per, 256 * 4 * 2048.0 / 32.0 * ghz * 1e9 / 1e12);
        }
    }
    printf("variant: production stores (sc1 write-through)\n");
#define T(ABL, parts, name) printf("  %-64s %6.2f us\n", name, time_pair<ABL>(st, dZ, W, X, act, dX, G, M, N, K, parts, ad, a, b))
    T(0, 15, "complete: dgrad || wgrad || bias || 256 Adam workgroups");
    T(0, 7, "no Adam workgroups");
    T(0, 3, "dgrad || wgrad only");
    T(0, 1, "dgrad workgroups only (256)");
    T(0, 2, "wgrad workgroups only (256)");
    T(0, 8, "Adam workgroups only (256)");
    {
        T(4, 3, "dgrad || wgrad, no MFMA");
        T(1, 3, "dgrad || wgrad, no global loads / LDS writes");
        T(2, 3, "dgrad || wgrad, no LDS fragment reads");
        T(8, 3, "dgrad || wgrad, no barriers");
        T(11, 3, "dgrad || wgrad, MFMA only");
        T(3, 3, "dgrad || wgrad, MFMA + barriers");
        T(4, 1, "dgrad only, no MFMA");
        T(11, 1, "dgrad only, MFMA only");
        T(1, 1, "dgrad only, no global loads / LDS writes");
        T(4, 2, "wgrad only, no MFMA");
        T(11, 2, "wgrad only, MFMA only");
        T(1, 2, "wgrad only, no global loads / LDS writes");
    }
    return 0;
}
