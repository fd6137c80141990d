// Weight-gradient kernel (gemm_wgrad_reg_kernel, 1024x1024 output, 64x64 tiles, one workgroup per
// CU): launch period vs batch rows (K of the contraction) and per-iteration cost with parts of the
// loop switched off (ABL bits: 1 no global loads / LDS writes, 2 no LDS fragment reads, 4 no MFMA,
// 8 no barriers).  One iteration = BK 32 rows = 32 MFMAs per wave (436 ns of matrix-pipe issue).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_probe.hip -o tools/wgrad_probe.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ABL>
static float run(const float* dZ, const float* X, float* G, int rows, hipStream_t st) {
    const int N = 1024, Kin = 1024;
    const GemmGrid g = make_grid(N, Kin, 64, 64);
    EpiGradStore e{G, Kin};
    e.gb = G + (size_t)N * Kin;
    auto go = [&]() {
        hipLaunchKernelGGL((gemm_wgrad_reg_kernel<EpiGradStore, ABL>), dim3(g.grid), dim3(256), 0, st,
                           PVAE_GA_PASS((GemmArgs{dZ, N, X, Kin, rows, g.tiles_q, g.tiles_p, g.p_per_xcd})), g.grid, e, AdamSeg());
    };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) go();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    const int iters = 100;
    for (int i = 0; i < iters; ++i) go();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *dZ, *X, *G;
    CK(hipMalloc(&dZ, (size_t)4096 * 1024 * 4)); CK(hipMalloc(&X, (size_t)4096 * 1024 * 4));
    CK(hipMalloc(&G, (size_t)1025 * 1024 * 4));
    CK(hipMemset(dZ, 0, (size_t)4096 * 1024 * 4)); CK(hipMemset(X, 0, (size_t)4096 * 1024 * 4));
    printf("rows (K)   period us\n");
    float t32 = 0, t2048 = 0;
    for (int rows : {32, 64, 128, 256, 512, 1024, 2048}) {
        const float t = run<0>(dZ, X, G, rows, st);
        if (rows == 32) t32 = t;
        if (rows == 2048) t2048 = t;
        printf("%8d   %8.2f\n", rows, t);
    }
    printf("full loop: %.0f ns per 32-row iteration (MFMA issue alone: 436)\n", (t2048 - t32) / 63 * 1e3);
#define ROW(name, ABL) printf("%-28s %.0f ns per iteration\n", name, (run<ABL>(dZ, X, G, 2048, st) - run<ABL>(dZ, X, G, 32, st)) / 63 * 1e3)
    ROW("no global loads / LDS writes", 1);
    ROW("no LDS fragment reads", 2);
    ROW("no MFMA", 4);
    ROW("no barriers", 8);
    ROW("MFMA only (1+2+8)", 11);
    ROW("loads + LDS only (4)", 4);
    return 0;
}
