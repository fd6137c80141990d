// The last launch of a backward pass: weight gradients (+Adam) of layer 1 (1024x1024) and layer 0
// (1024x256) over 256 batch rows in one launch.  How long does each take alone, and together?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wgrad_pair_probe.hip -o ab_libs/wgrad_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const int M = 256, N = 1024;
    float *dZ1, *dZ0, *X1, *X0, *W, *m, *v;
    const size_t nw = (size_t)N * 1024 + N;
    CK(hipMalloc(&dZ1, (size_t)M * N * 4)); CK(hipMalloc(&dZ0, (size_t)M * N * 4));
    CK(hipMalloc(&X1, (size_t)M * 1024 * 4)); CK(hipMalloc(&X0, (size_t)M * 256 * 4));
    CK(hipMalloc(&W, 2 * nw * 4)); CK(hipMalloc(&m, 2 * nw * 4)); CK(hipMalloc(&v, 2 * nw * 4));
    CK(hipMemset(dZ1, 0, (size_t)M * N * 4)); CK(hipMemset(dZ0, 0, (size_t)M * N * 4));
    CK(hipMemset(X1, 0, (size_t)M * 1024 * 4)); CK(hipMemset(X0, 0, (size_t)M * 256 * 4));
    CK(hipMemset(W, 0, 2 * nw * 4)); CK(hipMemset(m, 0, 2 * nw * 4)); CK(hipMemset(v, 0, 2 * nw * 4));
    AdamScalars as{5e-4f, 1.f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.001f};
    EpiGradAdam e1{W, m, v, 1024, as};
    e1.b = W + (size_t)N * 1024; e1.bm = m + (size_t)N * 1024; e1.bv = v + (size_t)N * 1024;
    EpiGradAdam e0{W + nw, m + nw, v + nw, 256, as};
    e0.b = W + nw + (size_t)N * 256; e0.bm = m + nw + (size_t)N * 256; e0.bv = v + nw + (size_t)N * 256;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto go) {
        for (int i = 0; i < 10; ++i) go();
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < 200; ++i) go();
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-64s %6.2f us\n", name, ms * 5.0f);
    };
    timeit("layer 1 alone (1024x1024, 256 workgroups)", [&]() { gemm_wgrad(dZ1, N, X1, 1024, N, 1024, M, e1, st); });
    timeit("layer 0 alone (1024x256, 64 workgroups)", [&]() { gemm_wgrad(dZ0, N, X0, 256, N, 256, M, e0, st); });
    timeit("both, two launches", [&]() { gemm_wgrad(dZ1, N, X1, 1024, N, 1024, M, e1, st); gemm_wgrad(dZ0, N, X0, 256, N, 256, M, e0, st); });
    timeit("both, one launch (layer 1 blocks first)", [&]() { gemm_wgrad_pair(dZ1, N, X1, 1024, N, 1024, e1, dZ0, N, X0, 256, N, 256, e0, M, st); });
    timeit("both, one launch (layer 0 blocks first)", [&]() { gemm_wgrad_pair(dZ0, N, X0, 256, N, 256, e0, dZ1, N, X1, 1024, N, 1024, e1, M, st); });
    return 0;
}
