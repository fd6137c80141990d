// The step's trailing launch in the world phase: weight gradient of the stack's first layer (1024 x 256, K = 256
// batch rows, Adam fused in its epilogue) + the 256 workgroups that apply the deferred Adam update of layer 1
// (1024 x 1024: 28 MB of p, g, m, v).  What does each part cost alone and together?  Random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/trailing_lab.hip -o ab_libs/trailing_lab
#include <hip/hip_runtime.h>
#include <cstdio>
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
    hipStream_t st; CK(hipStreamCreate(&st));
    const int M = 256, N = 1024, K0 = 256;
    const size_t n1 = (size_t)N * 1024 + N, n0 = (size_t)N * K0 + N;
    float *dZ0, *X0, *W0, *m0, *v0, *G0, *p1, *g1, *m1, *v1;
    CK(hipMalloc(&dZ0, (size_t)M * N * 4)); CK(hipMalloc(&X0, (size_t)M * K0 * 4));
    CK(hipMalloc(&W0, n0 * 4)); CK(hipMalloc(&m0, n0 * 4)); CK(hipMalloc(&v0, n0 * 4)); CK(hipMalloc(&G0, n0 * 4));
    CK(hipMalloc(&p1, n1 * 4)); CK(hipMalloc(&g1, n1 * 4)); CK(hipMalloc(&m1, n1 * 4)); CK(hipMalloc(&v1, n1 * 4));
    float* bufs[] = {dZ0, X0, W0, m0, v0, p1, g1, m1, v1};
    const size_t lens[] = {(size_t)M * N, (size_t)M * K0, n0, n0, n0, n1, n1, n1, n1};
    for (int i = 0; i < 9; ++i) hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, bufs[i], lens[i], 31u * (i + 1));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, st, v0, n0, 7u);
    CK(hipStreamSynchronize(st));
    AdamScalars as{5e-6f, 1.f, 0.9f, 0.999f, 1e-8f, 0.1f, 0.001f};
    EpiGradAdam ea{W0, m0, v0, K0, as};
    ea.b = W0 + (size_t)N * K0; ea.bm = m0 + (size_t)N * K0; ea.bv = v0 + (size_t)N * K0;
    EpiGradStore es{G0, K0};
    es.gb = G0 + (size_t)N * K0;
    AdamSeg seg;
    seg.p = p1; seg.g = g1; seg.m = m1; seg.v = v1; seg.n4 = (long long)(n1 / 4); seg.s = as;
    const AdamPair ad(seg), none;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char* name, auto go) {
        for (int i = 0; i < 20; ++i) go();
        hipStreamSynchronize(st);
        hipEventRecord(a, st);
        for (int i = 0; i < 300; ++i) go();
        hipEventRecord(b, st); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("  %-72s %6.2f us\n", name, ms * 1e3f / 300);
    };
    timeit("first-layer weight gradient, gradient stored", [&]() { gemm_wgrad(dZ0, N, X0, K0, N, K0, M, es, st, &none); });
    timeit("first-layer weight gradient, Adam in its epilogue", [&]() { gemm_wgrad(dZ0, N, X0, K0, N, K0, M, ea, st, &none); });
    timeit("the same + 256 deferred-Adam workgroups for layer 1 (= the trailing launch)", [&]() { gemm_wgrad(dZ0, N, X0, K0, N, K0, M, ea, st, &ad); });
    timeit("gradient stored + 256 deferred-Adam workgroups for layer 1", [&]() { gemm_wgrad(dZ0, N, X0, K0, N, K0, M, es, st, &ad); });
    return 0;
}
