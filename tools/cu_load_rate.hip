// How many bytes per second can ONE CU pull through its vector memory path, by bytes in flight and by where the
// data lives?  Every CU runs one workgroup of W waves; each lane keeps U independent 16-byte loads in flight
// (W x 64 x U x 16 bytes per CU), streaming a working set of `ws` bytes that the 32 workgroups of an XCD share
// (1.5 MB: stays in that XCD's L2 -- the operand panels of one layer; 48 MB: comes from the Infinity Cache).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/cu_load_rate.hip -o ab_libs/cu_load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <int U>
__global__ void __launch_bounds__(512) stream_kernel(const float4* __restrict__ src, size_t n4, int iters, float* sink) {
    const int T = blockDim.x;
    float4 acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    // workgroups of one XCD (blockIdx & 7) walk the same region, each starting at its own offset
    size_t pos = ((size_t)(blockIdx.x >> 3) * 4099 * T) % n4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            size_t i = pos + (size_t)u * T + threadIdx.x;
            if (i >= n4) i -= n4;
            const float4 v = src[i];
            acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
        }
        pos += (size_t)U * T;
        if (pos >= n4) pos -= n4;
    }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; ++u) s += acc[u].x + acc[u].y + acc[u].z + acc[u].w;
    if (s == 123.456f) sink[0] = s;
}
template <int U>
static float run(const float4* src, size_t n4, int threads, float* sink, hipEvent_t a, hipEvent_t b) {
    const int iters = (int)((size_t)8 * 1024 * 1024 / ((size_t)U * threads * 16));     // 8 MB per workgroup
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((stream_kernel<U>), dim3(256), dim3(threads), 0, 0, src, n4, iters, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)iters * U * threads * 16;
    return (float)(bytes / (ms * 1e-3) / 1e9);        // GB/s per CU
}
int main() {
    float4* buf; float* sink;
    const size_t big = (size_t)48 * 1024 * 1024;
    CK(hipMalloc(&buf, big)); CK(hipMemset(buf, 0, big)); CK(hipMalloc(&sink, 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (size_t ws : {(size_t)1536 * 1024, big}) {
        printf("working set %.1f MB per XCD-shared region (%s)\n", ws / 1048576.0, ws < (4u << 20) ? "L2-resident" : "Infinity Cache / HBM");
        for (int threads : {256, 512}) {
            const size_t n4 = ws / 16;
            printf("  %d waves per CU:", threads / 64);
            printf("  %3d KB in flight %6.1f GB/s", threads * 2 * 16 / 1024, run<2>(buf, n4, threads, sink, a, b));
            printf("  %3d KB %6.1f", threads * 4 * 16 / 1024, run<4>(buf, n4, threads, sink, a, b));
            printf("  %3d KB %6.1f", threads * 8 * 16 / 1024, run<8>(buf, n4, threads, sink, a, b));
            printf("  %3d KB %6.1f", threads * 16 * 16 / 1024, run<16>(buf, n4, threads, sink, a, b));
            printf("  %3d KB %6.1f GB/s per CU\n", threads * 32 * 16 / 1024, run<32>(buf, n4, threads, sink, a, b));
        }
    }
    return 0;
}
