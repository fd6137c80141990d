// Stage-only model of a hidden-layer forward launch (256 x 1024 x 1024, 32x32 output tiles, 16 k-tiles of 64,
// production tile -> XCD mapping): every workgroup pulls its X tile and W tile (16 KB) per k-tile through REGISTERS
// into an LDS ring, one barrier per tile, no MFMAs -- with hand-counted waits (the loads are inline asm, so hipcc
// cannot drain them early).  How fast do tiles land with 4, 8 or 16 waves loading, 2 tiles ahead?
// (Production: 4 loader waves issuing LDS-DMA, 272 ns per tile with the MFMAs beside them, 218 ns of MFMA.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/stage_rate.hip -o ab_libs/stage_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f ld16(const float* p) {
    v4f r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int THREADS>
__global__ void __launch_bounds__(THREADS) stage_kernel(const float* __restrict__ X, const float* __restrict__ W, int K,
                                                        int reps, float* sink) {
    constexpr int PER = 1024 / THREADS;            // float4 per thread per 16 KB tile
    constexpr int S = 3;
    __shared__ __attribute__((aligned(16))) float lds[S * 4096];
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * 4 + loc / 8, tile_q = loc % 8;
    const int tid = threadIdx.x;
    const float* src[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int j = tid + u * THREADS;
        const int jj = j & 511, row = jj >> 4, c = jj & 15;
        src[u] = (j < 512 ? X + (size_t)(tile_q * 32 + row) * K : W + (size_t)(tile_p * 32 + row) * K) + c * 4;
    }
    const int nk = K / 64;
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        v4f r0[PER], r1[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) r0[u] = ld16(src[u]);
#pragma unroll
        for (int u = 0; u < PER; ++u) r1[u] = ld16(src[u] + 64);
        for (int t = 0; t < nk; t += 2) {
            const int ta = t + 2 < nk ? t + 2 : nk - 1, tb = t + 3 < nk ? t + 3 : nk - 1;
            wait_vm<PER>();                                                   // tile t landed, t+1 in flight
            float* s0 = lds + (t % S) * 4096;
#pragma unroll
            for (int u = 0; u < PER; ++u) *reinterpret_cast<v4f*>(s0 + (tid + u * THREADS) * 4) = r0[u];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < PER; ++u) r0[u] = ld16(src[u] + (size_t)ta * 64);
            __syncthreads();
            acc += s0[(tid * 7) & 4095];
            wait_vm<PER>();                                                   // tile t+1 landed, t+2 in flight
            float* s1 = lds + ((t + 1) % S) * 4096;
#pragma unroll
            for (int u = 0; u < PER; ++u) *reinterpret_cast<v4f*>(s1 + (tid + u * THREADS) * 4) = r1[u];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < PER; ++u) r1[u] = ld16(src[u] + (size_t)tb * 64);
            __syncthreads();
            acc += s1[(tid * 7) & 4095];
        }
        wait_vm<0>();
    }
    if (acc == 123.456f) sink[0] = acc;
}

template <int THREADS>
static void run(const float* X, const float* W, float* sink, hipEvent_t a, hipEvent_t b) {
    const int reps = 50;
    for (int i = 0; i < 2; ++i) {
        hipEventRecord(a, 0);
        hipLaunchKernelGGL((stage_kernel<THREADS>), dim3(256), dim3(THREADS), 0, 0, X, W, 1024, reps, sink);
        hipEventRecord(b, 0); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / (reps * 16.0);
    printf("  %2d waves loading, 2 tiles ahead (32 KB in flight per CU): %6.1f ns per 16 KB tile = %5.1f GB/s per CU\n", THREADS / 64, ns,
           16384.0 / ns);
}
int main() {
    float *X, *W, *sink;
    CK(hipMalloc(&X, 256 * 1024 * 4)); CK(hipMalloc(&W, 1024 * 1024 * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(X, 0, 256 * 1024 * 4)); CK(hipMemset(W, 0, 1024 * 1024 * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    run<256>(X, W, sink, a, b); run<512>(X, W, sink, a, b); run<1024>(X, W, sink, a, b);
    run<256>(X, W, sink, a, b); run<512>(X, W, sink, a, b); run<1024>(X, W, sink, a, b);
    return 0;
}
