// Do the fp32 matrix pipe and the fp32 vector pipe of a SIMD run side by side?
// (Both peak at 256 flop/clk/CU = 157.3 TFLOP/s on MI355X; a weight gradient fed from scalar registers could
// in principle run on the VALU while the input gradient holds the matrix pipe.)
// One 768-thread workgroup per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x4_f32 on 4 independent
// accumulators, waves 4-11 (two per SIMD) issue v_fma_f32 with one scalar operand on 32 independent
// accumulators -- registers only, random operands.  Timed alone and together: shader cycles (s_memtime) and
// 100 MHz wall clock per wave of workgroup 0, kernel wall time by events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dual_pipe.hip -o ab_libs/dual_pipe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Scal { float b[32]; };

template <bool PLAIN>
__global__ void __launch_bounds__(768) dual_kernel(const float* __restrict__ src, float* out, unsigned long long* t,
                                                   int n_mfma, int n_fma, int mask, Scal sc) {
    __shared__ float pad[36 * 1024];          // 144 KB: one workgroup per CU
    pad[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    unsigned long long c0 = 0, c1 = 0, w0 = 0, w1 = 0;
    float s = 0.f;
    if (wave < 4) {
        if (mask & 1) {
            float a[4], b[4];
            for (int u = 0; u < 4; ++u) { a[u] = src[(threadIdx.x + 977 * u + 64 * blockIdx.x) & 65535]; b[u] = src[(threadIdx.x * 3 + 131 * u) & 65535]; }
            v4f acc[4];
            for (int i = 0; i < 4; ++i) acc[i] = v4f{0, 0, 0, 0};
            c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j % 4], b[(j / 4) % 4], acc[j % 4], 0, 0, 0);
            }
            c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
            for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        }
    } else if (mask & 2) {
        float a[4];
        for (int u = 0; u < 4; ++u) a[u] = src[(threadIdx.x + 4099 * u + 64 * blockIdx.x) & 65535];
        float acc[32];
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j] = 0.f;
        c0 = __builtin_readcyclecounter(); w0 = wall_clock64();
        for (int i = 0; i < n_fma; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    if (PLAIN) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a[u]), "s"(sc.b[j]));   // un-packed
                    else acc[j] = __builtin_fmaf(a[u], sc.b[j], acc[j]);     // hipcc packs these: v_pk_fma_f32 v[..], v[..], s[..]
                }
        }
        c1 = __builtin_readcyclecounter(); w1 = wall_clock64();
#pragma unroll
        for (int j = 0; j < 32; ++j) s += acc[j];
    }
    __syncthreads();
    out[blockIdx.x * 768 + threadIdx.x] = s + pad[threadIdx.x];
    if ((threadIdx.x & 63) == 0) { t[(blockIdx.x * 12 + wave) * 2] = c1 - c0; t[(blockIdx.x * 12 + wave) * 2 + 1] = w1 - w0; }
}

int main() {
    float *out, *src; unsigned long long* t;
    CK(hipMalloc(&out, 256 * 768 * 4)); CK(hipMalloc(&t, 256 * 12 * 16)); CK(hipMalloc(&src, 65536 * 4));
    {
        static float h[65536];
        unsigned x = 12345;
        for (int i = 0; i < 65536; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) % 20001 - 10000) * 1e-4f; }
        CK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
    }
    Scal sc;
    for (int j = 0; j < 32; ++j) sc.b[j] = 0.37f + 0.011f * j;
    const int n_mfma = 8192;                 // x16 MFMAs per wave = 131072 MFMAs x 32 cycles = 4.2 M cycles
    const int n_fma = 2048;                  // x128 FMAs per wave  = 262144 FMAs
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[8] = {"", "matrix pipe alone (4 waves / CU)", "vector pipe alone, v_pk_fma_f32 (8 waves / CU)", "both (v_pk_fma_f32)",
                            "", "", "vector pipe alone, v_fma_f32", "both (v_fma_f32)"};
    for (int mask : {1, 2, 3, 6, 7}) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(t, 0, 256 * 12 * 16));
            CK(hipEventRecord(e0, 0));
            if (mask & 4) hipLaunchKernelGGL(dual_kernel<true>, dim3(256), dim3(768), 0, 0, src, out, t, n_mfma, n_fma, mask, sc);
            else hipLaunchKernelGGL(dual_kernel<false>, dim3(256), dim3(768), 0, 0, src, out, t, n_mfma, n_fma, mask, sc);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned long long h[24];
        CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));
        const double fl_m = (mask & 1) ? 256.0 * 4 * 16.0 * n_mfma * 2048.0 : 0, fl_v = (mask & 2) ? 256.0 * 8 * 128.0 * n_fma * 128.0 : 0;
        printf("%-36s kernel %.1f us -> %.1f TFLOP/s (matrix %.1f + vector %.1f)\n", names[mask], ms * 1e3,
               (fl_m + fl_v) / (ms * 1e-3) / 1e12, fl_m / (ms * 1e-3) / 1e12, fl_v / (ms * 1e-3) / 1e12);
        for (int w = 0; w < 12; ++w) {
            if (!h[2 * w + 1]) continue;
            const double us = h[2 * w + 1] / 100.0;
            const double per = w < 4 ? (double)h[2 * w] / (16.0 * n_mfma) : (double)h[2 * w] / (128.0 * n_fma);
            printf("    wave %2d (%s): %.0f us, clock %.3f GHz, %.2f cycles per %s\n", w, w < 4 ? "mfma" : "fma ", us,
                   h[2 * w] / (us * 1e3), per, w < 4 ? "MFMA" : "v_fma_f32");
        }
    }
    return 0;
}
