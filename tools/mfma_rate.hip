// Issue rate of v_mfma_f32_16x16x4_f32, the clock the chip sustains under it, and how the waves
// of a 512-thread workgroup share the four SIMDs (which waves may run MFMAs side by side).
// One workgroup per CU (LDS footprint), 4 independent accumulators per wave (the production
// kernels' dependency distance), N MFMAs back to back in the waves selected by `mask`.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_rate.hip -o tools/mfma_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int THREADS>
__global__ void __launch_bounds__(THREADS) mfma_chain(float* out, unsigned long long* t, int n, float a, float b, int mask) {
    __shared__ float pad[36 * 1024];          // 144 KB: one workgroup per CU
    pad[threadIdx.x] = a;
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0, 0, 0, 0};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    unsigned long long c0 = 0, c1 = 0, w0 = 0, w1 = 0;
    if ((mask >> wave) & 1) {
        c0 = __builtin_readcyclecounter();
        w0 = wall_clock64();
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % 4], 0, 0, 0);
        }
        c1 = __builtin_readcyclecounter();
        w1 = wall_clock64();
    }
    __syncthreads();
    float s = pad[threadIdx.x];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && ((mask >> wave) & 1)) { t[(blockIdx.x * 8 + wave) * 2] = c1 - c0; t[(blockIdx.x * 8 + wave) * 2 + 1] = w1 - w0; }
}

int main() {
    float* out; unsigned long long* t;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&t, 256 * 8 * 16));
    const int n = 4096;
    struct Cfg { int threads, mask; const char* what; };
    const Cfg cfgs[] = {{256, 0x0f, "256 threads, all 4 waves"},
                        {512, 0x0f, "512 threads, waves 0-3 (4-7 idle at the barrier)"},
                        {512, 0xf0, "512 threads, waves 4-7 (0-3 idle)"},
                        {512, 0x55, "512 threads, waves 0,2,4,6"},
                        {512, 0x33, "512 threads, waves 0,1,4,5"},
                        {512, 0xff, "512 threads, all 8 waves"}};
    for (const Cfg& c : cfgs) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(t, 0, 256 * 8 * 16));
            if (c.threads == 256) hipLaunchKernelGGL(mfma_chain<256>, dim3(256), dim3(256), 0, 0, out, t, n, 1.0f, 2.0f, c.mask);
            else hipLaunchKernelGGL(mfma_chain<512>, dim3(256), dim3(512), 0, 0, out, t, n, 1.0f, 2.0f, c.mask);
            CK(hipDeviceSynchronize());
        }
        unsigned long long h[16];
        CK(hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost));       // workgroup 0
        const double mf = 16.0 * n;
        printf("%-52s cycles/MFMA per wave:", c.what);
        double ns = 0; int cnt = 0;
        for (int w = 0; w < 8; ++w)
            if ((c.mask >> w) & 1) { printf(" w%d %.1f", w, h[w * 2] / mf); ns += h[w * 2 + 1] * 10.0 / mf; ++cnt; }
        printf("   (%.2f ns/MFMA)\n", ns / cnt);
    }
    return 0;
}
