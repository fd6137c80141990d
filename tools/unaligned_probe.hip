// Can the layer-0 loaders read rows of the demonstration set where they lie?  A state row is dim_body floats (197 at the
// BASELINE dims): row bases are 4-byte aligned only.  Two questions, answered on the hardware:
//   (a) global_load_dwordx4 into VGPRs from a 4-byte-aligned address: right values?  at what rate against aligned?
//   (b) global_load_lds_dwordx4 (LDS-DMA, 16 bytes per lane) from a 4-byte-aligned address: right values?  rate?
// Pattern = what a [32 rows][64 k] tile load does: lane j of a 512-lane group reads 16 bytes at row (j >> 4), chunk
// (j & 15) of a row-indirect source with row stride `ld` floats (197: unaligned rows; 256: aligned rows).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/unaligned_probe.hip -o /tmp/unaligned_probe && /tmp/unaligned_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4f ld16(const float* p) {
    v4f r;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    return r;
}
__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}

// correctness: out[row][k] for 32 rows x 64 k of tile (row0, k0), rows picked through idx[]
__global__ void __launch_bounds__(512) check_kernel(const float* src, const int* idx, int ld, int k0, float* out_reg, float* out_dma) {
    __shared__ __attribute__((aligned(16))) float lds[32 * 64];
    const int j = threadIdx.x, row = j >> 4, c = j & 15;
    const float* p = src + (size_t)idx[row] * ld + k0 + c * 4;
    const v4f r = ld16(p);
    *reinterpret_cast<v4f*>(out_reg + j * 4) = r;
    const int wave = __builtin_amdgcn_readfirstlane(j >> 6);
    dma16(p, lds + wave * 256);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    *reinterpret_cast<v4f*>(out_dma + j * 4) = *reinterpret_cast<const v4f*>(lds + j * 4);
}

// rate: every workgroup walks `nk` k-tiles of its 32 gathered rows, MODE 0 = registers, 1 = LDS-DMA
template <int MODE>
__global__ void __launch_bounds__(512) rate_kernel(const float* src, const int* idx, int ld, int nk, int reps, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 32 * 64];
    const int j = threadIdx.x, row = j >> 4, c = j & 15;
    const int wave = __builtin_amdgcn_readfirstlane(j >> 6);
    const float* p = src + (size_t)idx[blockIdx.x * 32 + row] * ld + c * 4;
    float acc = 0.f;
    for (int rep = 0; rep < reps; ++rep) {
        for (int t = 0; t < nk; ++t) {
            if (MODE == 0) {
                v4f r;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p + t * 64) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc += r[0] + r[3];
            } else {
                dma16(p + t * 64, lds + (t & 3) * 2048 + wave * 256);
            }
        }
        if (MODE == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); acc += lds[j]; }
    }
    if (acc == 123.456f) sink[0] = acc;
}

int main() {
    const int R = 20000, LDS[2] = {197, 256};
    for (int li = 0; li < 2; ++li) {
        const int ld = LDS[li];
        std::vector<float> h((size_t)R * ld + 64);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(i % 100003) * 0.5f;
        std::vector<int> hi(8192);
        for (int i = 0; i < 8192; ++i) hi[i] = (i * 7919 + 13) % (R - 2);
        float *d, *o1, *o2, *sink; int* di;
        CK(hipMalloc(&d, h.size() * 4)); CK(hipMalloc(&o1, 2048 * 4)); CK(hipMalloc(&o2, 2048 * 4)); CK(hipMalloc(&sink, 4));
        CK(hipMalloc(&di, hi.size() * 4));
        CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
        int bad_reg = 0, bad_dma = 0;
        for (int k0 = 0; k0 < 2 * ld - 64; k0 += 64) {
            hipLaunchKernelGGL(check_kernel, dim3(1), dim3(512), 0, 0, d, di, ld, k0, o1, o2);
            CK(hipDeviceSynchronize());
            std::vector<float> a(2048), b(2048);
            CK(hipMemcpy(a.data(), o1, 2048 * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), o2, 2048 * 4, hipMemcpyDeviceToHost));
            for (int j = 0; j < 512; ++j)
                for (int e = 0; e < 4; ++e) {
                    const float want = h[(size_t)hi[j >> 4] * ld + k0 + (j & 15) * 4 + e];
                    bad_reg += a[j * 4 + e] != want;
                    bad_dma += b[j * 4 + e] != want;
                }
        }
        printf("row stride %d floats (%s): 16-byte loads to registers: %d wrong values; LDS-DMA 16 bytes per lane: %d wrong values\n",
               ld, ld % 4 ? "rows 4-byte aligned" : "rows 16-byte aligned", bad_reg, bad_dma);
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int nk = 2 * ld / 64, reps = 200, grid = 256;
        for (int mode = 0; mode < 2; ++mode) {
            for (int it = 0; it < 2; ++it) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(512), 0, 0, d, di, ld, nk, reps, sink);
                else hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(512), 0, 0, d, di, ld, nk, reps, sink);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double tiles = (double)grid * nk * reps;
            printf("  %s: %.1f ns per 8 KB tile per workgroup (one per CU), %.2f TB/s over the chip\n",
                   mode == 0 ? "registers (each load waited for)" : "LDS-DMA (4-slot ring, drained per pass)",
                   ms * 1e6 / (nk * reps), tiles * 8192 / (ms * 1e-3) / 1e12);
        }
        hipFree(d); hipFree(o1); hipFree(o2); hipFree(sink); hipFree(di);
    }
    return 0;
}
