// Do LDS-DMA loads (global_load_lds_dwordx4) of ONE wave retire in issue order, as a counted `s_waitcnt vmcnt(N)` assumes?
// The wave-specialised forward kernel keeps several k-tiles in flight per loader wave and certifies the OLDEST ones with a
// partial wait (vmcnt(8) / vmcnt(4)); under memory load from other processes it produced rare wrong 32x32 tiles (round 5:
// tools/p2p_race_hunt.py).  Probe: a wave issues K LDS-DMA loads, the FIRST from a cold address (a new 4 KB page of a 2 GB
// buffer every time: HBM), the others from one hot line (L2); after `s_waitcnt vmcnt(K-1)` -- "all but the K-1 youngest have
// landed" -- it reads the first load's LDS slot.  A sentinel there = the counted wait let a read pass an unfinished older load.
// Runs alone and beside `hammer` workgroups that stream a 1 GB buffer.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/lds_dma_order_probe.hip -o ab_libs/lds_dma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int K>
__global__ void __launch_bounds__(64) probe_kernel(const float* cold, size_t cold_floats, const float* hot, int iters, int probes,
                                                   const float* stream, size_t stream_f4, unsigned* bad, float* sink) {
    __shared__ __attribute__((aligned(16))) float lds[K * 256];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= probes) {                       // hammer: stream a big buffer, nothing else
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        const v4f* s = reinterpret_cast<const v4f*>(stream);
        for (int r = 0; r < 4; ++r)
            for (size_t i = (size_t)(blockIdx.x - probes) * 64 + lane; i < stream_f4; i += (size_t)(gridDim.x - probes) * 64) acc += s[i];
        if (acc[0] == 123.f) sink[0] = acc[1];
        return;
    }
    unsigned wrong = 0;
    const size_t pages = cold_floats / 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < K; ++k) *reinterpret_cast<v4f*>(lds + k * 256 + lane * 4) = v4f{-1.f, -1.f, -1.f, -1.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const size_t page = ((size_t)blockIdx.x * 7919 + (size_t)it * 104729) % pages;
        const float* c = cold + page * 1024 + lane * 4;     // value there = its own float index (< 2^24 is exact; only != -1 matters)
        dma16(c, lds);                                        // the OLDEST load: cold
#pragma unroll
        for (int k = 1; k < K; ++k) dma16(hot + lane * 4, lds + k * 256);
        wait_vm<K - 1>();                                     // "everything but the K-1 youngest has landed"
        const v4f got = *reinterpret_cast<const v4f*>(lds + lane * 4);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        wrong += (got[0] == -1.f) + (got[3] == -1.f);
        wait_vm<0>();
    }
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    const size_t cold_floats = (size_t)512 << 20, stream_floats = (size_t)256 << 20;       // 2 GB, 1 GB
    float *cold, *hot, *stream, *sink; unsigned* bad;
    CK(hipMalloc(&cold, cold_floats * 4)); CK(hipMalloc(&hot, 4096)); CK(hipMalloc(&stream, stream_floats * 4));
    CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bad, 4));
    CK(hipMemset(cold, 0x3f, cold_floats * 4)); CK(hipMemset(hot, 0x40, 4096)); CK(hipMemset(stream, 0, stream_floats * 4));
    for (int hammer = 0; hammer <= 1; ++hammer)
        for (int kk = 0; kk < 3; ++kk) {
            CK(hipMemset(bad, 0, 4));
            const int probes = 256, iters = 4000, grid = probes + (hammer ? 1024 : 0);
            if (kk == 0) hipLaunchKernelGGL(probe_kernel<2>, dim3(grid), dim3(64), 0, 0, cold, cold_floats, hot, iters, probes, stream, stream_floats / 4, bad, sink);
            if (kk == 1) hipLaunchKernelGGL(probe_kernel<5>, dim3(grid), dim3(64), 0, 0, cold, cold_floats, hot, iters, probes, stream, stream_floats / 4, bad, sink);
            if (kk == 2) hipLaunchKernelGGL(probe_kernel<13>, dim3(grid), dim3(64), 0, 0, cold, cold_floats, hot, iters, probes, stream, stream_floats / 4, bad, sink);
            CK(hipDeviceSynchronize());
            unsigned h;
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("%s, %2d LDS-DMA loads in flight, oldest cold / others hot, read behind vmcnt(K-1): %u stale reads of %d\n",
                   hammer ? "beside 1024 streaming workgroups" : "alone", kk == 0 ? 2 : kk == 1 ? 5 : 13, h, 2 * probes * iters);
        }
    return 0;
}
