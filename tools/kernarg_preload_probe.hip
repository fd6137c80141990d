// kernarg_preload_probe.hip -- is the kernel-argument fetch on the critical path of a dependent launch, and does
// gfx950's kernarg preload (the first user SGPRs filled by the hardware at wave launch instead of by an s_load in the
// kernel; hipcc -mllvm -amdgpu-kernarg-preload-count=N) take it off?
// A chain of dependent launches; every workgroup needs its pointer arguments before it can issue its first load:
//   out[i] = in[i] + 1   (256 workgroups x 256 threads, one float4 each; buffers alternate)
// Period per launch, device-bound (the host runs ahead), for this translation unit compiled with and without the flag.
// build: hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=8] -o build/kpp_{off,on} tools/kernarg_preload_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) step_k(const v4f* in, v4f* out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        v4f v = in[i];
        if (v.x < 0.f) return;                             // (the load has landed)
        const long long t0 = wall_clock64();               // 4 us of "work", so that the host runs ahead of the device
        while (wall_clock64() - t0 < 400) __builtin_amdgcn_s_sleep(1);
        v += 1.f; out[i] = v;
    }
}
// the same behind a by-value struct (what the production kernels take)
struct Args { const v4f* in; v4f* out; int n; int pad[13]; };
__global__ void __launch_bounds__(256) step_struct_k(Args a) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.n) {
        v4f v = a.in[i];
        if (v.x < 0.f) return;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < 400) __builtin_amdgcn_s_sleep(1);
        v += 1.f; a.out[i] = v;
    }
}
__global__ void empty_k() { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < 400) __builtin_amdgcn_s_sleep(1); }
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    v4f *x, *y; CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&y, 1 << 20)); CK(hipMemset(x, 0, 1 << 20)); CK(hipMemset(y, 0, 1 << 20));
    const int N = 4000, n = 65536;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 100; ++i) launch(i);
        (void)hipStreamSynchronize(s);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(a, s);
            for (int i = 0; i < N; ++i) launch(i);
            (void)hipEventRecord(b, s);
            (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-56s %6.2f us/launch\n", name, best * 1e3 / N);
    };
    time("4 us of work, no arguments, no memory access", [&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s); });
    time("load, 4 us of work, store; scalar / pointer arguments", [&](int i) { hipLaunchKernelGGL(step_k, dim3(256), dim3(256), 0, s, (i & 1) ? y : x, (i & 1) ? x : y, n); });
    time("load, 4 us of work, store; one 80-byte struct by value", [&](int i) {
        Args g{}; g.in = (i & 1) ? y : x; g.out = (i & 1) ? x : y; g.n = n;
        hipLaunchKernelGGL(step_struct_k, dim3(256), dim3(256), 0, s, g); });
    float h[4]; CK(hipMemcpy(h, x, 16, hipMemcpyDeviceToHost));
    printf("check: x[0] = %.0f\n", h[0]);
    return 0;
}
