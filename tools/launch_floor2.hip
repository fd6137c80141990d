// launch_floor2.hip -- what does a dependent launch of an EMPTY kernel cost as a function of what the production
// launches carry: kernel-argument bytes (by-value structs of ~200 B), workgroup size (256 / 512 threads), static LDS
// (0 / 64 KB), number of workgroups (256 / 800), and where the runtime puts the arguments (HIP_FORCE_DEV_KERNARG=0/1,
// set in the environment by the caller).
// build: hipcc --offload-arch=gfx950 -O3 -o build/launch_floor2 tools/launch_floor2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float* p[16]; int v[24]; };          // 224 B, like GemmArgs + an epilogue
__global__ void k0() {}
__global__ void kbig(Big b) { if (b.v[0] == 12345) b.p[0][0] = 1.f; }
__global__ void __launch_bounds__(512) k512(Big b) { if (b.v[0] == 12345) b.p[0][0] = 1.f; }
__global__ void __launch_bounds__(512) klds(Big b) {
    __shared__ float lds[16384];                   // 64 KB
    if (b.v[0] == 12345) { lds[threadIdx.x] = 1.f; __syncthreads(); b.p[0][0] = lds[0]; }
}
__global__ void __launch_bounds__(256) klds256(Big b) {
    __shared__ float lds[12288];                   // 48 KB
    if (b.v[0] == 12345) { lds[threadIdx.x] = 1.f; __syncthreads(); b.p[0][0] = lds[0]; }
}
// busy for `ticks` of the 100 MHz wall clock, so that the host runs ahead and the period is the device's
__global__ void spin0() { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < 500) __builtin_amdgcn_s_sleep(2); }
__global__ void spinbig(Big b) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < b.v[1]) __builtin_amdgcn_s_sleep(2);
    if (b.v[0] == 12345) b.p[0][0] = 1.f;
}
// the same with the arguments behind ONE pointer to a block that stays where it is (device memory, warm in L2)
__global__ void spinptr(const Big* pb) {
    const long long t0 = wall_clock64();
    const int ticks = pb->v[1];
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
    if (pb->v[0] == 12345) pb->p[0][0] = 1.f;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float* buf; CK(hipMalloc(&buf, 1 << 20));
    Big big{}; big.p[0] = buf;
    const int N = 4000;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 100; ++i) launch();
        hipStreamSynchronize(s);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(a, s);
            for (int i = 0; i < N; ++i) launch();
            hipEventRecord(b, s);
            hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("%-64s %6.2f us/launch\n", name, best * 1e3 / N);
    };
    time("no arguments, 256 WG x 256 threads", [&] { hipLaunchKernelGGL(k0, dim3(256), dim3(256), 0, s); });
    time("224 B of arguments, 256 WG x 256 threads", [&] { hipLaunchKernelGGL(kbig, dim3(256), dim3(256), 0, s, big); });
    time("224 B of arguments, 256 WG x 512 threads", [&] { hipLaunchKernelGGL(k512, dim3(256), dim3(512), 0, s, big); });
    time("224 B of arguments, 256 WG x 512 threads, 64 KB LDS", [&] { hipLaunchKernelGGL(klds, dim3(256), dim3(512), 0, s, big); });
    time("224 B of arguments, 800 WG x 256 threads, 48 KB LDS", [&] { hipLaunchKernelGGL(klds256, dim3(800), dim3(256), 0, s, big); });
    time("224 B of arguments, 128 WG x 256 threads", [&] { hipLaunchKernelGGL(kbig, dim3(128), dim3(256), 0, s, big); });
    // device-bound periods: every kernel busy for 5.00 us, the host runs ahead
    big.v[1] = 500;
    Big* dbig; CK(hipMalloc(&dbig, sizeof(Big))); CK(hipMemcpy(dbig, &big, sizeof(Big), hipMemcpyHostToDevice));
    time("5 us of work, no arguments, 256 WG x 256", [&] { hipLaunchKernelGGL(spin0, dim3(256), dim3(256), 0, s); });
    time("5 us of work, 224 B of arguments (the wait length among them)", [&] { hipLaunchKernelGGL(spinbig, dim3(256), dim3(256), 0, s, big); });
    time("5 us of work, one pointer to a resident argument block", [&] { hipLaunchKernelGGL(spinptr, dim3(256), dim3(256), 0, s, (const Big*)dbig); });
    return 0;
}
