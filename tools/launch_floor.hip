// Launch-floor microbenchmark: how long does a dependent kernel boundary cost on this box?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void empty_k() {}
__global__ void touch_k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
__global__ void spin_k(long long cycles, float* p) {
    long long t0 = clock64();
    while (clock64() - t0 < cycles) {}
    if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] = 1.f;
}
int main() {
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float* buf; CK(hipMalloc(&buf, 64 << 20)); CK(hipMemset(buf, 0, 64 << 20));
    const int N = 2000;
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 50; ++i) launch(i);
        hipStreamSynchronize(s); hipStreamSynchronize(s2);
        hipEventRecord(a, s);
        for (int i = 0; i < N; ++i) launch(i);
        hipEventRecord(b, s);
        hipEventSynchronize(b); hipStreamSynchronize(s2);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-48s %8.2f us/launch\n", name, ms * 1e3 / N);
    };
    time("empty <<<1,64>>>", [&](int) { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s); });
    time("empty <<<256,256>>>", [&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, s); });
    time("empty <<<2048,256>>>", [&](int) { hipLaunchKernelGGL(empty_k, dim3(2048), dim3(256), 0, s); });
    time("touch 256 KB <<<256,256>>>", [&](int) { hipLaunchKernelGGL(touch_k, dim3(256), dim3(256), 0, s, buf, 65536); });
    time("touch 12 MB <<<12288,256>>>", [&](int) { hipLaunchKernelGGL(touch_k, dim3(12288), dim3(256), 0, s, buf, 3 << 20); });
    time("spin 2000 cyc <<<256,256>>>", [&](int) { hipLaunchKernelGGL(spin_k, dim3(256), dim3(256), 0, s, 2000ll, buf); });
    time("spin 10000 cyc <<<256,256>>>", [&](int) { hipLaunchKernelGGL(spin_k, dim3(256), dim3(256), 0, s, 10000ll, buf); });
    time("spin 10000 cyc alternating 2 streams", [&](int i) { hipLaunchKernelGGL(spin_k, dim3(128), dim3(256), 0, (i & 1) ? s2 : s, 10000ll, buf); });
    // graph of 16 empty kernels
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 16; ++i) hipLaunchKernelGGL(touch_k, dim3(256), dim3(256), 0, s, buf, 65536);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < 200; ++i) hipGraphLaunch(ge, s);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-48s %8.2f us/kernel\n", "graph of 16 touch kernels", ms * 1e3 / 200 / 16);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("clockRate %d kHz, CUs %d, name %s\n", clk, pr.multiProcessorCount, pr.name);
    return 0;
}
