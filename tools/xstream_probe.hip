// Cost of a cross-stream hand-off (event record on A, wait on B, kernel on B, record, A waits),
// the pattern of the overlapped gradient exchange.  hipcc --offload-arch=gfx950 -O2 xstream_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void work(float* p, int n, int iters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = p[i];
    for (int k = 0; k < iters; ++k) v = v * 1.0001f + 0.5f;
    p[i] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const int n = 1 << 20;
    for (int nullA = 0; nullA < 2; ++nullA)
    for (int flags : {0, (int)hipEventDisableTiming})
    for (int hand = 0; hand < 3; ++hand) {      // 0: all on A; 1: hand-off to B and back; 2: two hand-offs per iteration
        hipStream_t A = nullptr, B;
        if (!nullA) hipStreamCreateWithFlags(&A, hipStreamNonBlocking);
        hipStreamCreateWithFlags(&B, hipStreamNonBlocking);
        hipEvent_t e[4];
        for (auto& x : e) hipEventCreateWithFlags(&x, flags);
        const int iters = 300;
        double t0 = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipDeviceSynchronize();
            t0 = now();
            for (int it = 0; it < iters; ++it) {
                for (int k = 0; k < 6; ++k) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, A, d, n, 200);
                for (int h = 0; h < (hand ? hand : 1); ++h) {
                    hipStream_t S = hand ? B : A;
                    if (hand) { hipEventRecord(e[2 * h], A); hipStreamWaitEvent(S, e[2 * h], 0); }
                    hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, S, d + (1 << 21), n, 50);
                    if (hand) { hipEventRecord(e[2 * h + 1], S); hipStreamWaitEvent(A, e[2 * h + 1], 0); }
                    for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(work, dim3(n / 256), dim3(256), 0, A, d, n, 200);
                }
            }
            hipDeviceSynchronize();
        }
        printf("A=%s event flags %d hand-offs %d: %.1f us / iteration\n", nullA ? "null" : "own", flags, hand, (now() - t0) / iters * 1e6);
        if (A) hipStreamDestroy(A);
        hipStreamDestroy(B);
    }
    return 0;
}
