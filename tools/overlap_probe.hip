// overlap_probe.hip -- what does a dependent kernel boundary cost when the consumer is launched WITHOUT
// the AQL barrier bit (hipExtAnyOrderLaunch) and waits for its producer through per-row-block arrival
// counters instead?  Models one MLP layer chain of the training step: 256 workgroups per layer in the
// production tile -> XCD mapping (tiles_q = 8 row blocks, 32 column tiles); workgroup (q, p) of layer l reads
// the whole 32 x 1024 row block q of layer l-1's output, "computes" for `work` us and writes its 32 x 32 tile
// with write-through (sc1) stores.  Every word read is checked against the value the producer must have
// written in THIS step (buffers are reused every step, so a stale L1 / L2 line shows up as a mismatch).
//
//   mode 0: plain launches (barrier bit), no counters                   -- today's schedule
//   mode 1: any-order launches + counters, consumer reads with sc1 loads (no fence)
//   mode 2: any-order launches + counters, one agent acquire fence, plain loads
//   mode 3: plain launches + counters + sc1 loads                       -- what the counters alone cost
//   mode 4: layers alternate between TWO streams (two hardware queues, no events) + counters + sc1 loads
//   mode 5: the same with one agent acquire fence and plain loads
// (hipExtAnyOrderLaunch clears the AQL barrier bit -- AMD_LOG_LEVEL=4 shows barrier=0 -- but on this part the
//  packet processor still runs the packets of ONE queue one after the other: test at the top of main.)
//
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));

struct Args {
    const float* in;       // [256][1024] output of the previous layer (null: first layer of a step)
    float* out;            // [256][1024]
    unsigned* cnt_in;      // [8] arrivals per row block of the previous layer
    unsigned* cnt_out;     // [8]
    unsigned target;       // arrivals that mean "row block complete" (32 x step count)
    int work_ticks;        // 100 MHz ticks of busy work
    int mode;
    float expect, val;
    unsigned* err;         // [0] mismatches, [1] poll time-outs
};

__device__ inline v4f load_sc1(const float* p) {
    v4f v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ inline void store_sc1(float* p, const v4f& v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__global__ void __launch_bounds__(256) layer_k(Args a) {
    const int b = blockIdx.x, xcd = b & 7, loc = b >> 3;
    const int tile_p = xcd * 4 + loc / 8, tile_q = loc % 8;
    const int tid = threadIdx.x;
    const bool flags = a.mode != 0;
    if (a.in) {
        if (flags) {
            if (tid == 0) {
                unsigned spins = 0;
                while (__hip_atomic_load(a.cnt_in + tile_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.target) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 22)) { atomicAdd(a.err + 1, 1u); break; }
                }
                if (a.mode == 2 || a.mode == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
        }
        // read the row block: 32 rows x 1024 floats = 8192 float4, 32 per thread
        const float* base = a.in + (size_t)tile_q * 32 * 1024;
        unsigned bad = 0;
        const bool sc1 = a.mode == 1 || a.mode == 3 || a.mode == 4;
        for (int i = 0; i < 32; i += 16) {
            v4f v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float* p = base + ((size_t)(i + u) * 256 + tid) * 4;
                v[u] = sc1 ? load_sc1(p) : *reinterpret_cast<const v4f*>(p);
            }
            if (sc1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 16; ++u)
                bad += (v[u][0] != a.expect) + (v[u][1] != a.expect) + (v[u][2] != a.expect) + (v[u][3] != a.expect);
        }
        if (bad) atomicAdd(a.err, bad);
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < a.work_ticks) __builtin_amdgcn_s_sleep(1);
    // own 32 x 32 tile: 256 float4
    {
        const int r = tid >> 3, c = (tid & 7) * 4;
        store_sc1(a.out + (size_t)(tile_q * 32 + r) * 1024 + tile_p * 32 + c, v4f{a.val, a.val, a.val, a.val});
    }
    if (flags) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(a.cnt_out + tile_q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void empty_k() {}
__global__ void spin_k(int ticks, unsigned long long* t) {
    const long long t0 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
}

int main(int argc, char** argv) {
    const int layers = 16, steps = argc > 1 ? atoi(argv[1]) : 300;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<float*> buf(layers);
    for (auto& p : buf) { CK(hipMalloc(&p, 256 * 1024 * 4)); CK(hipMemset(p, 0, 256 * 1024 * 4)); }
    unsigned* cnt; CK(hipMalloc(&cnt, layers * 8 * 4));
    unsigned* err; CK(hipMalloc(&err, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    // launch floor: empty 256-workgroup kernels back to back, with and without the barrier bit
    for (int any = 0; any < 2; ++any) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 2000; ++i) {
            if (any) hipExtLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch);
            else hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty 256-WG kernel, %s: %.2f us per launch\n", any ? "any-order" : "plain    ", ms * 1000 / 2000);
    }

    // does a packet without the barrier bit overlap its predecessor in the SAME queue?  A: 64 workgroups busy for
    // 30 us (a quarter of the chip), B: the same, launched any-order behind it.
    {
        unsigned long long* ts; CK(hipMalloc(&ts, 64));
        for (int any = 0; any < 2; ++any) {
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 10; ++i) {
                hipLaunchKernelGGL(spin_k, dim3(64), dim3(256), 0, st, 3000, ts);
                if (any) hipExtLaunchKernelGGL(spin_k, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 3000, ts + 2);
                else hipLaunchKernelGGL(spin_k, dim3(64), dim3(256), 0, st, 3000, ts + 2);
            }
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[4]; CK(hipMemcpy(h, ts, 32, hipMemcpyDeviceToHost));
            printf("two independent 30 us kernels on 64 CUs each, second %s: %.1f us per pair; B started %.2f us after A\n",
                   any ? "any-order" : "plain    ", ms * 1000 / 10, ((long long)h[2] - (long long)h[0]) / 100.0);
        }
    }

    hipStream_t st2;
    CK(hipStreamCreate(&st2));
    const int works[] = {0, 300, 600};
    for (int work : works) {
        for (int mode = 0; mode < 6; ++mode) {
            CK(hipMemsetAsync(cnt, 0, layers * 8 * 4, st));
            CK(hipMemsetAsync(err, 0, 8, st));
            CK(hipMemsetAsync(buf[layers - 1], 0, 256 * 1024 * 4, st));
            float total_ms = 0;
            unsigned g = 0;                               // layers launched so far: one long dependent chain over a ring of buffers
            for (int rep = 0; rep < 2; ++rep) {          // rep 0 warms up
                CK(hipStreamSynchronize(st));
                CK(hipStreamSynchronize(st2));
                CK(hipEventRecord(e0, st));
                for (int s = 0; s < steps; ++s) {
                    for (int l = 0; l < layers; ++l, ++g) {
                        const int li = (l + layers - 1) % layers;
                        Args a;
                        a.in = buf[li];
                        a.out = buf[l];
                        a.cnt_in = cnt + li * 8;
                        a.cnt_out = cnt + l * 8;
                        a.target = 32u * ((g + layers - 1) / layers);       // completed productions of buf[li]
                        a.work_ticks = work;
                        a.mode = mode;
                        a.expect = g ? (float)(g - 1) : 0.f;
                        a.val = (float)g;
                        a.err = err;
                        if (mode >= 4) hipLaunchKernelGGL(layer_k, dim3(256), dim3(256), 0, (g & 1) ? st2 : st, a);
                        else if (mode == 1 || mode == 2) hipExtLaunchKernelGGL(layer_k, dim3(256), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, a);
                        else hipLaunchKernelGGL(layer_k, dim3(256), dim3(256), 0, st, a);
                    }
                }
                CK(hipStreamSynchronize(st2));
                CK(hipEventRecord(e1, st));
                CK(hipStreamSynchronize(st));
                CK(hipEventElapsedTime(&total_ms, e0, e1));
            }
            unsigned h[2];
            CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
            printf("work %.1f us  mode %d: %.2f us per layer (%.1f us per %d-layer step)  mismatches %u  time-outs %u\n",
                   work / 100.0, mode, total_ms * 1000 / (steps * layers), total_ms * 1000 / steps, layers, h[0], h[1]);
        }
    }
    return 0;
}
