// Would an error-compensated bf16 path ever pay on the hidden-layer launches?  (a PROBE, timing only.)
// The standing reading of the forward kernel's k-loop: 337 ns per 16 KB k-tile -- the LDS-DMA stream slowed by the concurrent
// fp32-MFMA issue (171 ns without MFMAs, 213 ns of MFMA alone).  If the matrix pipe were busy for less time per tile, would
// the loop get shorter?  This probe runs the production loop's SHAPE -- one 512-thread workgroup per CU, four loader waves
// landing a [32][64] Q tile + a [32][64] P tile per k-tile by LDS-DMA into a 4-slot ring, four compute waves that split K,
// one barrier per tile, the production fragment reads (4 x ds_read_b128 per tile and wave) -- with four matrix-pipe loads:
//   fp32      16 x v_mfma_f32_16x16x4_f32 per tile and wave                                  (what runs)
//   split     the VALU split of both operands' fragments into three bf16 terms each (x = hi + mid + lo) and the six
//             products that an fp32-accurate result needs (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid):
//             24 x v_mfma_f32_16x16x16_bf16 per tile and wave -- a lane's four floats of a fragment are exactly the four
//             bf16 values per lane that instruction takes, so the split is lane-local
//   presplit  the 24 bf16 MFMAs alone (operands split by their producers: an upper bound; the 1.5x fragment bytes such a
//             layout would add to the DMA stream are NOT modelled)
//   none      no matrix work (the DMA stream + barriers + fragment reads alone)
// Results are garbage by construction; only the time per k-tile means anything.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/bf16_split_probe.hip -o ab_libs/bf16_split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// x = hi + mid + lo, three bf16 terms (truncating split: every term exact, the residual below 2^-24 |x|)
__device__ __forceinline__ void split3(const v4f& x, v4s& hi, v4s& mid, v4s& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned b0 = __float_as_uint(x[e]) & 0xffff0000u;
        const float r1 = x[e] - __uint_as_float(b0);
        const unsigned b1 = __float_as_uint(r1) & 0xffff0000u;
        const float r2 = r1 - __uint_as_float(b1);
        hi[e] = (short)(b0 >> 16); mid[e] = (short)(b1 >> 16); lo[e] = (short)(__float_as_uint(r2) >> 16);
    }
}

template <int MODE>
__global__ void __launch_bounds__(512) loop_kernel(const float* __restrict__ X, const float* __restrict__ W, int K, int reps, float* sink) {
    constexpr int S = 4, kTile = 32 * 64, kStage = 2 * kTile;
    __shared__ __attribute__((aligned(16))) float lds[S * kStage];
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * 4 + loc / 8, tile_q = loc % 8;           // 256 x 1024 outputs: 8 x 32 tiles, production mapping
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4, nk = K / 64;
    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int rep = 0; rep < reps; ++rep) {
        if (wave >= 4) {                                               // loaders: 2 Q + 2 P instructions per tile and wave
            const int u0 = wave - 4;
            const float* sq[2]; const float* sp[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = (u0 + 4 * u) * 64 + lane, row = j >> 4, c = (j & 15) ^ (row & 15);
                sq[u] = X + (size_t)(tile_q * 32 + row) * K + c * 4;
                sp[u] = W + (size_t)(tile_p * 32 + row) * K + c * 4;
            }
            auto issue = [&](int t) {
                float* slot = lds + (t % S) * kStage;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dma16(sq[u] + t * 64, slot + (u0 + 4 * u) * 256);
                    dma16(sp[u] + t * 64, slot + kTile + (u0 + 4 * u) * 256);
                }
            };
            issue(0); issue(1); issue(2);
            for (int t = 0; t < nk; ++t) {
                const int y = nk - 1 - t < 2 ? nk - 1 - t : 2;         // tiles younger than t still in flight
                if (y >= 2) wait_vm<8>(); else if (y == 1) wait_vm<4>(); else wait_vm<0>();
                __builtin_amdgcn_s_barrier();                          // tile t landed; tile t-1's slot is free
                asm volatile("" ::: "memory");
                if (t + 3 < nk) issue(t + 3);
            }
        } else {
            int oq[2];
#pragma unroll
            for (int a = 0; a < 2; ++a) { const int row = 16 * a + li; oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2); }
            for (int t = 0; t < nk; ++t) {
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const float* st = lds + (t % S) * kStage;
                v4f fq[2], fp[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) { fq[a] = *reinterpret_cast<const v4f*>(st + oq[a]); fp[a] = *reinterpret_cast<const v4f*>(st + kTile + oq[a]); }
                if (MODE == 0) {
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[b][s2], fq[a][s2], acc[a][b], 0, 0, 0);
                } else if (MODE == 1 || MODE == 2) {
                    v4s qh[2], qm[2], ql[2], ph[2], pm[2], pl[2];
                    if (MODE == 1) {
#pragma unroll
                        for (int a = 0; a < 2; ++a) { split3(fq[a], qh[a], qm[a], ql[a]); split3(fp[a], ph[a], pm[a], pl[a]); }
                    } else {                                           // operands arrive split: reinterpret what was read
#pragma unroll
                        for (int a = 0; a < 2; ++a) {
                            qh[a] = __builtin_bit_cast(v4s, (__attribute__((ext_vector_type(2))) float){fq[a][0], fq[a][1]});
                            qm[a] = __builtin_bit_cast(v4s, (__attribute__((ext_vector_type(2))) float){fq[a][2], fq[a][3]});
                            ql[a] = qh[a];
                            ph[a] = __builtin_bit_cast(v4s, (__attribute__((ext_vector_type(2))) float){fp[a][0], fp[a][1]});
                            pm[a] = __builtin_bit_cast(v4s, (__attribute__((ext_vector_type(2))) float){fp[a][2], fp[a][3]});
                            pl[a] = ph[a];
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pl[b], qh[a], acc[a][b], 0, 0, 0);   // smallest terms first
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ph[b], ql[a], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pm[b], qm[a], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pm[b], qh[a], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ph[b], qm[a], acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ph[b], qh[a], acc[a][b], 0, 0, 0);
                        }
                } else {
                    asm volatile("" ::"v"(fq[0]), "v"(fq[1]), "v"(fp[0]), "v"(fp[1]));
                }
            }
        }
        __syncthreads();
    }
    if (wave < 4) {
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) s += acc[a][b][0] + acc[a][b][3];
        if (s == 123.456f) sink[0] = s;
    }
}

int main() {
    const int M = 256, N = 1024, K = 1024, reps = 50;
    float *X, *W, *sink;
    CK(hipMalloc(&X, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(X, 0x3c, (size_t)M * K * 4)); CK(hipMemset(W, 0x3b, (size_t)N * K * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[4] = {"fp32: 16 x v_mfma_f32_16x16x4_f32", "split: VALU 3-way split + 24 x v_mfma_f32_16x16x16_bf16",
                            "presplit: 24 x v_mfma_f32_16x16x16_bf16 only", "none: DMA stream + barriers + fragment reads"};
    for (int pass = 0; pass < 2; ++pass)
        for (int mode = 0; mode < 4; ++mode) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(loop_kernel<0>, dim3(256), dim3(512), 0, 0, X, W, K, reps, sink);
            if (mode == 1) hipLaunchKernelGGL(loop_kernel<1>, dim3(256), dim3(512), 0, 0, X, W, K, reps, sink);
            if (mode == 2) hipLaunchKernelGGL(loop_kernel<2>, dim3(256), dim3(512), 0, 0, X, W, K, reps, sink);
            if (mode == 3) hipLaunchKernelGGL(loop_kernel<3>, dim3(256), dim3(512), 0, 0, X, W, K, reps, sink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (pass) printf("%-62s %6.1f ns per k-tile (256 x 1024 x 1024: %5.2f us per layer pass)\n", names[mode],
                             ms * 1e6 / (reps * (K / 64)), ms * 1e3 / reps);
        }
    return 0;
}
