// tools/gemm_dma_ring.h -- the LDS-DMA ring variant of the contraction kernels (measurement only).
// global_load_lds_dwordx4 into an 8-slot LDS ring, 7 k-tiles in flight per CU behind hand-counted
// `s_waitcnt vmcnt(N)`, one raw s_barrier per k-tile.  Correct (it passed the full GPU parity suite
// in round 1) but ~12 % slower than the register-staged kernels of physicsvae_amd/csrc/pvae_gemm.h:
// see DESIGN.md section 4 and tools/gemm_ablate.hip.  Include AFTER pvae_gemm.h.
#pragma once
namespace pvae {

template <int STAGES, int G>
__device__ inline void wait_tile_landed(int younger_in_flight) {
    // tile t of this wave has landed once at most `younger_in_flight` tiles (G DMA instructions
    // each) issued after it are still outstanding
    switch (younger_in_flight) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<(STAGES > 2 ? 1 : 0) * G>(); break;
        case 2: wait_vmcnt<(STAGES > 3 ? 2 : 0) * G>(); break;
        case 3: wait_vmcnt<(STAGES > 4 ? 3 : 0) * G>(); break;
        case 4: wait_vmcnt<(STAGES > 5 ? 4 : 0) * G>(); break;
        case 5: wait_vmcnt<(STAGES > 6 ? 5 : 0) * G>(); break;
        default: wait_vmcnt<(STAGES > 7 ? 6 : 0) * G>(); break;
    }
}

// ---- forward / dgrad: C[32 q][32 p] per workgroup, BK = 64, waves split K ------------------
//   Q is always ROW (X or dZ, k-contiguous).  P_ROW: W[p][k] (forward);  !P_ROW: W[k][p] (dgrad).
// ABL (tools/gemm_ablate.hip only; 0 in production): 1 = no DMA refill in the loop, 2 = no LDS
// fragment reads, 4 = no MFMA, 8 = no barrier / vmcnt wait.
template <bool P_ROW, int STAGES, class Epi, int ABL = 0>
__global__ void __launch_bounds__(256)
gemm_splitk_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ P, int ldp, int K,
                   int tiles_q, int tiles_p, int p_per_xcd, Epi epi) {
    constexpr int BK = 64, kTile = 32 * 64, kStage = 2 * kTile, G = 4;
    static_assert((STAGES - 2) * G <= 63, "vmcnt is a 6-bit counter");
    static_assert(STAGES * kStage >= 4 * 32 * 36, "ring must hold the split-K reduction buffer");

    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 32, p0 = tile_p * 32;

    __shared__ __attribute__((aligned(16))) float lds[STAGES * kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4;

    // per-lane DMA sources (k-tile 0); slot j = 16-byte position inside the 8 KB tile image
    const float* sq[2];
    const float* sp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = (wave + 4 * u) * 64 + lane;
        {
            const int row = j >> 4, c = (j & 15) ^ (row & 15);
            sq[u] = Q + (size_t)(q0 + row) * ldq + c * 4;
        }
        if (P_ROW) {
            const int row = j >> 4, c = (j & 15) ^ (row & 15);
            sp[u] = P + (size_t)(p0 + row) * ldp + c * 4;
        } else {
            const int r = j >> 3, k = r ^ ((r >> 2) & 1);
            sp[u] = P + (size_t)k * ldp + p0 + (j & 7) * 4;
        }
    }
    const size_t kstep_p = P_ROW ? (size_t)BK : (size_t)BK * ldp;
    auto issue = [&](int t, float* slot) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            lds_dma16(sq[u] + (size_t)t * BK, slot + (wave + 4 * u) * 256);
            lds_dma16(sp[u] + (size_t)t * kstep_p, slot + kTile + (wave + 4 * u) * 256);
        }
    };

    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (floats) inside a stage
    int oq[2], op[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 16 * a + li;
        oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
        op[a] = kTile + oq[a];                                   // P_ROW: same image shape
    }
    const int kq = 16 * wave + 4 * lh;                           // first k of this lane's 4-chunk

    const int nk = K / BK;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) issue(t, lds + t * kStage);

    for (int t = 0; t < nk; ++t) {
        const int rem = (nk - 1 - t) < (STAGES - 2) ? (nk - 1 - t) : (STAGES - 2);
        if (!(ABL & 8)) {
            wait_tile_landed<STAGES, G>((ABL & 1) ? 0 : rem);
            __builtin_amdgcn_s_barrier();   // every wave's share of tile t landed; tile t-1 fully read
        }
        asm volatile("" ::: "memory");
        const int tn = t + STAGES - 1;
        if (!(ABL & 1) && tn < nk) issue(tn, lds + (tn % STAGES) * kStage);   // refill the slot tile t-1 vacated
        const float* st = lds + (t % STAGES) * kStage;
        v4f fq[2], fp[2];
        v2f fc[4];
        if (ABL & 2) {
#pragma unroll
            for (int a = 0; a < 2; ++a) { fq[a] = v4f{1.f, 2.f, 3.f, 4.f} * (float)lane; fp[a] = fq[a] + 1.f; }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) fc[s2] = v2f{1.f + s2, 2.f} * (float)lane;
            asm volatile("" : "+v"(fq[0]), "+v"(fq[1]), "+v"(fp[0]), "+v"(fp[1]));
        } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) fq[a] = *reinterpret_cast<const v4f*>(st + oq[a]);
        if (P_ROW) {
#pragma unroll
            for (int b = 0; b < 2; ++b) fp[b] = *reinterpret_cast<const v4f*>(st + op[b]);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)       // global row k = kq + s2 lives in LDS row k ^ ((k>>2)&1) = k ^ (lh&1)
                fc[s2] = *reinterpret_cast<const v2f*>(st + kTile + (((kq + s2) ^ (lh & 1)) * 32) + 2 * li);
        }
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float pv = P_ROW ? fp[b][s2] : fc[s2][b];
                    if (ABL & 4) {
                        acc[a][b][0] += pv * fq[a][s2];
                    } else {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, fq[a][s2], acc[a][b], 0, 0, 0);
                    }
                }
    }

    // split-K reduction through LDS (fixed order: wave 0..3), then the epilogue on float4s.
    // D[i = 4*lh + r][j = li]: i indexes the P side, j the Q side.
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    constexpr int RS = 36;                                       // padded row stride of the partial tiles
    float* red = lds + wave * (32 * RS);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * a + li;
                const int pl = P_ROW ? (16 * b + 4 * lh + r) : (8 * lh + 2 * r + b);
                red[ql * RS + pl] = acc[a][b][r];
            }
    __syncthreads();
    {
        const int ql = tid >> 3, pl = (tid & 7) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (32 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epi.preload(q0 + ql, p0 + pl));
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
}

// ---- wgrad: G[64 q][64 p] per workgroup, reduction over batch rows (BK = 32), both operands COL
template <int STAGES, class Epi, int ABL = 0>
__global__ void __launch_bounds__(256)
gemm_wgrad_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ P, int ldp, int K,
                  int tiles_q, int tiles_p, int p_per_xcd, Epi epi) {
    constexpr int BK = 32, kTile = 32 * 64, kStage = 2 * kTile, G = 4;
    static_assert((STAGES - 2) * G <= 63, "vmcnt is a 6-bit counter");

    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 64, p0 = tile_p * 64;

    __shared__ __attribute__((aligned(16))) float lds[STAGES * kStage];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4;
    const int wq = (wave >> 1) * 32, wp = (wave & 1) * 32;

    const float* sq[2];
    const float* sp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = (wave + 4 * u) * 64 + lane;
        const int row = j >> 4, c = (j & 15) ^ ((row & 1) << 3);
        sq[u] = Q + (size_t)row * ldq + q0 + c * 4;
        sp[u] = P + (size_t)row * ldp + p0 + c * 4;
    }
    auto issue = [&](int t, float* slot) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            lds_dma16(sq[u] + (size_t)t * BK * ldq, slot + (wave + 4 * u) * 256);
            lds_dma16(sp[u] + (size_t)t * BK * ldp, slot + kTile + (wave + 4 * u) * 256);
        }
    };

    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    v2f bsum = v2f{0.f, 0.f};   // running column sums of the Q (dZ) fragments = bias gradient

    // lane (li, lh) reads row k = kk + lh, outputs (w + 2*li, w + 2*li + 1)
    const int sw = (lh & 1) << 3;                                // chunk swizzle of odd rows
    const int cq = wq + 2 * li, cp = wp + 2 * li;
    const int oq = lh * 64 + ((((cq >> 2) ^ sw)) << 2) + (cq & 3);
    const int op = kTile + lh * 64 + ((((cp >> 2) ^ sw)) << 2) + (cp & 3);

    const int nk = K / BK;
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
        if (t < nk) issue(t, lds + t * kStage);

    for (int t = 0; t < nk; ++t) {
        const int rem = (nk - 1 - t) < (STAGES - 2) ? (nk - 1 - t) : (STAGES - 2);
        if (!(ABL & 8)) {
            wait_tile_landed<STAGES, G>((ABL & 1) ? 0 : rem);
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("" ::: "memory");
        const int tn = t + STAGES - 1;
        if (!(ABL & 1) && tn < nk) issue(tn, lds + (tn % STAGES) * kStage);
        const float* st = lds + (t % STAGES) * kStage;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            v2f fq, fp;
            if (ABL & 2) {
                fq = v2f{1.f, 2.f} * (float)(lane + kk);
                fp = v2f{3.f, 4.f} * (float)(lane + kk);
                asm volatile("" : "+v"(fq), "+v"(fp));
            } else {
                fq = *reinterpret_cast<const v2f*>(st + oq + kk * 64);
                fp = *reinterpret_cast<const v2f*>(st + op + kk * 64);
            }
            bsum += fq;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (ABL & 4) {
                        acc[a][b][0] += fp[b] * fq[a];
                    } else {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[b], fq[a], acc[a][b], 0, 0, 0);
                    }
                }
        }
    }

    // lane holds q = wq + 2*li + a ; p = wp + 8*lh + 2*r + b  -> two float4 per q-row
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int q = q0 + wq + 2 * li + a, p = p0 + wp + 8 * lh;
        epi(q, p, v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]});
        epi(q, p + 4, v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]});
    }
    // bias gradient db[q] = sum over batch rows of dZ[:, q] (autograd of nn.Linear's bias): the
    // first p-tile's two q-halves (waves 0 and 2) own it; lanes lh = 0..3 hold k = lh (mod 4)
    if (epi.has_bias() && tile_p == 0 && (wave & 1) == 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            float v = bsum[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (lh == 0) epi.bias(q0 + wq + 2 * li + e, v);
        }
    }
    if (epi.loss.out && bid == 0 && wave == 3) finalize_loss_wave(epi.loss, lane);
}

// ---- wave-specialised LDS-DMA ring: 8 waves = 4 compute (split K, ds_read + MFMA only, fragments
// double-buffered in registers) + 4 loaders (global_load_lds only).  The loaders run one tile ahead
// of what the compute waves read, STAGES-1 tiles ahead in flight.
template <bool P_ROW, int STAGES, class Epi>
__global__ void __launch_bounds__(512)
gemm_splitk_wsN_kernel(GemmArgs ga, Epi epi) {
    constexpr int BK = 64, kTile = 32 * 64, kStage = 2 * kTile, G = 4;
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 32, p0 = tile_p * 32;

    __shared__ __attribute__((aligned(16))) float lds[STAGES * kStage];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4;
    const int nk = K / BK;

    if (wave >= 4) {
        // ---------------- loader waves ----------------
        const int u0 = wave - 4;
        const float* sq[2];
        const float* sp[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (u0 + 4 * u) * 64 + lane;
            {
                const int row = j >> 4, c = (j & 15) ^ (row & 15);
                sq[u] = Q + (size_t)(q0 + row) * ldq + c * 4;
            }
            if (P_ROW) {
                const int row = j >> 4, c = (j & 15) ^ (row & 15);
                sp[u] = P + (size_t)(p0 + row) * ldp + c * 4;
            } else {
                const int r = j >> 3, k = r ^ ((r >> 2) & 1);
                sp[u] = P + (size_t)k * ldp + p0 + (j & 7) * 4;
            }
        }
        const size_t kstep_p = P_ROW ? (size_t)BK : (size_t)BK * ldp;
        auto issue = [&](int t) {
            float* slot = lds + (t % STAGES) * kStage;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                lds_dma16(sq[u] + (size_t)t * BK, slot + (u0 + 4 * u) * 256);
                lds_dma16(sp[u] + (size_t)t * kstep_p, slot + kTile + (u0 + 4 * u) * 256);
            }
        };
#pragma unroll
        for (int t = 0; t < STAGES - 1; ++t)
            if (t < nk) issue(t);
        {   // tile 0 landed
            const int y = nk - 1 < STAGES - 2 ? nk - 1 : STAGES - 2;
            wait_tile_landed<STAGES, G>(y);
        }
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            // tile t+1 landed: younger tiles in flight = t+2 .. min(t+STAGES-2, nk-1)
            int y = nk - 2 - t;
            if (y > STAGES - 3) y = STAGES - 3;
            if (y < 0) y = 0;
            wait_tile_landed<STAGES, G>(y);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + STAGES - 1 < nk) issue(t + STAGES - 1);
        }
        __syncthreads();
        __syncthreads();
        return;
    }

    // ---------------- compute waves ----------------
    int oq[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 16 * a + li;
        oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
    }
    const int kq = 16 * wave + 4 * lh;
    struct Frag { v4f q[2], p[2]; v2f c[4]; };
    auto fread = [&](const float* st, Frag& f) {
#pragma unroll
        for (int a = 0; a < 2; ++a) f.q[a] = *reinterpret_cast<const v4f*>(st + oq[a]);
        if (P_ROW) {
#pragma unroll
            for (int b = 0; b < 2; ++b) f.p[b] = *reinterpret_cast<const v4f*>(st + kTile + oq[b]);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
                f.c[s2] = *reinterpret_cast<const v2f*>(st + kTile + (((kq + s2) ^ (lh & 1)) * 32) + 2 * li);
        }
    };
    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    auto mfmas = [&](const Frag& f) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float pv = P_ROW ? f.p[b][s2] : f.c[s2][b];
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, f.q[a][s2], acc[a][b], 0, 0, 0);
                }
    };
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    Frag F0, F1;
    fread(lds, F0);
    for (int t0 = 0; t0 < nk; t0 += 2) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const int t = t0 + d;
            if (t < nk) {
                Frag& F = d ? F1 : F0;
                Frag& Gf = d ? F0 : F1;
                __builtin_amdgcn_s_barrier();          // tile t+1 landed (all loaders), tile t-1's slot is free
                asm volatile("" ::: "memory");
                fread(lds + ((t + 1) % STAGES) * kStage, Gf);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(F);
            }
        }
    }
    __syncthreads();
    constexpr int RS = 36;
    float* red = lds + wave * (32 * RS);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * a + li;
                const int pl = P_ROW ? (16 * b + 4 * lh + r) : (8 * lh + 2 * r + b);
                red[ql * RS + pl] = acc[a][b][r];
            }
    __syncthreads();
    {
        const int ql = tid >> 3, pl = (tid & 7) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (32 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epi.preload(q0 + ql, p0 + pl));
    }
}

}  // namespace pvae
