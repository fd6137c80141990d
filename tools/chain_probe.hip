// A stack's hidden forward layers as ONE persistent launch, against the same layers as back-to-back launches of the
// production kernel.  256 rows x 1024 x 1024 per layer, ReLU, L layers chained X_l -> X_{l+1}.
//
// The chain kernel keeps the production kernel's workgroup (4 loader waves landing [32][64] Q + [32][64] P k-tiles by
// LDS-DMA into a six-slot ring, 4 compute waves that split every k-tile four ways, super-steps of two k-tiles per
// workgroup barrier, fixed-order split-K reduction through LDS) and its arithmetic, so its outputs equal the launch
// chain's bit for bit (checked).  What changes is the layer edge:
//   * ROW mapping (mode bit 2 clear): workgroup b owns row block b % 8 -- the XCD it is dispatched to -- and column tile
//     b / 8 of EVERY layer.  A layer's output tile is stored with plain stores (they stay in that XCD's L2), each compute
//     wave then sets a tagged word; the loader waves of the 32 workgroups of the row block poll the 128 words of their row
//     block (sc1 loads: served by the L2, past the CU's L1) and fetch the next layer's Q tiles with sc1 LDS-DMA.  Every XCD
//     streams the whole weight matrix (8 x 4 MB per layer over the fabric).
//   * COLUMN mapping (bit 2 set): the production launches' tile -> XCD map (an XCD owns 4 column tiles and all 8 row
//     blocks; W is fetched once chip-wide); the hand-over crosses XCDs, so outputs and words are written through (sc1).
//   * bit 0: the loaders keep streaming the NEXT layer's P (weight) k-tiles into the ring across the edge -- W does not
//     depend on the previous layer; only the Q half of a slot waits for the words.  Clear: P and Q are requested together
//     once the words have arrived (what a drain-barrier-refill edge does).
//   * bit 1: output tiles stored write-through (sc1) also in the ROW mapping.
//   * bit 3: the loaders wait per k-tile for the eight words of the two producers whose columns it holds, instead of for
//     all 128 words of the row block before the first tile (a late producer then only stalls the tile that needs it).
// Per-edge timeline (wall clock, 100 MHz) of every workgroup: main loop done -> words stored -> all 128 words of the row
// block seen by the loaders -> Q tiles requested -> tile 0 seen by the compute waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/chain_probe.hip -o ab_libs/chain_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "../physicsvae_amd/csrc/pvae_gemm.h"
using namespace pvae;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kRows = 256, kN = 1024, kNK = kN / 64, kSlots = 6, kTileF = 32 * 64, kStageF = 2 * kTileF;
constexpr int kRedF = 4 * 32 * 36;
constexpr int kChainLds = (kSlots * kStageF + kRedF) * 4;
constexpr int kMarks = 8;

struct ChainArgs {
    const float* X0;          // [256][1024]
    float* act;               // L x [256][1024]: act + l * 256 * 1024 = output of layer l
    const float* W;           // L x [1024][1024]
    const float* bias;        // L x [1024]
    unsigned* words;          // [L][8 row blocks][128]
    unsigned long long* marks;   // [256][L][kMarks] (may be null)
    unsigned* err;            // [0]: polls that gave up; [1 + xcc]: census of workgroups whose XCC differs from b % 8
    unsigned seq;
    int L;
};

template <bool SC1>
__device__ inline void dma16(const float* src, float* dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, SC1 ? 16 : 0);
}
__device__ inline void wait_vm(int n) {           // (uniform n; the tail of a layer needs counts the steady state does not)
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 16: wait_vmcnt<16>(); break;
        default: wait_vmcnt<0>(); break;
    }
}
__device__ inline unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int MODE>
__global__ void __launch_bounds__(512) chain_kernel(ChainArgs a) {
    constexpr bool kPrefetch = MODE & 1, kColumns = MODE & 4, kWriteThrough = (MODE & 2) || kColumns, kPerTile = MODE & 8;
    extern __shared__ float lds[];
    float* red = lds + kSlots * kStageF;
    const int bid = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_q, tile_p;
    if (kColumns) { const int xcd = bid & 7, loc = bid >> 3; tile_p = xcd * 4 + loc / 8; tile_q = loc % 8; }
    else { tile_q = bid & 7; tile_p = bid >> 3; }
    const int q0 = tile_q * 32, p0 = tile_p * 32;
    const int L = a.L;
    const unsigned seq = a.seq;
    if (tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
        if (xcc != (unsigned)(bid & 7)) atomicAdd(a.err + 1 + xcc, 1u);
    }
    auto mark = [&](int l, int id) {
        if (a.marks) a.marks[((size_t)bid * L + l) * kMarks + id] = wall_clock64();
    };
    if (wave >= 4) {
        // ---------------- loader waves ----------------
        const int u0 = wave - 4;
        size_t qoff[2], poff[2];
        int dst[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = (u0 + 4 * u) * 64 + lane, row = j >> 4, c = (j & 15) ^ (row & 15);
            qoff[u] = (size_t)(q0 + row) * kN + c * 4;
            poff[u] = (size_t)(p0 + row) * kN + c * 4;
            dst[u] = (u0 + 4 * u) * 256;
        }
        auto issue_q = [&](int l, int t) {
            const float* X = l == 0 ? a.X0 : a.act + (size_t)(l - 1) * kRows * kN;
            float* slot = lds + ((l * kNK + t) % kSlots) * kStageF;
#pragma unroll
            for (int u = 0; u < 2; ++u) dma16<true>(X + qoff[u] + (size_t)t * 64, slot + dst[u]);
        };
        auto issue_p = [&](int l, int t) {
            const float* W = a.W + (size_t)l * kN * kN;
            float* slot = lds + ((l * kNK + t) % kSlots) * kStageF + kTileF;
#pragma unroll
            for (int u = 0; u < 2; ++u) dma16<false>(W + poff[u] + (size_t)t * 64, slot + dst[u]);
        };
        if (kPrefetch)
            for (int t = 0; t < 5; ++t) issue_p(0, t);
        for (int l = 0; l < L; ++l) {
            const bool has_next = l + 1 < L;
            // words of the previous layer's row block as this wave last saw them: bit i of m0 / m1 = word i / 64 + i carries seq
            unsigned long long m0 = ~0ull, m1 = ~0ull;
            const unsigned* wds = a.words + ((size_t)(l > 0 ? l - 1 : 0) * 8 + tile_q) * 128;
            auto poll = [&]() {
                const unsigned v0 = ld_sc1(wds + lane), v1 = ld_sc1(wds + 64 + lane);
                m0 = __builtin_amdgcn_ballot_w64(v0 == seq);
                m1 = __builtin_amdgcn_ballot_w64(v1 == seq);
            };
            // k-tile t reads the columns of producers 2t, 2t+1: words [8t, 8t + 8)
            auto ensure = [&](int t) {
                if (l == 0) return;
                unsigned spins = 0;
                for (;;) {
                    const unsigned long long m = t < 8 ? m0 >> (8 * t) : m1 >> (8 * (t - 8));
                    if ((m & 0xffull) == 0xffull) break;
                    if (++spins > (1u << 16)) { if (lane == 0) atomicAdd(a.err, 1u); break; }
                    if (spins > 1) __builtin_amdgcn_s_sleep(1);
                    poll();
                }
            };
            if (l > 0) {
                if (kPerTile) { m0 = m1 = 0; ensure(0); }
                else {
                    unsigned spins = 0;
                    for (;;) {
                        poll();
                        if ((m0 & m1) == ~0ull) break;
                        if (++spins > (1u << 16)) { if (lane == 0) atomicAdd(a.err, 1u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
                if (tid == 256) mark(l, 3);
            }
            if (kPrefetch) {
                for (int t = 0; t < 5; ++t) { if (kPerTile) ensure(t); issue_q(l, t); }
                if (tid == 256) mark(l, 4);
                wait_vm(8);                                          // tile 0: Q(1..4) are younger
            } else {
                for (int t = 0; t < 5; ++t) { if (kPerTile) ensure(t); issue_q(l, t); issue_p(l, t); }
                if (tid == 256) mark(l, 4);
                wait_vm(16);
            }
            __builtin_amdgcn_s_barrier();                            // A: tile 0 landed
            asm volatile("" ::: "memory");
            for (int t0 = 0; t0 < kNK; t0 += 2) {
                // tiles t0+1, t0+2 landed: what this wave issued after them
                int y = 0;
                auto cost = [&](int t) { return t < kNK ? 4 : (kPrefetch && has_next ? 2 : 0); };
                if (t0 == 0) y = kPrefetch ? 4 : 8;                  // Q(3), Q(4) / full tiles 3, 4
                else {
                    y = cost(t0 + 3) + cost(t0 + 4);
                    if (t0 + 2 >= kNK) y += cost(t0 + 2);            // (the next layer's tile 0 is not waited for here)
                }
                wait_vm(y);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int d = 5; d <= 6; ++d) {
                    const int t = t0 + d;
                    if (t < kNK) { if (kPerTile) ensure(t); issue_q(l, t); issue_p(l, t); }
                    else if (kPrefetch && has_next) issue_p(l + 1, t - kNK);
                }
            }
            __builtin_amdgcn_s_barrier();                            // B1
            __builtin_amdgcn_s_barrier();                            // B2
            asm volatile("" ::: "memory");
        }
        wait_vmcnt<0>();
        return;
    }
    // ---------------- compute waves ----------------
    const int li = lane & 15, lh = lane >> 4;
    int oq[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int row = 16 * x + li;
        oq[x] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
    }
    struct Frag { v4f q[2], p[2]; };
    auto fread = [&](const float* st, Frag& f) {
#pragma unroll
        for (int x = 0; x < 2; ++x) f.q[x] = *reinterpret_cast<const v4f*>(st + oq[x]);
#pragma unroll
        for (int b = 0; b < 2; ++b) f.p[b] = *reinterpret_cast<const v4f*>(st + kTileF + oq[b]);
    };
    const int ql = tid >> 3, pl = (tid & 7) << 2;
    for (int l = 0; l < L; ++l) {
        v4f acc[2][2];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[x][b] = v4f{0.f, 0.f, 0.f, 0.f};
        auto mfmas = [&](const Frag& f) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[x][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.p[b][s2], f.q[x][s2], acc[x][b], 0, 0, 0);
        };
        const v4f bias = *reinterpret_cast<const v4f*>(a.bias + (size_t)l * kN + p0 + pl);
        const int T0 = l * kNK;
        __builtin_amdgcn_s_barrier();                                // A: tile 0 landed
        asm volatile("" ::: "memory");
        if (tid == 0) mark(l, 0);
        Frag F0, F1;
        fread(lds + (T0 % kSlots) * kStageF, F0);
        for (int t0 = 0; t0 < kNK; t0 += 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            fread(lds + ((T0 + t0 + 1) % kSlots) * kStageF, F1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(F0);
            fread(lds + ((T0 + t0 + 2) % kSlots) * kStageF, F0);     // (past the layer's end: never used)
            __builtin_amdgcn_sched_barrier(0);
            mfmas(F1);
        }
        if (tid == 0) mark(l, 1);
        __syncthreads();                                             // B1
        store_partial_32x32<true>(red + wave * (32 * 36), acc, li, lh);
        __syncthreads();                                             // B2
        v4f v = *reinterpret_cast<const v4f*>(red + ql * 36 + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(red + w * (32 * 36) + ql * 36 + pl);
        v += bias;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        float* o = a.act + (size_t)l * kRows * kN + (size_t)(q0 + ql) * kN + p0 + pl;
        if (kWriteThrough) store_stream(o, v);
        else *reinterpret_cast<v4f*>(o) = v;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned* w = a.words + ((size_t)l * 8 + tile_q) * 128 + tile_p * 4 + wave;
            if (kWriteThrough) __hip_atomic_store(w, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *(volatile unsigned*)w = seq;
        }
        if (tid == 0) mark(l, 2);
    }
}

// reference: the production kernel, one launch per layer
static int launch_chain_reference(const float* X0, float* act, const float* W, const float* bias, int L, int rowxcd, hipStream_t st) {
    const GemmGrid g = make_grid(kRows, kN, 32, 32);
    for (int l = 0; l < L; ++l) {
        EpiBiasAct e{act + (size_t)l * kRows * kN, kN, bias + (size_t)l * kN, 1};
        e.n_valid = kN;
        GemmArgs ga{l == 0 ? X0 : act + (size_t)(l - 1) * kRows * kN, kN, W + (size_t)l * kN * kN, kN, kN, g.tiles_q, g.tiles_p, g.p_per_xcd};
        ga.rowxcd = rowxcd;
        hipLaunchKernelGGL((gemm_splitk_ws_kernel<true, EpiBiasAct>), dim3(g.grid), dim3(kWsThreads), 0, st, PVAE_GA_PASS(ga), e);
    }
    return 0;
}

template <int MODE>
static int run_chain(ChainArgs a, hipStream_t st) {
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)chain_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, kChainLds)); once = true; }
    hipLaunchKernelGGL((chain_kernel<MODE>), dim3(256), dim3(512), kChainLds, st, a);
    return 0;
}
static int run_mode(int mode, const ChainArgs& a, hipStream_t st) {
    switch (mode) {
        case 0: return run_chain<0>(a, st);
        case 1: return run_chain<1>(a, st);
        case 2: return run_chain<2>(a, st);
        case 3: return run_chain<3>(a, st);
        case 4: return run_chain<4>(a, st);
        case 5: return run_chain<5>(a, st);
        case 9: return run_chain<9>(a, st);
        case 12: return run_chain<12>(a, st);
        case 13: return run_chain<13>(a, st);
    }
    return 1;
}

int main(int argc, char** argv) {
    const int Lmax = 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    float *X0, *W, *B, *act_ref, *act; unsigned *words, *err; unsigned long long* marks;
    const size_t panel = (size_t)kRows * kN, wsz = (size_t)kN * kN;
    CK(hipMalloc(&X0, panel * 4)); CK(hipMalloc(&W, Lmax * wsz * 4)); CK(hipMalloc(&B, Lmax * kN * 4));
    CK(hipMalloc(&act_ref, Lmax * panel * 4)); CK(hipMalloc(&act, Lmax * panel * 4));
    CK(hipMalloc(&words, Lmax * 8 * 128 * 4)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&marks, (size_t)256 * Lmax * kMarks * 8));
    CK(hipMemset(words, 0, Lmax * 8 * 128 * 4)); CK(hipMemset(err, 0, 64));
    {
        std::mt19937 rng(1);
        std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> h(panel);
        for (auto& v : h) v = nd(rng);
        CK(hipMemcpy(X0, h.data(), panel * 4, hipMemcpyHostToDevice));
        std::vector<float> w(wsz);
        const float s = std::sqrt(2.0f / kN);
        for (int l = 0; l < Lmax; ++l) {
            for (auto& v : w) v = nd(rng) * s;
            CK(hipMemcpy(W + l * wsz, w.data(), wsz * 4, hipMemcpyHostToDevice));
        }
        std::vector<float> b(Lmax * kN);
        for (auto& v : b) v = nd(rng) * 0.05f;
        CK(hipMemcpy(B, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    unsigned seq = 0;
    const int iters = 200;
    std::vector<float> ref(panel), got(panel);
    for (int L : {5, 20}) {
        printf("==== %d layers of 256 x 1024 x 1024 ====\n", L);
        for (int rowxcd = 0; rowxcd < 2; ++rowxcd) {
            for (int i = 0; i < 10; ++i) launch_chain_reference(X0, act_ref, W, B, L, rowxcd, st);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch_chain_reference(X0, act_ref, W, B, L, rowxcd, st);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("launches, %s: %.2f us per chain = %.2f us per layer\n", rowxcd ? "row blocks per XCD   " : "production tile map  ",
                   ms * 1e3 / iters, ms * 1e3 / iters / L);
        }
        CK(hipMemcpy(ref.data(), act_ref + (size_t)(L - 1) * panel, panel * 4, hipMemcpyDeviceToHost));
        for (int mode : {1, 9, 5, 4, 13, 12}) {
            ChainArgs a{X0, act, W, B, words, nullptr, err, 0, L};
            CK(hipMemset(act, 0, Lmax * panel * 4));
            for (int i = 0; i < 10; ++i) { a.seq = ++seq; if (run_mode(mode, a, st)) return 1; }
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) { a.seq = ++seq; run_mode(mode, a, st); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // one more with marks
            a.marks = marks; a.seq = ++seq;
            run_mode(mode, a, st);
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(got.data(), act + (size_t)(L - 1) * panel, panel * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < panel; ++i) bad += memcmp(&got[i], &ref[i], 4) != 0;
            unsigned herr[9]; CK(hipMemcpy(herr, err, 36, hipMemcpyDeviceToHost));
            unsigned mis = 0; for (int i = 1; i < 9; ++i) mis += herr[i];
            printf("chain mode %d (%s%s%s): %.2f us per chain = %.2f us per layer; %zu of %zu outputs differ from the launches; polls given up %u; misplaced workgroups %u\n",
                   mode, (mode & 4) ? "column map, write-through" : "row block per XCD", (mode & 1) ? ", W streams across the edge" : ", refill after the edge",
                   (mode & 2) ? ", write-through stores" : ((mode & 8) ? ", Q tiles gated tile by tile" : ""), ms * 1e3 / iters, ms * 1e3 / iters / L, bad, panel, herr[0], mis);
            CK(hipMemset(err, 0, 64));
            // per-edge timeline: medians over the 256 workgroups, layers 1 .. L-1
            std::vector<unsigned long long> h((size_t)256 * L * kMarks);
            CK(hipMemcpy(h.data(), marks, h.size() * 8, hipMemcpyDeviceToHost));
            auto med = [&](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
            std::vector<double> d_loop, d_flag, d_seen, d_req, d_tile0, d_edge;
            for (int b = 0; b < 256; ++b)
                for (int l = 1; l < L; ++l) {
                    const unsigned long long* p = &h[((size_t)b * L + (l - 1)) * kMarks];      // previous layer
                    const unsigned long long* c = &h[((size_t)b * L + l) * kMarks];
                    d_loop.push_back((double)(p[1] - p[0]) * 0.01);     // main loop of the previous layer
                    d_flag.push_back((double)(p[2] - p[1]) * 0.01);     // loop done -> words stored
                    d_seen.push_back((double)((long long)(c[3] - p[2])) * 0.01);     // own words stored -> all 128 seen
                    d_req.push_back((double)(c[4] - c[3]) * 0.01);      // seen -> Q tiles requested
                    d_tile0.push_back((double)(c[0] - c[4]) * 0.01);    // requested -> tile 0 seen by the compute waves
                    d_edge.push_back((double)(c[0] - p[1]) * 0.01);     // the edge: loop done -> next loop starts
                }
            printf("    medians over workgroups x edges: main loop %.2f us | loop done -> words stored %.2f | stored -> all 128 words seen %.2f | seen -> Q requested %.2f | requested -> tile 0 seen %.2f | EDGE (loop done -> next loop starts) %.2f us\n",
                   med(d_loop), med(d_flag), med(d_seen), med(d_req), med(d_tile0), med(d_edge));
        }
    }
    return 0;
}
