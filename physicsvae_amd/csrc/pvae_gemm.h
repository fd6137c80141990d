// pvae_gemm.h -- fp32 MFMA kernels for the three contractions of an MLP layer at minibatch
// scale (M = 32..512 rows).  gfx950 only.
//
//   C[q][p] = sum_k Q(q,k) * P(p,k)            C row-major, p contiguous
//
//   forward  : q = batch row, p = out feature, k = in feature   Q = X  (k-contig)  P = W  (k-contig)
//   dgrad    : q = batch row, p = in feature,  k = out feature  Q = dZ (k-contig)  P = W  (p-contig)
//   wgrad    : q = out feat,  p = in feature,  k = batch row    Q = dZ (q-contig)  P = X  (p-contig)
//
// An operand is "ROW" when its reduction index is the contiguous one in memory and "COL" when
// its output index is.  Tiles are fetched with full-line 16-byte accesses along the contiguous
// index and kept in that orientation in LDS (no transposes); the MFMA is
// v_mfma_f32_16x16x4_f32 (exact fp32 fma chain, 157 TFLOP/s peak).  The MFMA "A" slot takes the P
// fragment and the "B" slot the Q fragment, so D = C^T-tile: each lane ends up with consecutive
// p of one q -> float4 epilogue loads/stores.
//
// Block -> tile mapping is XCD-aware: block b runs on XCD b%8 (observed dispatch order, used for
// L2 locality only); each XCD is given a contiguous range of p-tiles and all q-tiles of it, so
// the P panel it streams is fetched once per XCD and the (small) Q operand stays in that L2.
//
// What bounds these kernels (tools/timeline_probe.hip, pair_probe.hip, wgrad_probe.hip, mfma_rate.hip
// on MI355X): v_mfma_f32_16x16x4_f32 issues every 32 cycles at 2.35-2.39 GHz,
// i.e. 218 ns for the 16 MFMAs a compute wave spends on a 32x32x64 k-tile.  A forward / dgrad
// launch is ~1.7 us of launch + drain, ~1 us until the first k-tile is in LDS, 272 ns per k-tile
// (the per-CU load path delivers a 16 KB tile in 170-240 ns when nothing else runs) and ~0.5 us of
// split-K reduction + epilogue; the weight-gradient body runs at ~63 % of the matrix pipe's issue
// rate (one LDS round trip in front of every 8 MFMAs) and its Adam epilogue moves 24 B per
// parameter.  hipcc's wait-count pass only stays exact in branch-free loop bodies, which is why the
// register-staged loops are written as full groups plus a guarded tail.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>
// Timeline probe (tools/timeline_probe.hip defines PVAE_TIMELINE): thread `tid_` of each workgroup
// stores the 100 MHz wall clock at numbered points of the wave-specialised kernel, and GemmArgs::krot
// values >= 2 switch parts of it off (ablation).  Compiled out of the library.
#ifdef PVAE_TIMELINE
extern __device__ unsigned long long* g_timeline;
#define PVAE_MARK(tid_, id_)                                                                   \
    do {                                                                                       \
        if ((int)threadIdx.x == (tid_)) g_timeline[(size_t)blockIdx.x * 8 + (id_)] = wall_clock64(); \
    } while (0)
#define PVAE_PROBE(mode_) (ga.krot == (mode_))
// slot 6: HW_ID (bits 8..11 CU, 12..13 SH, 13..15 SE on gfx9), slot 7: XCC_ID
#define PVAE_MARK_HW()                                                                         \
    do {                                                                                       \
        if (threadIdx.x == 0) {                                                                \
            g_timeline[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
            g_timeline[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)); \
        }                                                                                      \
    } while (0)
#else
#define PVAE_MARK_HW() do { } while (0)
#define PVAE_MARK(tid_, id_) do { } while (0)
#define PVAE_PROBE(mode_) false
#endif
#include <cstring>

namespace pvae {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// Streamed 16-byte store.  Everything these kernels write (activations, input gradients, weight
// gradients, Adam's p / m / v) is consumed by a LATER launch, on other CUs and mostly other XCDs, so
// keeping the lines dirty in the writer's L2 buys nothing: they are written back when the kernel ends,
// and the next launch waits for that (MI355X_MICROARCH.md, `boundary` row: + dirty bytes / 6 TB/s, i.e.
// ~2.8 us behind the 17 MB a fused backward launch leaves behind).  `sc1` stores are written through
// while the kernel is still computing.  (Measured alternatives, docs/experiments.md round 3: `sc0 sc1` the same,
// `nt` / `sc1 nt` 13 % slower -- the consumer launch then finds nothing in the Infinity Cache; plain stores +1 %.)
__device__ inline void store_stream(float* p, const v4f& v) {
    // (s_nop 1: the store reads its four data registers over the following states and hipcc pads nothing
    //  inside an asm statement -- cdna_hip_programming.md 5.7)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// ---------------------------------------------------------------------------------------
// The contraction kernels.
//
// Fragments are fetched with as few, as wide DS reads as the MFMA layout allows, because with
// one wave per SIMD ds_read_b32/b64 issue at ~1/5 of the LDS rate (MI355X_MICROARCH.md LDS
// table) while ds_read_b128 runs at full rate:
//   * k-contiguous ("ROW") operands: ONE ds_read_b128 = 4 k-values of one row per lane; lane
//     (i, h) holds k = 16c + 4h + s, MFMA step s takes element s (the k order inside a 16-chunk
//     is a permutation, shared by both operands);
//   * output-contiguous ("COL") operands: ds_read_b64 = 2 consecutive outputs of one k per
//     lane, feeding TWO interleaved MFMA tiles (tile u owns outputs 2i + u);
//   * the 4 waves of a workgroup split K instead of the tile (forward / dgrad): every wave owns
//     the whole 32x32 output as 2x2 MFMA tiles, so each fragment feeds two MFMAs; partial tiles
//     are summed through LDS in a fixed order at the end.
// Tiles travel global -> VGPR (plain global_load_dwordx4, D register sets per lane, see kRegDepth*) ->
// ds_write_b128 into a 2-slot LDS image.  The image is lane-linear (slot j = 16-byte chunk j of
// the tile) with the chunk order XOR-permuted on the way in and, identically, on the fragment
// read (involutions), which removes bank conflicts without padding:
//   ROW tile [32][64]   b128 reads : chunk ^= row & 15
//   COL tile [64 k][32] b64 reads  : LDS row = k ^ ((k >> 2) & 1)   (rows k, k+4 of one 32-lane
//                                    group land in different bank halves)
//   COL tile [32 k][64] b64 reads  : chunk ^= (row & 1) << 3        (rows k, k+1 likewise)
// (An LDS-DMA ring variant of the same kernels -- global_load_lds, 8 stages, counted vmcnt --
// measured ~12 % slower because a DMA instruction costs ~150 issue cycles on the wave that also
// feeds the matrix pipe; it was dropped after round 1, the measurement is in DESIGN.md section 4.)
// ---------------------------------------------------------------------------------------
// ---- forward / dgrad: C[32 q][32 p] per workgroup, BK = 64, the 4 waves split K -------------
//   Q is always ROW (X or dZ, k-contiguous).  P_ROW: W[p][k] (forward);  !P_ROW: W[k][p] (dgrad).
// ABL (tools/pair_lab.hip, tools/pair_probe.hip only; 0 in production): 1 = no tile staging in the loop, 2 = no LDS
// fragment reads, 4 = no MFMA, 8 = no barrier.
// ---------------------------------------------------------------------------------------
// minibatch staging (device side; launched as stage_batch_kernel or as tail blocks of the step's
// last weight-gradient launch)
// ---------------------------------------------------------------------------------------
// One call = one (padded) batch row r of time step t.  Builds the network input panels and the two
// target panels from either the HBM-resident demonstration set (window_row != null: step t of
// window i reads rows s+t and s+t+1 of `states`, row s+t of `actions`; tpv:133-156 windows,
// tm:52-56 float64->float32, tm:166-175 collate) or explicit x[rows][L][2Db] / y[rows][L][Da]
// (tpv:365-376).  Pad rows and pad columns are written as zeros so that every GEMM can run on
// whole tiles without bounds checks.  Time step t lives in row block t of every panel
// (pvae_layout.h).  For t > 0 the current-state columns are left zero here: they are the world
// model's own prediction of step t-1 (tpv:421), copied in by scatter_state_kernel during the
// forward pass.  `wm_pred` (L > 1 only) is the input panel of the world-model invocations that
// take the decoder's action.
struct StageArgs {
    const float* states;
    const float* next_states;     // null: the second half of x / the target s2 is the NEXT row of `states`;
                                  // else row-aligned with `states` (cond "rel": s_{t+1} - s_t, tpv:149-150)
    const float* actions;
    const int32_t* window_row;
    long long first_window;
    const float* x;
    const float* y;
    int rows, rows_pad, Db, Da, L;
    float* te_in; int ld_te;
    float* md_in; int ld_md;
    float* wm_in; int ld_wm;
    float* s2; int ld_s2;
    float* act_t; int ld_a;
    float* wm_pred;
    float* pr_in; int ld_pr;      // input panel of the learned prior stack [s1 | 0] (null: no such stack)
    int in_off = 0;               // input subsets (pvae_config.te_inputs / md_inputs): blocks written as ZEROS -- 1: the encoder's
                                  // s_t, 2: the encoder's s_{t+1}, 4: the decoder's s_t (the weights there are structural zeros;
                                  // a zero operand keeps their gradient exactly zero)
};

__device__ inline void stage_row(const StageArgs& a, int r, int t, int rows_pad) {
    const int Db = a.Db, Da = a.Da;
    const size_t prow = (size_t)t * rows_pad + r;          // row inside the stacked panels
    const bool valid = r < a.rows;
    const bool first = t == 0;
    const float* p1 = nullptr;
    const float* p2 = nullptr;
    const float* pa = nullptr;
    if (valid) {
        if (a.window_row) {
            const long long s = (long long)a.window_row[a.first_window + r] + t;
            p1 = a.states + s * Db;
            p2 = a.next_states ? a.next_states + s * Db : p1 + Db;
            pa = a.actions + s * Da;
        } else {
            p1 = a.x + ((size_t)r * a.L + t) * 2 * Db;
            p2 = p1 + Db;
            pa = a.y ? a.y + ((size_t)r * a.L + t) * Da : nullptr;
        }
    }
    int ld_max = a.ld_te;
    if (a.ld_md > ld_max) ld_max = a.ld_md;
    if (a.ld_wm > ld_max) ld_max = a.ld_wm;
    for (int c = threadIdx.x; c < ld_max; c += 256) {
        if (a.pr_in && c < a.ld_pr) a.pr_in[prow * a.ld_pr + c] = (valid && first && c < Db) ? p1[c] : 0.f;
        const float v1 = (valid && first && c < Db) ? p1[c] : 0.f;
        const float v2 = (valid && c < Db) ? p2[c] : 0.f;
        const float va = (valid && pa && c < Da) ? pa[c] : 0.f;
        if (c < a.ld_te) {
            float u = v1;
            if (c >= Db) u = (valid && c < 2 * Db) ? p2[c - Db] : 0.f;
            if (a.in_off & (c < Db ? 1 : 2)) u = 0.f;
            a.te_in[prow * a.ld_te + c] = u;
        }
        if (c < a.ld_md) a.md_in[prow * a.ld_md + c] = (a.in_off & 4) ? 0.f : v1;      // z columns filled by reparam
        if (c < a.ld_wm) {
            float u = v1;
            if (c >= Db) u = (valid && pa && c < Db + Da) ? pa[c - Db] : 0.f;
            a.wm_in[prow * a.ld_wm + c] = u;
            if (a.wm_pred) a.wm_pred[prow * a.ld_wm + c] = c < Db ? v1 : 0.f;   // a_hat columns filled by the decoder
        }
        if (c < a.ld_s2) a.s2[prow * a.ld_s2 + c] = v2;
        if (c < a.ld_a) a.act_t[prow * a.ld_a + c] = va;
    }
}

// The same row by ONE wave with every source load in flight before the first store (the stand-alone gather launch:
// stage_batch_kernel, four rows per workgroup).  Lane l owns columns l + 64 k of every panel, so each load and each
// store instruction of the wave covers 256 contiguous bytes -- whole 128-byte lines on the panel side (rows are
// 64-float aligned), and on the source side as well as rows of dim_body / dim_action floats allow (they are 4-byte
// aligned only, which is why the copy is dword-granular: 16-byte accesses would straddle).  Every access is a BUFFER
// access through a descriptor that spans exactly one source row / one panel row: a load past the row's end returns 0
// and a store past it is dropped by the address unit, which is precisely "pad columns are zero" / "this panel is
// narrower" -- so the body has no branch at all and hipcc's wait counts stay exact: per chunk of 512 columns up to
// 5 x 8 independent loads ([s_t | s_{t+1}] is ONE contiguous run of 2 Db floats of `states`; the target panel and the
// action columns of the world-model panel re-read it at their own lane mapping: L1 hits) and only then the stores,
// which retire in the background while the wave's next chunk / the SIMD's other seven waves load.  The dataset rows
// are read once per epoch: nontemporal loads.  Same values as stage_row / stage_row_vec, bit for bit (a copy).
// `w` = the wave's index inside the workgroup, wave-uniform by construction (readfirstlane'd by the caller).
__device__ inline void stage_row_wave(const StageArgs& a, int r, int t, int rows_pad, int lane) {
    const int Db = a.Db, Da = a.Da;
    const size_t prow = (size_t)t * rows_pad + r;
    const bool valid = r < a.rows;
    const bool first = t == 0;
    const float* p1 = a.te_in;                          // (any mapped address: descriptors of absent rows have length 0)
    const float* p2 = a.te_in;
    const float* pa = a.te_in;
    bool have_a = false;
    if (valid) {
        if (a.window_row) {
            const long long s = (long long)__builtin_amdgcn_readfirstlane(a.window_row[a.first_window + r]) + t;
            p1 = a.states + s * Db;
            p2 = a.next_states ? a.next_states + s * Db : p1 + Db;
            pa = a.actions + s * Da;
            have_a = true;
        } else {
            p1 = a.x + ((size_t)r * a.L + t) * 2 * Db;
            p2 = p1 + Db;
            if (a.y) { pa = a.y + ((size_t)r * a.L + t) * Da; have_a = true; }
        }
    }
    int ld_max = a.ld_te;
    if (a.ld_md > ld_max) ld_max = a.ld_md;
    if (a.ld_wm > ld_max) ld_max = a.ld_wm;
    if (a.ld_s2 > ld_max) ld_max = a.ld_s2;
    if (a.ld_a > ld_max) ld_max = a.ld_a;
    if (a.pr_in && a.ld_pr > ld_max) ld_max = a.ld_pr;
    constexpr int kFlags = 0x00020000, kNt = 2;         // raw buffer, 32-bit data; aux bit 1 = nontemporal
    const auto src = [](const float* p, int n) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, n * 4, kFlags); };
    const auto dst = [&](float* p, int ld) { return __builtin_amdgcn_make_buffer_rsrc(p ? p + prow * ld : a.te_in, 0, p ? ld * 4 : 0, kFlags); };
    const __amdgpu_buffer_rsrc_t rs1 = src(p1, valid && first ? Db : 0), rs2 = src(p2, valid ? Db : 0), ra = src(pa, have_a ? Da : 0);
    const __amdgpu_buffer_rsrc_t dte = dst(a.te_in, a.ld_te), dmd = dst(a.md_in, a.ld_md), dwm = dst(a.wm_in, a.ld_wm),
                                 dwp = dst(a.wm_pred, a.ld_wm), dpr = dst(a.pr_in, a.ld_pr), ds2 = dst(a.s2, a.ld_s2),
                                 dac = dst(a.act_t, a.ld_a);
    constexpr int KC = 8;                               // columns per lane and chunk
    for (int c0 = 0; c0 < ld_max; c0 += 64 * KC) {
        float v1[KC], v2[KC], vt[KC], vw[KC], va[KC];   // s1, s2 at the encoder's mapping; s2 at the target's; action at the world model's; action
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = c0 + 64 * k + lane;           // (offsets below zero wrap to far beyond any row: read as 0)
            v1[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, (unsigned)c * 4u, 0, kNt));
            v2[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, (unsigned)(c - Db) * 4u, 0, kNt));
            vt[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, (unsigned)c * 4u, 0, kNt));
            vw[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (unsigned)(c - Db) * 4u, 0, kNt));
            va[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (unsigned)c * 4u, 0, kNt));
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = c0 + 64 * k + lane;
            const unsigned off = (unsigned)c * 4u;
            const unsigned u1 = __builtin_bit_cast(unsigned, v1[k]);
            __builtin_amdgcn_raw_buffer_store_b32(u1, dpr, off, 0, 0);
            const float ute = (a.in_off & (c < Db ? 1 : 2)) ? 0.f : (c < Db ? v1[k] : v2[k]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ute), dte, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32((a.in_off & 4) ? 0u : u1, dmd, off, 0, 0);   // z columns filled by the sampler
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, c < Db ? v1[k] : vw[k]), dwm, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(u1, dwp, off, 0, 0);            // a_hat columns filled by the decoder
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, vt[k]), ds2, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, va[k]), dac, off, 0, 0);
        }
    }
}

// The same row by one wave THROUGH LDS (what the stand-alone gather launch runs when a row's sources fit the wave's
// 8 KB of LDS): the sources -- [s_t | s_{t+1}] and a_t, 4-byte aligned runs of dim_body / dim_action floats -- land in
// LDS by LDS-DMA (global_load_lds_dword: 256 contiguous bytes per instruction, no register, nothing waited for until
// all of the row is in flight: ceil((2 Db + Da) / 64) instructions), and every panel row is then written with whole
// 16-byte stores assembled from LDS (the misalignment between a source run and the 64-float-aligned panel is absorbed
// by the LDS reads, which cost nothing next to the memory system).  Per window at the configs[2] dims: 8 loads + 6
// stores instead of stage_row_wave's 40 + 56 dword-granular buffer instructions (most of them out of range), which
// made that form instruction-bound (16.2 us per 8192 windows against 13.3 for stage_row_vec).  Bit-identical (a copy).
constexpr int kStageLdsFloats = 2048;                   // per wave, at most (the launch sizes it to the row: stage_lds_floats)
__device__ inline void stage_row_lds(const StageArgs& a, int r, int t, int rows_pad, int lane, float* sl, int mask) {
    const int Db = a.Db, Da = a.Da;
    const size_t prow = (size_t)t * rows_pad + r;
    const bool valid = r < a.rows;
    const bool first = t == 0;
    bool have_a = false;
    if (valid) {
        const float* p1; const float* p2; const float* pa = nullptr;
        if (a.window_row) {
            const long long s = (long long)__builtin_amdgcn_readfirstlane(a.window_row[a.first_window + r]) + t;
            p1 = a.states + s * Db;
            p2 = a.next_states ? a.next_states + s * Db : p1 + Db;
            pa = a.actions + s * Da;
        } else {
            p1 = a.x + ((size_t)r * a.L + t) * 2 * Db;
            p2 = p1 + Db;
            if (a.y) pa = a.y + ((size_t)r * a.L + t) * Da;
        }
        have_a = pa != nullptr;
        // LDS image: [0, Db) s1, [Db, 2 Db) s2, [2 Db, 2 Db + Da) action; the tail of each 64-float DMA group that
        // reaches past a run lands in the next run's space (overwritten by that run's own DMA, issued later) or past
        // the end (never read).  Order: s1, s2, action.
        auto dma = [&](const float* src, int n, int dst0) {
            for (int c = 0; c < n; c += 64) {
                const int cc = c + lane < n ? c + lane : n - 1;          // (clamped: no read past the run)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + cc),
                                                 (__attribute__((address_space(3))) void*)(sl + dst0 + c), 4, 0, 2);
            }
        };
        if (p2 == p1 + Db) dma(p1, 2 * Db, 0);
        else { dma(p1, Db, 0); dma(p2, Db, Db); }
        if (have_a) dma(pa, Da, 2 * Db);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // value(c) = (LDS index, condition): the read is unconditional (any index inside the wave's region is safe), the
    // zero-fill a select -- no divergent branch per element
    auto put = [&](float* panel, int ld, auto&& where) {                 // 16 bytes per lane and trip
        if (!panel) return;
        float* row = panel + prow * ld;
        // (read j of lane l takes element (j + l / 8) & 3 of the lane's four: lanes l, l + 8, l + 16, l + 24 -- whose
        //  elements would share a bank, 4 l + e mod 32 -- hit four different banks; PMC: 70 % conflict cycles without)
        const int rot = (lane >> 3) & 3;
        for (int c = lane * 4; c < ld; c += 256) {
            v4f v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (j + rot) & 3;
                int idx; bool on;
                where(c + e, idx, on);
                const float raw = sl[idx & mask];                 // (unconditional: any masked index is inside the wave's image)
                const float x = on ? raw : 0.f;
                v[0] = e == 0 ? x : v[0];
                v[1] = e == 1 ? x : v[1];
                v[2] = e == 2 ? x : v[2];
                v[3] = e == 3 ? x : v[3];
            }
            *reinterpret_cast<v4f*>(row + c) = v;
        }
    };
    const bool s1_on = valid && first;
    const bool te_b = s1_on && !(a.in_off & 1), te_t = valid && !(a.in_off & 2), md_b = s1_on && !(a.in_off & 4);
    put(a.te_in, a.ld_te, [&](int c, int& i, bool& on) { i = c; on = c < Db ? te_b : (te_t && c < 2 * Db); });
    put(a.md_in, a.ld_md, [&](int c, int& i, bool& on) { i = c; on = md_b && c < Db; });           // z columns filled by the sampler
    put(a.wm_in, a.ld_wm, [&](int c, int& i, bool& on) { i = c < Db ? c : c + Db; on = c < Db ? s1_on : (have_a && c < Db + Da); });
    put(a.wm_pred, a.ld_wm, [&](int c, int& i, bool& on) { i = c; on = s1_on && c < Db; });        // a_hat columns filled by the decoder
    put(a.pr_in, a.ld_pr, [&](int c, int& i, bool& on) { i = c; on = s1_on && c < Db; });
    put(a.s2, a.ld_s2, [&](int c, int& i, bool& on) { i = Db + c; on = valid && c < Db; });
    put(a.act_t, a.ld_a, [&](int c, int& i, bool& on) { i = 2 * Db + c; on = have_a && c < Da; });
}

// host-side default for GemmArgs::krot: off (measured: no gain); PVAE_KROT=1 enables
// register sets in flight per lane in the register-staged kernels (A/B on the whole step: 2 / 2 /
// 4 beats 4 / 4 / 4 by 1.8 %, 3 and 8 lose; compile-time switches for re-measuring)
// register sets in flight per lane in the register-staged kernels (re-measured every round: 2 / 2 / 4 at 256 rows, 1 for the
// 64x32 input-gradient body of the >= 512-row pairs; profiles/r03_ab_regdepth_final.txt, r03_ab_regdepth_c5.txt)
constexpr int kRegDepthD = 2, kRegDepthD64 = 1, kRegDepth16 = 4, kRegDepthW = 2;
inline int g_krot = 0;                       // pvae_set_option(NULL, "krot", 1)
// Experiment (off by default): XCD x takes q-tile (row block) x of a 256-row layer and ALL its p-tiles, instead of
// all q-tiles of an eighth of the p-tiles -- the operand traffic a row-block-stationary multi-layer kernel would
// have (every XCD streams the whole weight matrix through its L2).  Same tiles, same results.
inline int g_rowxcd = 0;                     // pvae_set_option(NULL, "rowxcd", 1)
struct GemmArgs {
    const float* Q;
    int ldq;
    const float* P;
    int ldp;
    int K, tiles_q, tiles_p, p_per_xcd;
    int krot = g_krot;        // rotate each workgroup's K order (see k_rotation)
    int tile32 = 0;           // weight gradient on 32x32 output tiles (wgrad32_body) instead of 64x64
    int tile16 = 0;           // input gradient on 16x16 output tiles (splitk_reg16_body<false>) instead of 32x32
    int rowxcd = g_rowxcd;    // see g_rowxcd
};
// Kernel-argument preload (gfx950: the first user SGPRs are filled by the hardware at wave launch; hipcc -mllvm
// -amdgpu-kernarg-preload-count=16 in physicsvae_amd/build.py, at most 14 dwords, scalars and pointers only -- a
// struct passed by value is fetched by s_load inside the kernel, ~0.26 us on the critical path of every dependent
// launch: tools/kernarg_preload_probe.hip).  The contraction kernels therefore take the operands every workgroup
// needs before it can issue its first load as LEADING scalar parameters (11 dwords) and rebuild the GemmArgs from
// them; epilogue structs follow and arrive by s_load while the first tiles are in flight.
#define PVAE_GA_PARAMS(n) const float* n##Q, const float* n##P, int n##ldq, int n##ldp, int n##K, int n##tq, int n##tp, \
                          int n##ppx, int n##fl
#define PVAE_GA_PASS(g) (g).Q, (g).P, (g).ldq, (g).ldp, (g).K, (g).tiles_q, (g).tiles_p, (g).p_per_xcd, ga_flags(g)
#define PVAE_GA_OF(n) GemmArgs{n##Q, n##ldq, n##P, n##ldp, n##K, n##tq, n##tp, n##ppx, n##fl & 7, (n##fl >> 3) & 1, (n##fl >> 4) & 1, (n##fl >> 5) & 1}
// Launches that hold TWO contractions (input gradient || weight gradient, or the two trailing weight gradients) carry
// both sets of operands in the 14 preloadable dwords: four pointers, and per contraction the two row strides, the two
// tile counts (16 bits each) and the contraction length with the six switch bits (26 + 6 bits).  The workgroup
// counts of the two bodies follow from the tile counts (make_grid).  A/B of what is preloaded, joint step / config-5
// sizes: nothing 248.2 / 391.1 us, weight-gradient operands 246.4 / 388.7, input-gradient operands 244.4 / 385.5,
// both (this) -- profiles/r03_ab_kernarg_preload.txt.
#define PVAE_GA2_PARAMS const float* aQ, const float* aP, const float* bQ, const float* bP, unsigned a_ld, unsigned b_ld, \
                        unsigned a_t, unsigned b_t, unsigned a_k, unsigned b_k
#define PVAE_GA2_PASS(ga, gb) (ga).Q, (ga).P, (gb).Q, (gb).P, ga_pack_ld(ga), ga_pack_ld(gb), ga_pack_t(ga), ga_pack_t(gb), \
                              ga_pack_k(ga), ga_pack_k(gb)
#define PVAE_GA2_A ga_unpack(aQ, aP, a_ld, a_t, a_k)
#define PVAE_GA2_B ga_unpack(bQ, bP, b_ld, b_t, b_k)
// (krot: three bits -- the probes under tools/ use it as their ablation mode)
inline int ga_flags(const GemmArgs& g) { return (g.krot & 7) | ((g.tile32 & 1) << 3) | ((g.tile16 & 1) << 4) | ((g.rowxcd & 1) << 5); }
inline unsigned ga_pack_ld(const GemmArgs& g) { return (unsigned)g.ldq | ((unsigned)g.ldp << 16); }
inline unsigned ga_pack_t(const GemmArgs& g) { return (unsigned)g.tiles_q | ((unsigned)g.tiles_p << 16); }
inline unsigned ga_pack_k(const GemmArgs& g) { return (unsigned)g.K | ((unsigned)ga_flags(g) << 26); }
// what the packed form can carry (checked by the launchers; pvae_layout.h refuses stacks beyond it)
inline bool ga_packable(const GemmArgs& g) {
    return g.ldq >= 0 && g.ldq < 65536 && g.ldp >= 0 && g.ldp < 65536 && g.tiles_q >= 0 && g.tiles_q < 65536 &&
           g.tiles_p >= 0 && g.tiles_p < 65536 && g.K >= 0 && g.K < (1 << 26) && g.p_per_xcd == (g.tiles_p + 7) / 8;
}
__host__ __device__ inline int ga_grid(const GemmArgs& g) { return 8 * g.p_per_xcd * g.tiles_q; }
__device__ inline GemmArgs ga_unpack(const float* Q, const float* P, unsigned ld, unsigned t, unsigned k) {
    const int tp = (int)(t >> 16), fl = (int)(k >> 26);
    return GemmArgs{Q, (int)(ld & 0xffffu), P, (int)(ld >> 16), (int)(k & 0x3ffffffu), (int)(t & 0xffffu), tp, (tp + 7) / 8,
                    fl & 7, (fl >> 3) & 1, (fl >> 4) & 1, (fl >> 5) & 1};
}

// Experiment (off by default): start each workgroup at a different k-tile and wrap around, so that
// the workgroups of an XCD that share operand rows (same q-tile: X rows, same p-tile: W rows) do
// not walk K in lockstep -- a k-slab would be pulled into L2 by one workgroup and found there by the
// others a few tiles later.  Hypothesis: lockstep first-touch misses bound the tile rate.
// Measured: no change to slightly worse (hidden layer 8.44 -> 8.47 us, step 114 -> 117 us), i.e.
// the per-iteration rate is not miss-latency bound.  The summation order over K would differ per
// workgroup but stay a fixed function of the block index (deterministic).
__device__ inline int k_rotation(const GemmArgs& ga, int loc, int tile_q, int nk) {
    if (!ga.krot || nk < 2) return 0;
    const int pi = loc / ga.tiles_q;                       // position among this XCD's p-tiles
    int sq = nk / ga.p_per_xcd, sp = nk / ga.tiles_q;
    if (sq < 1) sq = 1;
    if (sp < 1) sp = 1;
    return (pi * sq + tile_q * sp) % nk;
}
constexpr int kRegRingFloats = 2 * 2 * 32 * 64;      // 2 LDS slots x (Q tile + P tile) = 32 KB

// Split-K partial tile of a wave -> its LDS reduction buffer red[32][RS], with 16-byte writes: the MFMA layout gives
// each lane runs of consecutive p (P_ROW: acc[a][b] = 4 consecutive p of row 16a + li; output-contiguous operands:
// for one a the eight values acc[a][0..1][0..3] are columns 8 lh .. 8 lh + 7).  RS = 36: the eight lanes of a
// ds_write_b128 group (li = 0..7) land on rows 36 dwords apart = all 32 banks once; the scalar writes this replaces
// hit 4-8 lanes per bank (PMC: 12-15 % of LDS cycles in the narrow launches of round 2).
template <bool P_ROW>
__device__ inline void store_partial_32x32(float* red, const v4f (&acc)[2][2], int li, int lh) {
    constexpr int RS = 36;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float* row = red + (16 * a + li) * RS;
        if (P_ROW) {
#pragma unroll
            for (int b = 0; b < 2; ++b) *reinterpret_cast<v4f*>(row + 16 * b + 4 * lh) = acc[a][b];
        } else {
            *reinterpret_cast<v4f*>(row + 8 * lh) = v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]};
            *reinterpret_cast<v4f*>(row + 8 * lh + 4) = v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]};
        }
    }
}

template <bool P_ROW, class Epi, int ABL = 0>
__device__ inline void splitk_reg_body(float* lds, int bid, const GemmArgs& ga, Epi& epi) {
    constexpr int BK = 64, kTile = 32 * 64, kStage = 2 * kTile, D = kRegDepthD, S = 2;
    static_assert(S * kStage >= 4 * 32 * 36, "ring must hold the split-K reduction buffer");
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;

    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 32, p0 = tile_p * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;

    const float* sq[2];
    const float* sp[2];
    int slot_off[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = (wave + 4 * u) * 64 + lane;
        slot_off[u] = j * 4;
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

    v4f rg[D][4];
    auto gload = [&](int t, v4f(&r)[4]) {
        r[0] = *reinterpret_cast<const v4f*>(sq[0] + (size_t)t * BK);
        r[1] = *reinterpret_cast<const v4f*>(sp[0] + (size_t)t * kstep_p);
        r[2] = *reinterpret_cast<const v4f*>(sq[1] + (size_t)t * BK);
        r[3] = *reinterpret_cast<const v4f*>(sp[1] + (size_t)t * kstep_p);
    };
    auto lwrite = [&](float* slot, const v4f(&r)[4]) {
        *reinterpret_cast<v4f*>(slot + slot_off[0]) = r[0];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[0]) = r[1];
        *reinterpret_cast<v4f*>(slot + slot_off[1]) = r[2];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[1]) = r[3];
    };

    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    int oq[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = 16 * a + li;
        oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
    }
    const int kq = 16 * wave + 4 * lh;

    const int nk = K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d < nk ? d : nk - 1, rg[d]);     // (clamped, not guarded: exact vmcnt)
    lwrite(lds, rg[0]);
    gload(D < nk ? D : nk - 1, rg[0]);
    const typename Epi::Pre epre = epi.preload(q0 + (tid >> 3), p0 + ((tid & 7) << 2));   // arrives under the loop
    __syncthreads();

    v4f abl_q = v4f{1.f, 2.f, 3.f, 4.f} * (float)(lane + 1);
    asm volatile("" : "+v"(abl_q));
    // one k-tile; see wgrad_reg_body for why the full groups are branch-free (exact vmcnt: the
    // three younger register sets stay in flight across the LDS write of the oldest)
    auto tile_step = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
        v4f fq[2], fp[2];
        v2f fc[4];
        if (ABL & 2) {            // (probes) loop-invariant fragments the compiler cannot see through
#pragma unroll
            for (int a = 0; a < 2; ++a) { fq[a] = abl_q; fp[a] = abl_q; }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) fc[s2] = v2f{abl_q[0], abl_q[1]};
            asm volatile("" : "+v"(fq[0]), "+v"(fq[1]), "+v"(fp[0]), "+v"(fp[1]));
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a) fq[a] = *reinterpret_cast<const v4f*>(st + oq[a]);
            if (P_ROW) {
#pragma unroll
                for (int b = 0; b < 2; ++b) fp[b] = *reinterpret_cast<const v4f*>(st + kTile + oq[b]);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    fc[s2] = *reinterpret_cast<const v2f*>(st + kTile + (((kq + s2) ^ (lh & 1)) * 32) + 2 * li);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float pv = P_ROW ? fp[b][s2] : fc[s2][b];
                    if (ABL & 4) acc[a][b][0] += pv * fq[a][s2];
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, fq[a][s2], acc[a][b], 0, 0, 0);
                }
            if (s2 == 1 && !(ABL & 1)) {
                // tile t+1 sits in register set (d+1)%D: park it in the other LDS slot and
                // refill the registers with tile t+1+D
                if (!guarded) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D]);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D]);
                }
            }
        }
        if (!(ABL & 8)) __syncthreads();
    };
    int t0 = 0;
    for (; t0 + D <= nk; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) tile_step(t0 + d, d, false);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (t0 + d < nk) tile_step(t0 + d, d, true);

    constexpr int RS = 36;
    store_partial_32x32<P_ROW>(lds + wave * (32 * RS), acc, li, lh);
    __syncthreads();
    {
        const int ql = tid >> 3, pl = (tid & 7) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (32 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epre);
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
}


// ---- input gradient on 64 x 32 output tiles, register-staged (the dgrad half of the fused pairs at >= 512 rows) ----
// Same contraction and the same LDS images / swizzles as splitk_reg_body<false>, but the workgroup owns 64 batch rows:
// per k-tile it stages a [64][64] dZ tile + a [64 k][32] W tile (24 KB) for the MFMA work of two 32x32 tiles (which
// stage 2 x 16 KB), and every P fragment read feeds four MFMA tiles instead of two -- at 512 rows the 32x32 form put
// 512 input-gradient workgroups beside the weight-gradient ones and the launch was bound by what they stage
// (config-5 sizes).  Each wave still owns one k-quarter of every k-tile and walks it in the same order, and
// the four partial tiles are summed in the same order, so results equal the 32x32 body's bit for bit.
constexpr int kReg64RingFloats = 2 * (64 * 64 + 64 * 32);          // 2 slots x 24 KB = 48 KB
template <class Epi>
__device__ inline void splitk_reg64_body(float* lds, int bid, const GemmArgs& ga, Epi& epi) {
    constexpr int BK = 64, kTileQ = 64 * 64, kTileP = 64 * 32, kStage = kTileQ + kTileP, D = kRegDepthD64;
    constexpr int RS = 36;
    static_assert(2 * kStage >= 4 * 64 * RS, "ring must hold the split-K reduction buffer");
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;
    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 64, p0 = tile_p * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;

    const float* sq[4];
    const float* sp[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int j = tid + 256 * u, row = j >> 4, c = (j & 15) ^ (row & 15);
        sq[u] = Q + (size_t)(q0 + row) * ldq + c * 4;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = tid + 256 * u, r = j >> 3, k = r ^ ((r >> 2) & 1);
        sp[u] = P + (size_t)k * ldp + p0 + (j & 7) * 4;
    }
    const size_t kstep_p = (size_t)BK * ldp;
    v4f rg[D][6];
    auto gload = [&](int t, v4f(&r)[6]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = *reinterpret_cast<const v4f*>(sq[u] + (size_t)t * BK);
#pragma unroll
        for (int u = 0; u < 2; ++u) r[4 + u] = *reinterpret_cast<const v4f*>(sp[u] + (size_t)t * kstep_p);
    };
    auto lwrite = [&](float* slot, const v4f(&r)[6]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<v4f*>(slot + (tid + 256 * u) * 4) = r[u];
#pragma unroll
        for (int u = 0; u < 2; ++u) *reinterpret_cast<v4f*>(slot + kTileQ + (tid + 256 * u) * 4) = r[4 + u];
    };
    v4f acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    int oq[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int row = 16 * a + li;
        oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
    }
    const int kq = 16 * wave + 4 * lh;
    const int nk = K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d < nk ? d : nk - 1, rg[d]);     // (clamped, not guarded: exact vmcnt)
    lwrite(lds, rg[0]);
    gload(D < nk ? D : nk - 1, rg[0]);
    typename Epi::Pre epre[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) epre[h] = epi.preload(q0 + 32 * h + (tid >> 3), p0 + ((tid & 7) << 2));
    __syncthreads();
    auto tile_step = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
        v4f fq[4];
        v2f fc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) fq[a] = *reinterpret_cast<const v4f*>(st + oq[a]);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
            fc[s2] = *reinterpret_cast<const v2f*>(st + kTileQ + (((kq + s2) ^ (lh & 1)) * 32) + 2 * li);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fc[s2][b], fq[a][s2], acc[a][b], 0, 0, 0);
            if (s2 == 1) {
                if (!guarded) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D]);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D]);
                }
            }
        }
        __syncthreads();
    };
    int t0 = 0;
    for (; t0 + D <= nk; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) tile_step(t0 + d, d, false);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (t0 + d < nk) tile_step(t0 + d, d, true);

    float* red = lds + wave * (64 * RS);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float* row = red + (16 * a + li) * RS;                       // (16-byte writes: see store_partial_32x32)
        *reinterpret_cast<v4f*>(row + 8 * lh) = v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]};
        *reinterpret_cast<v4f*>(row + 8 * lh + 4) = v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]};
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ql = 32 * h + (tid >> 3), pl = (tid & 7) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (64 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epre[h]);
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
}

template <bool P_ROW, class Epi, int ABL = 0>
__global__ void __launch_bounds__(256)
gemm_splitk_reg_kernel(PVAE_GA_PARAMS(a_), Epi epi) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[kRegRingFloats];
    splitk_reg_body<P_ROW, Epi, ABL>(lds, blockIdx.x, ga, epi);
}

// ---- forward / dgrad, wave-specialised LDS-DMA ring (512 threads) ----------------------------
// 8 waves = 4 compute + 4 loaders.  The loaders only issue global_load_lds_dwordx4 (tiles land in
// LDS without touching a VGPR) into a ring of kWsStages slots and certify "tile t+1 has landed"
// at barrier t; the compute waves do nothing but ds_read_b128/b64 + MFMA, with the fragments of
// tile t+1 read while the MFMAs of tile t run.  An LDS-DMA instruction costs ~150 issue cycles,
// which is why it must not sit in the instruction stream that feeds the matrix pipe (a 4-wave
// DMA ring measured 9.5 us for 256x1024x1024; this one 7.7 us; the register-staged loop 8.45).
// Ring depth: 4 is the optimum on MI355X -- 3 starves, 5..8 get progressively slower (8: 8.5 us);
// more requests in flight per CU do not help the L2 -> CU path, they hurt it.
// Same LDS image / swizzles / split-K layout as splitk_reg_body, results are bit-identical.
// Ring of the plain 32x32 kernel: SIX slots, super-steps of TWO k-tiles per workgroup barrier -- at the barrier in front of
// tiles (t0, t0+1) the tiles t0+1, t0+2 have landed, t0+3, t0+4 are in flight and t0+5, t0+6 are requested: half the
// barriers, a continuous load stream (joint step 243.7 -> 241.5 us).  The same on a 4-slot ring leaves nothing in flight
// across a barrier and loses 9 % (profiles/r03_ab_ws_superstep.txt).  The Pro variant (a patch needs its own barrier per
// patched tile) keeps one barrier per tile on four slots.
constexpr int kWsLoaders = 4, kWsThreads = 256 + 64 * kWsLoaders, kWsPer = 8 / kWsLoaders;   // DMA pairs per loader wave and tile
constexpr int kWsStages = 4;
constexpr int kWsFloats = kWsStages * 2 * 32 * 64;       // 64 KB

template <int N>
__device__ inline void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ inline void lds_dma16(const float* src, float* dst_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst_wave_base, 16, 0, 0);
}
// this wave's DMAs of a tile have landed once at most `younger` tiles (4 instructions each)
// issued after it are still in flight
__device__ inline void wait_dma_tile(int younger) {
    switch (younger) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<2 * kWsPer>(); break;
        default: wait_vmcnt<4 * kWsPer>(); break;          // kWsStages - 2 = 2 is the most that can be younger
    }
}

// ---------------------------------------------------------------------------------------
// First layers that read the demonstration set where it lies (the torch.cat sites tpv:377, rmt:829, 842
// as address arithmetic inside the loaders).  The input of a stack's first layer is a VIRTUAL matrix
//     X[q][k] = s0[row(q) * ld0 + k]                               k <  n0        (rows of `states`: s_t, or [s_t | s_{t+1}])
//             = s1[(ind1 ? row(q) : q) * ld1 + (k - n0)]      n0 <= k <  n0 + n1   (a_t rows of `actions`, or a z / a_hat panel)
//             = 0                                              elsewhere, and for q >= rows
// never materialised: the forward kernels' loaders fetch every 16-byte chunk of a Q tile from the source that holds it
// (LDS-DMA from 4-byte-aligned addresses lands at the aligned rate: tools/unaligned_probe.hip) and the weight-gradient
// bodies do the same through registers, zeroing what lies beyond the valid columns exactly.  The sources must be
// readable 16 bytes past their last element (pvae_bind_dataset checks the allocations).
// ---------------------------------------------------------------------------------------
// Which row of the demonstration set batch row q reads.  A minibatch is a run of consecutive windows, and window -> row is
// the identity plus a jump at every episode end: with at most one jump inside the minibatch the map is two (base, start)
// pairs that travel as kernel arguments (the host keeps a copy of `window_row` from bind time) -- no index load in front of
// the first tile fetch of the launch.  Otherwise: the index array itself.
struct RowMap {
    const int32_t* row;           // window_row + first_window (host side only: not read by the kernels)
    int seg, q1, b0, b1;          // rows q < q1 read row b0 + q, the others b1 + (q - q1); seg = 0: more than one jump --
                                  // such a minibatch (episodes shorter than the batch) takes the staging launch instead
    __device__ inline int at(int q) const { return q < q1 ? b0 + q : b1 + (q - q1); }   // (pure arithmetic: no load, no branch)
};
struct XSrc {
    const float* s0; int ld0, n0;
    const float* s1; int ld1, n1;
    int ind1;                     // 1: s1 is row-indirect like s0 (a_t rows of `actions`); 0: s1 row = batch row (a panel)
    RowMap rm;
    int rows;
    const float* zero;            // >= 256 zero bytes
};
// Q-operand policies of the wave-specialised forward kernels: where lane-slot (row q, 16-byte chunk at column c4) of
// k-tile kt comes from.  QDense: the panel Q[q][ldq] (every launch but a gathered first layer).
struct QDense {
    static constexpr bool kAlwaysFast = true;
    struct Base { const float* p; };
    __device__ inline Base base(const float* Q, int ldq, int q, int c4) const { return Base{Q + (size_t)q * ldq + c4}; }
    __device__ inline bool fast(int) const { return true; }
    __device__ inline const float* tile_fast(const Base& b, int kt) const { return b.p + (size_t)kt * 64; }
    __device__ inline const float* tile(const Base& b, int kt) const { return b.p + (size_t)kt * 64; }
};
// QGather: the virtual matrix above.  Its s0 run -- base pointer, row stride, width, the batch's rows as two (base, start)
// runs -- travels in the kernel's leading scalar arguments, which gfx950 preloads into SGPRs: the k-tiles that lie wholly
// inside [0, n0) (`fast`) are fetched without waiting for anything else.  The rest (`x`: the s1 block, the zero line, the
// index array for a minibatch with more than one episode jump) arrives by s_load while those tiles are in flight and is
// first touched by a tile that needs it.  A chunk that straddles n0 (n0 % 4 != 0) comes from s0 with the first floats of
// what follows the row in memory behind it: the launch's Pro patch overwrites those columns (n1 > 0 stacks), or they meet
// W's zero pad columns (n1 == 0); likewise the floats past n0 + n1.  Rows past the batch re-read the batch's last row
// (their outputs are never used and their gradient rows are zero).  Forward only: garbage x 0 = 0, a gradient would not.
struct QGather {
    static constexpr bool kAlwaysFast = false;
    const float* s0; int ld0, n0, rows, q1, b0, b1, seg;
    XSrc x;
    struct Base { const float* p0; int c4, q, r; };
    __device__ inline Base base(const float*, int, int q, int c4) const {
        Base b;
        b.q = q < rows ? q : rows - 1;
        b.r = b.q < q1 ? b0 + b.q : b1 + (b.q - q1);
        b.c4 = c4;
        b.p0 = s0 + (size_t)b.r * ld0 + c4;
        return b;
    }
    __device__ inline bool fast(int kt) const { return (kt + 1) * 64 <= n0; }              // (uniform: a scalar branch)
    __device__ inline const float* tile_fast(const Base& b, int kt) const { return b.p0 + kt * 64; }
    __device__ inline const float* tile(const Base& b, int kt) const {
        const int k0 = b.c4 + kt * 64, j = k0 - n0;
        if (k0 < n0) return b.p0 + kt * 64;
        return j < x.n1 ? x.s1 + (size_t)(x.ind1 ? b.r : b.q) * x.ld1 + j : x.zero + (b.c4 & 60);
    }
};
// the leading (preloadable) arguments of the gathered forward kernels: 14 dwords
#define PVAE_GG_PARAMS const float* g_s0, const float* g_P, int g_ld0n0, int g_ldp, int g_K, int g_tq, int g_tp, int g_ppx, int g_fl, \
                       int g_rq1, int g_b0, int g_b1
#define PVAE_GG_GA GemmArgs{nullptr, 0, g_P, g_ldp, g_K, g_tq, g_tp, g_ppx, g_fl & 7, (g_fl >> 3) & 1, (g_fl >> 4) & 1, (g_fl >> 5) & 1}
#define PVAE_GG_QS(xs) QGather{g_s0, g_ld0n0 & 0xffff, (int)((unsigned)g_ld0n0 >> 16), (g_rq1 & 0xffff) + 1, (int)((unsigned)g_rq1 >> 16) + 1, \
                               g_b0, g_b1, (g_fl >> 6) & 1, xs}

// Prologue hook of the wave-specialised kernel: a launch can OWN some columns of its Q operand -- form them itself
// instead of reading what a separate launch stored.  The compute waves `prepare` (request the inputs of those
// values for the workgroup's 32 rows: the loads travel while the first k-tiles are contracted) and `patch` the
// values over the tile image of every k-tile that holds such columns, right after that tile has landed and before
// its fragments are read (one extra workgroup barrier per patched tile; whatever the DMA brought for those columns
// is overwritten); `publish` runs behind that barrier.  NoPro: nothing of this exists in the instantiation.  The one user is the decoder's first layer, whose input [s1 | z] carries the
// sampler's z = mu + eps exp(logvar / 2) (ProSampler in pvae.hip: the sampler launch disappears).
struct NoPro {
    static constexpr bool kActive = false;
    static constexpr int kScratchFloats = 4;
    struct State {};
    __device__ inline bool needs(int) const { return false; }
    __device__ inline State prepare(int, int) const { return State(); }
    __device__ inline void patch(float*, float*, State&, int, int, int, int, int) const {}
    __device__ inline void publish(const float*, int, int, int, int) const {}
};

// Pro patch that COPIES columns [c0, c0 + n) of the Q tile from a second source (the world model's first layer on the
// gathered operand: [s_t | a_t] with a_t a row of `actions`, or [s_t | a_hat] with a_hat the decoder's output panel).
// The loaders bring the s_t columns straight from `states`; the chunk that straddles c0 (dim_body % 4 != 0) arrives with
// foreign floats behind s_t's last one, and this patch overwrites all n columns after the tile has landed.
struct ProCols {
    static constexpr bool kActive = true;
    static constexpr int kMaxN = 96, kScratchFloats = 4;
    static constexpr int kPer = 32 * kMaxN / 256;          // elements per thread at n = kMaxN
    const float* src; int ld, ind;                         // value (q, j) = src[(ind ? rm.at(q) : q) * ld + j]
    RowMap rm;
    int c0, n, rows;
    struct State { float v[kPer]; };
    __device__ inline bool needs(int t) const { return t >= (c0 >> 6) && t <= ((c0 + n - 1) >> 6); }
    __device__ inline State prepare(int q0, int tid) const {
        State st;
        const float* __restrict__ s_ = src;
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + 256 * u, r = e / n, j = e - r * n, q = q0 + r;
            const bool live = e < 32 * n && q < rows;
            st.v[u] = live ? s_[(size_t)(ind ? rm.at(q) : q) * ld + j] : 0.f;
        }
        return st;
    }
    __device__ inline void patch(float* tile, float*, State& st, int t, int, int, int, int tid) const {
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + 256 * u;
            if (e >= 32 * n) break;
            const int r = e / n, c = c0 + (e - r * n);
            if ((c >> 6) != t) continue;
            const int kc = c & 63;
            tile[r * 64 + (((kc >> 2) ^ (r & 15)) << 2) + (kc & 3)] = st.v[u];
        }
    }
    __device__ inline void publish(const float*, int, int, int, int) const {}
};

template <bool P_ROW, class Epi, class Pro = NoPro, class QS = QDense>
__device__ inline void splitk_ws_body(float* lds, int bid, const GemmArgs& ga, Epi& epi, const Pro& pro = Pro(),
                                      float* scratch = nullptr, const QS& qs = QS()) {
    // (plain kernel: six slots, two k-tiles per barrier; with a Pro patch: four slots, one barrier per tile)
    constexpr int BK = 64, kTile = 32 * 64, kStage = 2 * kTile, S = Pro::kActive ? kWsStages : 6;
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;
    const int xcd = bid & 7, loc = bid >> 3;
    int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    int tile_q = loc % tiles_q;
    if (ga.rowxcd && tiles_q == 8 && tiles_p == 8 * ga.p_per_xcd) { tile_q = xcd; tile_p = loc; }
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 32, p0 = tile_p * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4;
    const int nk = K / BK;
    PVAE_MARK(0, 0);                                             // workgroup entered
    typename Epi::Pre epre{};
    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    const size_t kstep_p = P_ROW ? (size_t)BK : (size_t)BK * ldp;
    const int rot = k_rotation(ga, loc, tile_q, nk);
    // k-tile 0 is fetched by ALL eight waves (one eighth of each operand image per wave, two DMA
    // instructions each): it is in flight ~0.1 us after the workgroup starts instead of queueing
    // behind the loaders' three-tile prologue (each global_load_lds holds its wave ~150 cycles, so
    // the prologue alone took 0.67 us -- tools/timeline_probe.hip).
    if (!PVAE_PROBE(6) && wave < 8) {
        const int j = wave * 64 + lane;                   // 16-byte slot inside the 8 KB tile image
        typename QS::Base s0q;
        const float* s0p;
        {
            const int row = j >> 4, c = (j & 15) ^ (row & 15);
            s0q = qs.base(Q, ldq, q0 + row, c * 4);
        }
        if (P_ROW) {
            const int row = j >> 4, c = (j & 15) ^ (row & 15);
            s0p = P + (size_t)(p0 + row) * ldp + c * 4;
        } else {
            const int r = j >> 3, k = r ^ ((r >> 2) & 1);
            s0p = P + (size_t)k * ldp + p0 + (j & 7) * 4;
        }
        if (QS::kAlwaysFast || qs.fast(rot)) lds_dma16(qs.tile_fast(s0q, rot), lds + wave * 256);
        else lds_dma16(qs.tile(s0q, rot), lds + wave * 256);
        lds_dma16(s0p + (size_t)rot * kstep_p, lds + kTile + wave * 256);
    }
    if (wave >= 4) {
        // ---------------- loader waves ----------------  (s_setprio 1 here, or on the compute waves: +-0)
        const int u0 = wave - 4;
        typename QS::Base sq[kWsPer];
        const float* sp[kWsPer];
#pragma unroll
        for (int u = 0; u < kWsPer; ++u) {
            const int j = (u0 + kWsLoaders * u) * 64 + lane;       // 16-byte slot inside the 8 KB tile image
            {
                const int row = j >> 4, c = (j & 15) ^ (row & 15);
                sq[u] = qs.base(Q, ldq, q0 + row, c * 4);
            }
            if (P_ROW) {
                const int row = j >> 4, c = (j & 15) ^ (row & 15);
                sp[u] = P + (size_t)(p0 + row) * ldp + c * 4;
            } else {
                const int r = j >> 3, k = r ^ ((r >> 2) & 1);
                sp[u] = P + (size_t)k * ldp + p0 + (j & 7) * 4;
            }
        }
        auto issue = [&](int t) {
            if (PVAE_PROBE(6)) return;                    // probe: loaders only keep the barriers
            float* slot = lds + (t % S) * kStage;
            int kt = t + rot;                             // k-tile this workgroup reads at step t
            if (kt >= nk) kt -= nk;
            if (PVAE_PROBE(2)) kt = 0;                    // probe: every step re-reads tile 0 (cache-resident)
            if (QS::kAlwaysFast || qs.fast(kt)) {
#pragma unroll
                for (int u = 0; u < kWsPer; ++u) {
                    lds_dma16(qs.tile_fast(sq[u], kt), slot + (u0 + kWsLoaders * u) * 256);
                    lds_dma16(sp[u] + (size_t)kt * kstep_p, slot + kTile + (u0 + kWsLoaders * u) * 256);
                }
            } else {                              // (a gathered first layer's last tiles: the second block, the zero line)
#pragma unroll
                for (int u = 0; u < kWsPer; ++u) {
                    lds_dma16(qs.tile(sq[u], kt), slot + (u0 + kWsLoaders * u) * 256);
                    lds_dma16(sp[u] + (size_t)kt * kstep_p, slot + kTile + (u0 + kWsLoaders * u) * 256);
                }
            }
        };
        if (1 < nk) issue(1);                                    // rides on tile 0's flight time
        PVAE_MARK(256, 4);
        if (1 < nk) wait_vmcnt<2 * kWsPer>(); else wait_vmcnt<0>();   // this wave's share of tile 0 landed
        PVAE_MARK(256, 5);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (Pro::kActive) {
            if (pro.needs(0)) __builtin_amdgcn_s_barrier();        // (the compute waves patch tile 0 in between)
        }
        if constexpr (!Pro::kActive) {
            // super-steps of TWO k-tiles per workgroup barrier.  At the barrier in front of tiles (t0, t0+1): t0+1 and t0+2
            // have landed, t0+3 and t0+4 are in flight; the slots of t0-1 and t0 (whose fragment reads the compute waves
            // have retired, see there) take t0+5 and t0+6.
            if (2 < nk) issue(2);
            if (3 < nk) issue(3);
            if (4 < nk) issue(4);
            for (int t0 = 0; t0 < nk; t0 += 2) {
                const int younger = nk - 3 - t0;                   // tiles t0+3, t0+4 that exist
                if (younger >= 2) wait_vmcnt<4 * kWsPer>();
                else if (younger == 1) wait_vmcnt<2 * kWsPer>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (t0 + 5 < nk) issue(t0 + 5);
                if (t0 + 6 < nk) issue(t0 + 6);
            }
        } else {
#pragma unroll
        for (int t = 2; t < S - 1; ++t)
            if (t < nk) issue(t);
        for (int t = 0; t < nk; ++t) {
            // tile t+1 landed: younger tiles in flight = t+2 .. min(t+S-2, nk-1)
            int y = nk - 2 - t;
            if (y > S - 3) y = S - 3;
            if (y < 0) y = 0;
            wait_dma_tile(y);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (Pro::kActive) {
                if (t + 1 < nk && pro.needs(t + 1)) __builtin_amdgcn_s_barrier();
            }
            if (t + S - 1 < nk) issue(t + S - 1);                  // refill the slot tile t-1 vacated
        }
        }
    } else {
        // ---------------- compute waves ----------------
        int oq[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int row = 16 * a + li;
            oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
        }
        const int kq = 16 * wave + 4 * lh;
        wait_vmcnt<0>();                                           // this wave's share of tile 0 landed
        epre = epi.preload(q0 + (tid >> 3), p0 + ((tid & 7) << 2));   // epilogue operands: arrive under the loop
        typename Pro::State pst = pro.prepare(q0, tid);
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
        auto mfmas = [&](const Frag& f) {
            if (PVAE_PROBE(4)) {                          // probe: no MFMAs (fragments stay live)
                asm volatile("" ::"v"(f.q[0]), "v"(f.q[1]), "v"(f.p[0]), "v"(f.p[1]));
                return;
            }
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
        __builtin_amdgcn_s_barrier();                              // tile 0 landed
        asm volatile("" ::: "memory");
        PVAE_MARK(0, 1);
        if constexpr (Pro::kActive) {
            if (pro.needs(0)) {
                pro.patch(lds, scratch, pst, 0, q0, tile_p, tile_q, tid);
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                pro.publish(scratch, 0, tile_p, tile_q, tid);
            }
        }
        Frag F0, F1;
        fread(lds, F0);
        if constexpr (!Pro::kActive) {
            for (int t0 = 0; t0 < nk; t0 += 2) {
                // The fragments of tile t0 (read in the second half of the previous super-step) must have LEFT the LDS before
                // this barrier: behind it the loaders restage that slot (tile t0+6) and the slot of t0-1 (tile t0+5), ONE
                // phase after their last reads -- and nothing orders an LDS-DMA write behind a ds_read that is still queued
                // (cdna_hip_programming.md: restage >= 2 phases after the last read, or 1 phase when an lgkmcnt retired the
                // reads first).  Alone on the chip the reads return in ~100 cycles and the DMA data arrives >= 1000 cycles
                // later; with other processes' workgroups on the same CU saturating its LDS port the window opens: round 5
                // saw ~1 wrong 16- or 32-row tile per 100 hidden-layer launches with 4-8 processes on one GPU
                // (tools/p2p_race_hunt.py; the one-barrier-per-tile ring restages two phases later and never showed it).
                // The wait is free: the reads were issued before the 16 MFMAs of the previous half-step.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // tiles t0+1 and t0+2 landed; slots of t0-1, t0 are free
                asm volatile("" ::: "memory");
                fread(lds + ((t0 + 1) % S) * kStage, F1);          // (past the end: stale slot, never used)
                __builtin_amdgcn_sched_barrier(0);
                mfmas(F0);
                if (t0 + 1 < nk) {
                    fread(lds + ((t0 + 2) % S) * kStage, F0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(F1);
                }
            }
        } else
        for (int t0 = 0; t0 < nk; t0 += 2) {
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int t = t0 + d;
                if (t < nk) {
                    Frag& F = d ? F1 : F0;
                    Frag& Gf = d ? F0 : F1;
                    __builtin_amdgcn_s_barrier();                  // tile t+1 landed; tile t-1's slot is free
                    asm volatile("" ::: "memory");
                    if constexpr (Pro::kActive) {
                        if (t + 1 < nk && pro.needs(t + 1)) {
                            pro.patch(lds + ((t + 1) % S) * kStage, scratch, pst, t + 1, q0, tile_p, tile_q, tid);
                            __builtin_amdgcn_s_barrier();
                            asm volatile("" ::: "memory");
                            pro.publish(scratch, t + 1, tile_p, tile_q, tid);
                        }
                    }
                    fread(lds + ((t + 1) % S) * kStage, Gf);       // (past the end: stale slot, never used)
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(F);
                }
            }
        }
    }
    // split-K reduction through LDS (fixed order: compute wave 0..3), epilogue on float4s by the
    // 256 compute threads; every wave takes part in the barriers
    PVAE_MARK(0, 2);                                             // main loop done (compute wave 0)
    __syncthreads();
    constexpr int RS = 36;
    if (wave < 4) store_partial_32x32<P_ROW>(lds + wave * (32 * RS), acc, li, lh);
    __syncthreads();
    if (wave < 4) {
        const int ql = tid >> 3, pl = (tid & 7) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (32 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epre);
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
    PVAE_MARK(0, 3);                                             // epilogue stores issued
}

// The same kernel on 64 x 32 output tiles (Q tile 64 rows, P tile 32 rows), for 512 rows and more: with 32x32 tiles a
// 512-row layer is 512 workgroups, two per CU, each landing its own 16 KB per k-tile through an LDS-DMA path that
// saturates at ~27 B/clk per CU -- the loop there IS loader-bound (not fetching X at all buys 1 us
// of 13).  One 64x32 workgroup per CU lands 24 KB per k-tile for the MFMA work of two 32x32 tiles (32 KB): a quarter
// less DMA traffic per flop.  Every output element is still the sum of the same four k-quarters in the same order, so
// results equal the 32x32 kernel's bit for bit.  Ring: 4 slots x 24 KB = 96 KB.
constexpr int kWs64Stages = 4;      // ring slots of the 64x32 kernel (24 KB each; 5 and 6 slots measured 2-4 % slower)
// PT = 32: the 64x32 tile above.  PT = 64 (round 4, 1024 rows and more): 64x64 outputs per workgroup, 32 KB per k-tile for
// twice the MFMA work of the 24 KB of 64x32 -- 16 flop per DMA byte instead of 10.7; the k-loop is as long as its DMA stream
// (docs/experiments.md), so that is what pays.  Ring 4 x 32 KB.  Same k-quarters per wave, same order: bit-identical again.
template <int PT> constexpr int ws64_floats() { return (PT == 64 ? 4 : kWs64Stages) * (64 + PT) * 64; }
template <bool P_ROW, class Epi, int PT = 32, class QS = QDense>
__device__ inline void splitk_ws64_body(float* lds, int bid, const GemmArgs& ga, Epi& epi, const QS& qs = QS()) {
    static_assert(PT == 32 || PT == 64, "P tile of 32 or 64 rows");
    constexpr int BK = 64, kTileQ = 64 * 64, kTileP = PT * 64, kStage = kTileQ + kTileP, S = PT == 64 ? 4 : kWs64Stages;
    constexpr int NB = PT / 16;                                  // 16-wide p blocks of the tile = P DMA instructions per loader wave
    constexpr int PER = 4 + NB;                                  // DMA instructions per loader wave and tile
    static_assert(kWsLoaders == 4, "written for four loader waves");
    static_assert(S >= 4 && S <= 6, "the loaders' wait ladder covers up to four tiles in flight");
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;
    const int xcd = bid & 7, loc = bid >> 3;
    int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    int tile_q = loc % tiles_q;
    if (PT == 64 && ga.rowxcd) {
        // Many row blocks (1024 rows and more): every XCD takes a RANGE OF ROW BLOCKS and walks all column blocks with it,
        // row blocks fastest -- its share of X (rq x 256 KB) stays in its L2 while the W tiles stream through once per
        // XCD.  The column-range mapping above re-reads all of X for every column block an XCD owns: at 4096 rows 256 MB
        // of fabric traffic per layer against 48 MB here.
        const int rq = (tiles_q + 7) >> 3;
        tile_q = xcd * rq + loc % rq;
        tile_p = loc / rq;
        if (tile_q >= tiles_q) return;
    }
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 64, p0 = tile_p * PT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lh = lane >> 4;
    const int nk = K / BK;
    typename Epi::Pre epre[NB] = {};
    v4f acc[4][NB];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    const size_t kstep_p = P_ROW ? (size_t)BK : (size_t)BK * ldp;
    // 16-byte slot j of an operand image -> its global source (same swizzles as splitk_ws_body)
    auto src_q = [&](int j) {
        const int row = j >> 4, c = (j & 15) ^ (row & 15);
        return qs.base(Q, ldq, q0 + row, c * 4);
    };
    auto src_p = [&](int j) {
        if (P_ROW) {
            const int row = j >> 4, c = (j & 15) ^ (row & 15);
            return P + (size_t)(p0 + row) * ldp + c * 4;
        }
        if (PT == 64) return P + (size_t)(j >> 4) * ldp + p0 + (j & 15) * 4;   // [64 k][64 p]: a k-row is all 64 banks, no swizzle
        const int r = j >> 3, k = r ^ ((r >> 2) & 1);
        return P + (size_t)k * ldp + p0 + (j & 7) * 4;
    };
    // k-tile 0 by all eight waves: Q image = 1024 slots (two per lane and wave), P image = 512 / 1024 (one / two)
    if (QS::kAlwaysFast || qs.fast(0)) {
        lds_dma16(qs.tile_fast(src_q(wave * 64 + lane), 0), lds + wave * 256);
        lds_dma16(qs.tile_fast(src_q((wave + 8) * 64 + lane), 0), lds + (wave + 8) * 256);
    } else {
        lds_dma16(qs.tile(src_q(wave * 64 + lane), 0), lds + wave * 256);
        lds_dma16(qs.tile(src_q((wave + 8) * 64 + lane), 0), lds + (wave + 8) * 256);
    }
    lds_dma16(src_p(wave * 64 + lane), lds + kTileQ + wave * 256);
    if (PT == 64) lds_dma16(src_p((wave + 8) * 64 + lane), lds + kTileQ + (wave + 8) * 256);
    if (wave >= 4) {
        // ---------------- loader waves: 4 Q + NB P instructions per tile ----------------
        const int u0 = wave - 4;
        typename QS::Base sq[4];
        const float* sp[NB];
#pragma unroll
        for (int u = 0; u < 4; ++u) sq[u] = src_q((u0 + 4 * u) * 64 + lane);
#pragma unroll
        for (int u = 0; u < NB; ++u) sp[u] = src_p((u0 + 4 * u) * 64 + lane);
        auto issue = [&](int t) {
            float* slot = lds + (t % S) * kStage;
            if (QS::kAlwaysFast || qs.fast(t)) {
#pragma unroll
                for (int u = 0; u < 4; ++u) lds_dma16(qs.tile_fast(sq[u], t), slot + (u0 + 4 * u) * 256);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) lds_dma16(qs.tile(sq[u], t), slot + (u0 + 4 * u) * 256);
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) lds_dma16(sp[u] + (size_t)t * kstep_p, slot + kTileQ + (u0 + 4 * u) * 256);
        };
        if (1 < nk) issue(1);
        if (1 < nk) wait_vmcnt<PER>(); else wait_vmcnt<0>();     // this wave's share of tile 0 landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 2; t < S - 1; ++t)
            if (t < nk) issue(t);
        for (int t = 0; t < nk; ++t) {
            int y = nk - 2 - t;                                  // tiles younger than t+1 still in flight
            if (y > S - 3) y = S - 3;
            if (y <= 0) wait_vmcnt<0>();
            else if (y == 1) wait_vmcnt<PER>();
            else if (y == 2) wait_vmcnt<2 * PER>();
            else wait_vmcnt<3 * PER>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + S - 1 < nk) issue(t + S - 1);
        }
    } else {
        // ---------------- compute waves: 64 x PT over this wave's k-quarter ----------------
        int oq[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int row = 16 * a + li;
            oq[a] = row * 64 + (((4 * wave + lh) ^ (row & 15)) << 2);
        }
        const int kq = 16 * wave + 4 * lh;
        wait_vmcnt<0>();                                         // this wave's share of tile 0 landed
#pragma unroll
        for (int h = 0; h < NB; ++h)
            epre[h] = PT == 64 ? epi.preload(q0 + 16 * h + (tid >> 4), p0 + ((tid & 15) << 2))
                               : epi.preload(q0 + 32 * h + (tid >> 3), p0 + ((tid & 7) << 2));
        struct Frag { v4f q[4], p[NB]; v2f c[4]; v4f c4[4]; };   // (P_ROW: p; P_COL: c at PT = 32, c4 at PT = 64)
        auto fread = [&](const float* st, Frag& f) {
#pragma unroll
            for (int a = 0; a < 4; ++a) f.q[a] = *reinterpret_cast<const v4f*>(st + oq[a]);
            if (P_ROW) {
#pragma unroll
                for (int b = 0; b < NB; ++b) f.p[b] = *reinterpret_cast<const v4f*>(st + kTileQ + oq[b]);
            } else if (PT == 64) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) f.c4[s2] = *reinterpret_cast<const v4f*>(st + kTileQ + (kq + s2) * 64 + 4 * li);
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    f.c[s2] = *reinterpret_cast<const v2f*>(st + kTileQ + (((kq + s2) ^ (lh & 1)) * 32) + 2 * li);
            }
        };
        auto mfmas = [&](const Frag& f) {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const float pv = P_ROW ? f.p[b][s2] : (PT == 64 ? f.c4[s2][b] : f.c[s2][b]);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(pv, f.q[a][s2], acc[a][b], 0, 0, 0);
                    }
        };
        __builtin_amdgcn_s_barrier();                            // tile 0 landed
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
                    __builtin_amdgcn_s_barrier();                // tile t+1 landed; tile t-1's slot is free
                    asm volatile("" ::: "memory");
                    fread(lds + ((t + 1) % S) * kStage, Gf);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(F);
                }
            }
        }
    }
    // split-K reduction through LDS (fixed order: compute wave 0..3), epilogue on float4s by the 256 compute threads
    __syncthreads();
    constexpr int RS = PT + 4;
    if (wave < 4) {
        float* red = lds + wave * (64 * RS);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            float* row = red + (16 * a + li) * RS;           // (16-byte writes: see store_partial_32x32)
            if (P_ROW) {
#pragma unroll
                for (int b = 0; b < NB; ++b) *reinterpret_cast<v4f*>(row + 16 * b + 4 * lh) = acc[a][b];
            } else if constexpr (PT == 64) {                 // lane holds p = 16 lh + 4 j + b of row 16 a + li
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<v4f*>(row + 16 * lh + 4 * j) = v4f{acc[a][0][j], acc[a][1][j], acc[a][2][j], acc[a][3][j]};
            } else {
                *reinterpret_cast<v4f*>(row + 8 * lh) = v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]};
                *reinterpret_cast<v4f*>(row + 8 * lh + 4) = v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]};
            }
        }
    }
    __syncthreads();
    if (wave < 4) {
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const int ql = PT == 64 ? 16 * h + (tid >> 4) : 32 * h + (tid >> 3);
            const int pl = PT == 64 ? (tid & 15) << 2 : (tid & 7) << 2;
            v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
            for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (64 * RS) + ql * RS + pl);
            epi(q0 + ql, p0 + pl, v, epre[h]);
        }
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
}

template <bool P_ROW, class Epi, int PT = 32>
__global__ void __launch_bounds__(512)
gemm_splitk_ws64_kernel(PVAE_GA_PARAMS(a_), Epi epi) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[ws64_floats<PT>()];
    splitk_ws64_body<P_ROW, Epi, PT>(lds, blockIdx.x, ga, epi);
}

template <bool P_ROW, class Epi>
__global__ void __launch_bounds__(kWsThreads)
gemm_splitk_ws_kernel(PVAE_GA_PARAMS(a_), Epi epi) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[6 * 2 * 32 * 64];
    splitk_ws_body<P_ROW, Epi>(lds, blockIdx.x, ga, epi);
}
template <class Epi, class Pro>
__global__ void __launch_bounds__(kWsThreads)
gemm_splitk_ws_pro_kernel(PVAE_GA_PARAMS(a_), Epi epi, Pro pro) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[kWsFloats];
    __shared__ __attribute__((aligned(16))) float scratch[Pro::kScratchFloats];
    splitk_ws_body<true, Epi, Pro>(lds, blockIdx.x, ga, epi, pro, scratch);
}
// the first layer of a stack on the gathered operand (XSrc): 32x32 tiles, plain or with a Pro patch; 64 x PT tiles
template <class Epi>
__global__ void __launch_bounds__(kWsThreads)
gemm_splitk_ws_gather_kernel(PVAE_GG_PARAMS, Epi epi, XSrc xs) {
    const GemmArgs ga = PVAE_GG_GA;
    __shared__ __attribute__((aligned(16))) float lds[6 * 2 * 32 * 64];
    splitk_ws_body<true, Epi, NoPro, QGather>(lds, blockIdx.x, ga, epi, NoPro(), nullptr, PVAE_GG_QS(xs));
}
template <class Epi, class Pro>
__global__ void __launch_bounds__(kWsThreads)
gemm_splitk_ws_pro_gather_kernel(PVAE_GG_PARAMS, Epi epi, Pro pro, XSrc xs) {
    const GemmArgs ga = PVAE_GG_GA;
    __shared__ __attribute__((aligned(16))) float lds[kWsFloats];
    __shared__ __attribute__((aligned(16))) float scratch[Pro::kScratchFloats];
    splitk_ws_body<true, Epi, Pro, QGather>(lds, blockIdx.x, ga, epi, pro, scratch, PVAE_GG_QS(xs));
}
template <class Epi, int PT>
__global__ void __launch_bounds__(512)
gemm_splitk_ws64_gather_kernel(PVAE_GG_PARAMS, Epi epi, XSrc xs) {
    const GemmArgs ga = PVAE_GG_GA;
    __shared__ __attribute__((aligned(16))) float lds[ws64_floats<PT>()];
    splitk_ws64_body<true, Epi, PT, QGather>(lds, blockIdx.x, ga, epi, PVAE_GG_QS(xs));
}

// ---- forward, 16x16 tile per workgroup (narrow output layers) ----------------------------------
// An output layer with few features (197 / 64 / 45 -> 8 or fewer 32-wide p-tiles) gives the 32x32
// kernel under 128 workgroups: most CUs idle for a full K = 1024 pass.  Same algorithm on a
// 16x16 tile (one MFMA tile per wave, two accumulators alternating over the k-steps so the
// dependent-issue latency is covered; the 4 waves still split K): 4x the workgroups, each
// streaming (16 + 16) rows.  4 flop/B, so it only pays when the grid would otherwise be small.
// P_ROW = false is the input-gradient form (P = W[k][p], p contiguous) for first-layer gradients:
// 256 x 256 outputs are 64 tiles of 32x32 that run a full K = 1024 pass on a quarter of the CUs
// (9.5 us), 256 of these.  Its P image is [64 k][16 p] row-major, read one dword per k-step.
template <bool P_ROW, class Epi>
__device__ inline void splitk_reg16_body(float* lds, int bid, const GemmArgs& ga, Epi& epi) {
    constexpr int BK = 64, kTile = 16 * 64, kStage = 2 * kTile, D = kRegDepth16;
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;

    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 16, p0 = tile_p * 16;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;

    // one 16-byte chunk of each operand tile per thread: slot j = tid, row j>>4, chunk (j&15)^row
    const int srow = tid >> 4, schunk = (tid & 15) ^ srow;
    const float* sq = Q + (size_t)(q0 + srow) * ldq + schunk * 4;
    const float* sp = P_ROW ? P + (size_t)(p0 + srow) * ldp + schunk * 4
                            : P + (size_t)(tid >> 2) * ldp + p0 + (tid & 3) * 4;      // k = tid/4, 4 chunks per row
    const size_t kstep_p = P_ROW ? (size_t)BK : (size_t)BK * ldp;
    const int slot_off = tid * 4;
    // input-gradient form: P image rows k are kept at row k ^ ((k >> 2) & 1).  A fragment read touches rows
    // 4 lh + s2 (16 dwords each): rows 4 apart would otherwise fall on the same 16 banks for both halves of a
    // 32-lane read group (2-way conflict on every k-step: 15 % of the LDS cycles of the sampler-seed launch)
    const int pk = tid >> 2;
    const int slot_off_p = P_ROW ? slot_off : ((pk ^ ((pk >> 2) & 1)) * 16 + (tid & 3) * 4);

    v4f rg[D][2];
    auto gload = [&](int t, v4f(&r)[2]) {
        r[0] = *reinterpret_cast<const v4f*>(sq + (size_t)t * BK);
        r[1] = *reinterpret_cast<const v4f*>(sp + (size_t)t * kstep_p);
    };
    auto lwrite = [&](float* slot, const v4f(&r)[2]) {
        *reinterpret_cast<v4f*>(slot + slot_off) = r[0];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off_p) = r[1];
    };
    v4f acc[2] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    const int of = li * 64 + (((4 * wave + lh) ^ li) << 2);

    const int nk = K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d < nk ? d : nk - 1, rg[d]);     // (clamped, not guarded: exact vmcnt)
    lwrite(lds, rg[0]);
    gload(D < nk ? D : nk - 1, rg[0]);
    __syncthreads();

    // (full groups of D tiles are branch-free so that the compiler counts the outstanding loads
    // exactly -- see wgrad_reg_body)
    auto tile_step = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
        const v4f fq = *reinterpret_cast<const v4f*>(st + of);
        v4f fp;
        if (P_ROW) {
            fp = *reinterpret_cast<const v4f*>(st + kTile + of);
        } else {
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) fp[s2] = st[kTile + ((16 * wave + 4 * lh + s2) ^ (lh & 1)) * 16 + li];
        }
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            acc[s2 & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[s2], fq[s2], acc[s2 & 1], 0, 0, 0);
            if (s2 == 1) {
                if (!guarded) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D]);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D]);
                }
            }
        }
        __syncthreads();
    };
    int t0 = 0;
    for (; t0 + D <= nk; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) tile_step(t0 + d, d, false);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (t0 + d < nk) tile_step(t0 + d, d, true);

    // split-K reduction (fixed order) + epilogue: 64 threads x float4 cover the 16x16 tile
    constexpr int RS = 20;
    float* red = lds + wave * (16 * RS);
    const v4f mine = acc[0] + acc[1];
    *reinterpret_cast<v4f*>(red + li * RS + 4 * lh) = mine;
    __syncthreads();
    if (tid < 64) {
        const int ql = tid >> 2, pl = (tid & 3) << 2;
        v4f v = *reinterpret_cast<const v4f*>(lds + ql * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (16 * RS) + ql * RS + pl);
        epi(q0 + ql, p0 + pl, v, epi.preload(q0 + ql, p0 + pl));
    }
    epi.finish(lds, tile_q * tiles_p + tile_p, tid);
}

template <bool P_ROW, class Epi>
__global__ void __launch_bounds__(256)
gemm_splitk_reg16_kernel(PVAE_GA_PARAMS(a_), Epi epi) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * 16 * 64];
    splitk_reg16_body<P_ROW, Epi>(lds, blockIdx.x, ga, epi);
}

// ---- the gathered operand (XSrc) as the P = X operand of a weight-gradient body --------------------------------
// A lane owns U 16-byte chunks per k-tile: columns k0 .. k0 + 3 of batch row t * BKR + row.  Every chunk is fetched with
// TWO loads -- from the s0 row and from the s1 row (a zero line where the chunk has nothing from that source) -- and the
// elements are selected by masks that do not depend on the tile: branch-free, so hipcc's wait counts stay exact, and
// what lies beyond the valid columns is exactly zero (a gradient, unlike a forward product, must not see garbage there:
// it would move W's pad columns off zero).  The chunk that straddles n0 takes its tail from the START of the s1 row,
// shifted by n0 - k0 elements.  Row indices for the load D calls ahead are fetched BEFORE this call's data, so waiting
// for them never waits for younger data (loads retire in order).
struct PDense { static constexpr bool kGather = false; };
struct PGather { static constexpr bool kGather = true; XSrc x; };
template <int U, int BKR, int D>
struct XLanes {
    int k0[U], row[U];            // (everything else is recomputed per load from these and the wave-uniform descriptor:
                                  //  the weight-gradient bodies sit at the edge of three waves per SIMD)
    __device__ inline void set(const XSrc&, int u, int row_, int k0_) { k0[u] = k0_; row[u] = row_; }
    __device__ inline void fetch(const XSrc&, int, int) {}
    // the chunks of tile t (t_next, s: unused -- the row map is arithmetic for a minibatch with at most one episode jump;
    // otherwise an index load precedes the data, which only the rare minibatch with several jumps pays)
    __device__ inline void load(const XSrc& x, int t, int, int, v4f (&out)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int rr = t * BKR + row[u], k = k0[u];
            const bool valid = rr < x.rows;
            const int rc = valid ? rr : x.rows - 1;
            const int i0 = x.rm.at(rc), i1 = x.ind1 ? i0 : rc;
            const bool has0 = k < x.n0, has1 = k + 3 >= x.n0 && k < x.n0 + x.n1;
            const int d = has0 ? x.n0 - k : 0;                             // 1 .. 3 in the chunk that straddles n0, else 0 or >= 4
            // (always an address inside the row: what a source does not contribute, and every row past the batch, is masked below)
            const float* pa = x.s0 + (size_t)i0 * x.ld0 + (has0 ? k : 0);
            const float* pb = x.s1 + (size_t)i1 * x.ld1 + (has1 && k > x.n0 ? k - x.n0 : 0);
            const v4f a = *reinterpret_cast<const v4f*>(pa);
            const v4f w = *reinterpret_cast<const v4f*>(pb);
            v4f ws;                                                        // w shifted right by d elements (d < 4)
            ws[0] = d == 0 ? w[0] : 0.f;
            ws[1] = d == 0 ? w[1] : (d == 1 ? w[0] : 0.f);
            ws[2] = d == 0 ? w[2] : (d == 1 ? w[1] : (d == 2 ? w[0] : 0.f));
            ws[3] = d == 0 ? w[3] : (d == 1 ? w[2] : (d == 2 ? w[1] : w[0]));
#pragma unroll
            for (int e = 0; e < 4; ++e)
                out[u][e] = !valid ? 0.f : (k + e < x.n0 ? a[e] : (k + e < x.n0 + x.n1 ? ws[e] : 0.f));
        }
    }
};

// ---- wgrad: G[64 q][64 p] per workgroup, reduction over batch rows (BK = 32), both operands COL,
// waves 2x2 with 32x32 each; the epilogue operands (Adam's p, m, v) are fetched under the loop --
template <class Epi, int ABL = 0, class PS = PDense>
__device__ inline void wgrad_reg_body(float* lds, int bid, const GemmArgs& ga, Epi& epi, const PS& ps = PS()) {
    constexpr int BK = 32, kTile = 32 * 64, kStage = 2 * kTile, D = kRegDepthW, S = 2;
    static_assert(S * kStage == kRegRingFloats, "LDS budget");
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;

    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 64, p0 = tile_p * 64;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;
    const int wq = (wave >> 1) * 32, wp = (wave & 1) * 32;

    const float* sq[2];
    const float* sp[2];
    int slot_off[2];
    XLanes<2, BK, D> xl;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = (wave + 4 * u) * 64 + lane;
        slot_off[u] = j * 4;
        const int row = j >> 4, c = (j & 15) ^ ((row & 1) << 3);
        sq[u] = Q + (size_t)row * ldq + q0 + c * 4;
        sp[u] = P + (size_t)row * ldp + p0 + c * 4;
        if constexpr (PS::kGather) xl.set(ps.x, u, row, p0 + c * 4);
    }
    const int nk_ = K / BK;
    if constexpr (PS::kGather) {
#pragma unroll
        for (int d = 0; d < D; ++d) xl.fetch(ps.x, d < nk_ ? d : nk_ - 1, d);
    }
    v4f rg[D][4];
    auto gload = [&](int t, v4f(&r)[4], int s_ = 0) {
        if constexpr (PS::kGather) {
            v4f o[2];
            r[0] = *reinterpret_cast<const v4f*>(sq[0] + (size_t)t * BK * ldq);
            r[2] = *reinterpret_cast<const v4f*>(sq[1] + (size_t)t * BK * ldq);
            xl.load(ps.x, t, t + D < nk_ ? t + D : nk_ - 1, s_, o);
            r[1] = o[0]; r[3] = o[1];
        } else {
            r[0] = *reinterpret_cast<const v4f*>(sq[0] + (size_t)t * BK * ldq);
            r[1] = *reinterpret_cast<const v4f*>(sp[0] + (size_t)t * BK * ldp);
            r[2] = *reinterpret_cast<const v4f*>(sq[1] + (size_t)t * BK * ldq);
            r[3] = *reinterpret_cast<const v4f*>(sp[1] + (size_t)t * BK * ldp);
        }
    };
    auto lwrite = [&](float* slot, const v4f(&r)[4]) {
        *reinterpret_cast<v4f*>(slot + slot_off[0]) = r[0];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[0]) = r[1];
        *reinterpret_cast<v4f*>(slot + slot_off[1]) = r[2];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[1]) = r[3];
    };

    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    const int sw = (lh & 1) << 3;
    const int cq = wq + 2 * li, cp = wp + 2 * li;
    const int oq = lh * 64 + ((((cq >> 2) ^ sw)) << 2) + (cq & 3);
    const int op = kTile + lh * 64 + ((((cp >> 2) ^ sw)) << 2) + (cp & 3);

    const int nk = K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d < nk ? d : nk - 1, rg[d], d);     // (clamped, not guarded: exact vmcnt)
    lwrite(lds, rg[0]);
    gload(D < nk ? D : nk - 1, rg[0], 0);
    // epilogue operands: queued behind the first D tiles, they arrive while the loop runs
    typename Epi::Pre pre[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        pre[a][0] = epi.load(q0 + wq + 2 * li + a, p0 + wp + 8 * lh);
        pre[a][1] = epi.load(q0 + wq + 2 * li + a, p0 + wp + 8 * lh + 4);
    }
    __syncthreads();

    // one k-tile: 8 steps of (2 fragment reads, 4 MFMAs); after step 3 the next tile moves from its
    // register set into the other LDS slot and that set is refilled D tiles ahead
    // fragments of k-step kk+4 are requested before the MFMAs of k-step kk are issued: inside a tile
    // only the first read's latency is exposed (left to itself hipcc reads two k-steps with one
    // ds_read2st64_b64, waits for them, issues 8 MFMAs, and repeats: 4 exposed waits per tile;
    // fused backward launch 15.1 -> 14.8 us)
    auto tile_step_piped = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
        v2f fq[2], fp[2];
        fq[0] = *reinterpret_cast<const v2f*>(st + oq);
        fp[0] = *reinterpret_cast<const v2f*>(st + op);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            const int c = (kk >> 2) & 1, n = c ^ 1;
            if (kk + 4 < BK) {
                fq[n] = *reinterpret_cast<const v2f*>(st + oq + (kk + 4) * 64);
                fp[n] = *reinterpret_cast<const v2f*>(st + op + (kk + 4) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[c][b], fq[c][a], acc[a][b], 0, 0, 0);
            if (kk == 12) {
                if (!guarded) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D], (d + 1) % D);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D], (d + 1) % D);
                }
            }
        }
        __syncthreads();
    };
    v2f abl_f = v2f{1.f, 2.f} * (float)(lane + 1);
    asm volatile("" : "+v"(abl_f));
    auto tile_step_plain = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 4) {
            v2f fq, fp;
            if (ABL & 2) {        // (probes) loop-invariant fragments the compiler cannot see through
                fq = abl_f;
                fp = abl_f;
                asm volatile("" : "+v"(fq), "+v"(fp));
            } else {
                fq = *reinterpret_cast<const v2f*>(st + oq + kk * 64);
                fp = *reinterpret_cast<const v2f*>(st + op + kk * 64);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (ABL & 4) acc[a][b][0] += fp[b] * fq[a];
                    else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[b], fq[a], acc[a][b], 0, 0, 0);
                }
            if (kk == 12 && !(ABL & 1)) {
                if (!guarded) {
                    // branch-free: past the end this stores a stale register set into the idle slot
                    // and re-reads the last tile -- both unused.  Without branches the compiler counts
                    // the outstanding loads exactly (vmcnt(12): three younger sets stay in flight);
                    // with the guards it drained the whole prefetch queue before every LDS write.
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D], (d + 1) % D);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D], (d + 1) % D);
                }
            }
        }
        if (!(ABL & 8)) __syncthreads();
    };
    auto tile_step = [&](int t, int d, bool guarded) {
        if constexpr (ABL == 0) tile_step_piped(t, d, guarded);
        else tile_step_plain(t, d, guarded);
    };
    int t0 = 0;
    for (; t0 + D <= nk; t0 += D) {            // full groups of D tiles: no branch inside
#pragma unroll
        for (int d = 0; d < D; ++d) tile_step(t0 + d, d, false);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)                // remaining nk % D tiles
        if (t0 + d < nk) tile_step(t0 + d, d, true);

    PVAE_MARK(0, 2);                                             // contraction done, epilogue (Adam) starts
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int q = q0 + wq + 2 * li + a, p = p0 + wp + 8 * lh;
        epi.apply(q, p, v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]}, pre[a][0]);
        epi.apply(q, p + 4, v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]}, pre[a][1]);
    }
    if (epi.loss.out && bid == 0 && wave == 3) finalize_loss_wave(epi.loss, lane);
}

// ---- weight gradient, 32x32 output tile per workgroup ------------------------------------------
// A weight matrix with few 64x64 tiles (first / last layers: 1024x256 -> 64 tiles) occupies a
// quarter of the CUs with workgroups that run as long as those of a full 1024x1024 layer; in a fused
// launch they decide its length (layer-1 + layer-0 gradients: 16.7 us, layer 1 alone 11.2).  On 32x32
// tiles the same matrix gives four times the workgroups, each a quarter as long: the four waves
// split every 64-row k-tile (16 rows = 4 MFMA k-steps each) over the whole tile and reduce through
// LDS at the end, like the input-gradient body.  Needs K % 64 == 0 (wgrad_uses_32x32).
template <class Epi, class PS = PDense>
__device__ inline void wgrad32_body(float* lds, int bid, const GemmArgs& ga, Epi& epi, const PS& ps = PS()) {
    constexpr int BK = 64, kTile = 64 * 32, kStage = 2 * kTile, D = kRegDepthW, S = 2;
    static_assert(S * kStage == kRegRingFloats, "LDS budget");
    static_assert(S * kStage >= 4 * 32 * 36, "ring must hold the split-K reduction buffer");
    const float* __restrict__ Q = ga.Q;
    const float* __restrict__ P = ga.P;
    const int ldq = ga.ldq, ldp = ga.ldp, K = ga.K, tiles_q = ga.tiles_q, tiles_p = ga.tiles_p;

    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * ga.p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * 32, p0 = tile_p * 32;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;

    // operand images in LDS: [64 k-rows][32 columns], row-major (fragment reads and 16-byte writes
    // are conflict-free as they are); 512 16-byte chunks per operand, two per thread
    const float* sq[2];
    const float* sp[2];
    int slot_off[2];
    XLanes<2, BK, D> xl;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = tid + 256 * u;
        const int row = j >> 3, c = j & 7;
        slot_off[u] = j * 4;
        sq[u] = Q + (size_t)row * ldq + q0 + c * 4;
        sp[u] = P + (size_t)row * ldp + p0 + c * 4;
        if constexpr (PS::kGather) xl.set(ps.x, u, row, p0 + c * 4);
    }
    const int nk_ = K / BK;
    if constexpr (PS::kGather) {
#pragma unroll
        for (int d = 0; d < D; ++d) xl.fetch(ps.x, d < nk_ ? d : nk_ - 1, d);
    }
    v4f rg[D][4];
    auto gload = [&](int t, v4f(&r)[4], int s_ = 0) {
        if constexpr (PS::kGather) {
            v4f o[2];
            r[0] = *reinterpret_cast<const v4f*>(sq[0] + (size_t)t * BK * ldq);
            r[2] = *reinterpret_cast<const v4f*>(sq[1] + (size_t)t * BK * ldq);
            xl.load(ps.x, t, t + D < nk_ ? t + D : nk_ - 1, s_, o);
            r[1] = o[0]; r[3] = o[1];
        } else {
            r[0] = *reinterpret_cast<const v4f*>(sq[0] + (size_t)t * BK * ldq);
            r[1] = *reinterpret_cast<const v4f*>(sp[0] + (size_t)t * BK * ldp);
            r[2] = *reinterpret_cast<const v4f*>(sq[1] + (size_t)t * BK * ldq);
            r[3] = *reinterpret_cast<const v4f*>(sp[1] + (size_t)t * BK * ldp);
        }
    };
    auto lwrite = [&](float* slot, const v4f(&r)[4]) {
        *reinterpret_cast<v4f*>(slot + slot_off[0]) = r[0];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[0]) = r[1];
        *reinterpret_cast<v4f*>(slot + slot_off[1]) = r[2];
        *reinterpret_cast<v4f*>(slot + kTile + slot_off[1]) = r[3];
    };

    v4f acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = v4f{0.f, 0.f, 0.f, 0.f};

    const int of = (16 * wave + lh) * 32 + 2 * li;       // this lane's fragment pair at k-step 0 of its slice

    const int nk = K / BK;
#pragma unroll
    for (int d = 0; d < D; ++d) gload(d < nk ? d : nk - 1, rg[d], d);     // (clamped, not guarded: exact vmcnt)
    lwrite(lds, rg[0]);
    gload(D < nk ? D : nk - 1, rg[0], 0);
    const typename Epi::Pre pre = epi.load(q0 + (tid >> 3), p0 + ((tid & 7) << 2));   // arrives under the loop
    __syncthreads();

    auto tile_step = [&](int t, int d, bool guarded) {
        const float* st = lds + (t & 1) * kStage;
        v2f fq[2], fp[2];
        fq[0] = *reinterpret_cast<const v2f*>(st + of);
        fp[0] = *reinterpret_cast<const v2f*>(st + kTile + of);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int c = s2 & 1, n = c ^ 1;
            if (s2 + 1 < 4) {
                fq[n] = *reinterpret_cast<const v2f*>(st + of + (s2 + 1) * 128);
                fp[n] = *reinterpret_cast<const v2f*>(st + kTile + of + (s2 + 1) * 128);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[c][b], fq[c][a], acc[a][b], 0, 0, 0);
            if (s2 == 1) {
                if (!guarded) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    const int tn = t + 1 + D < nk ? t + 1 + D : nk - 1;
                    gload(tn, rg[(d + 1) % D], (d + 1) % D);
                } else if (t + 1 < nk) {
                    lwrite(lds + ((t + 1) & 1) * kStage, rg[(d + 1) % D]);
                    if (t + 1 + D < nk) gload(t + 1 + D, rg[(d + 1) % D], (d + 1) % D);
                }
            }
        }
        __syncthreads();
    };
    int t0 = 0;
    for (; t0 + D <= nk; t0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) tile_step(t0 + d, d, false);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (t0 + d < nk) tile_step(t0 + d, d, true);

    // split-K reduction over the four waves (fixed order), epilogue on float4s by all 256 threads
    // (lane (li, lh) holds, for a = 0, 1, the eight columns 8 lh .. 8 lh + 7 of output row 2 li + a.  Row 2 li + a is
    //  kept at LDS row 16 a + li, so that the eight lanes of a 16-byte write group are 36 dwords apart: every bank once)
    constexpr int RS = 36;
    float* red = lds + wave * (32 * RS);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float* row = red + (16 * a + li) * RS;
        *reinterpret_cast<v4f*>(row + 8 * lh) = v4f{acc[a][0][0], acc[a][1][0], acc[a][0][1], acc[a][1][1]};
        *reinterpret_cast<v4f*>(row + 8 * lh + 4) = v4f{acc[a][0][2], acc[a][1][2], acc[a][0][3], acc[a][1][3]};
    }
    __syncthreads();
    {
        const int ql = tid >> 3, pl = (tid & 7) << 2;
        const int qr = 16 * (ql & 1) + (ql >> 1);                   // LDS row of output row ql
        v4f v = *reinterpret_cast<const v4f*>(lds + qr * RS + pl);
#pragma unroll
        for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const v4f*>(lds + w * (32 * RS) + qr * RS + pl);
        epi.apply(q0 + ql, p0 + pl, v, pre);
    }
    if (epi.loss.out && bid == 0 && wave == 3) finalize_loss_wave(epi.loss, lane);
}

// either geometry, chosen per problem on the host
constexpr int kPairLdsFloats = kRegRingFloats;
template <class Epi, int ABL = 0, class PS = PDense>
__device__ inline void wgrad_body(float* lds, int bid, const GemmArgs& ga, Epi& epi, const PS& ps = PS()) {
    if (ga.tile32) wgrad32_body<Epi, PS>(lds, bid, ga, epi, ps);
    else wgrad_reg_body<Epi, ABL, PS>(lds, bid, ga, epi, ps);
}

struct AdamScalars {
    float step_size;          // lr / (1 - beta1^t)
    float inv_bc2_sqrt;       // 1 / sqrt(1 - beta2^t)
    float beta1, beta2, eps;
    float one_minus_beta1, one_minus_beta2;   // computed in double on the host, as torch does
    float weight_decay = 0.f;                 // L2 term folded into the gradient (0: the trainer's setting)
};

// torch.optim.Adam single-tensor update (amsgrad False), tm:119-122,143:
//   g <- g + weight_decay p (when set) ; m <- m + (g - m)(1 - b1) ; v <- v b2 + (1 - b2) g g ; p <- p - step_size * m / (sqrt(v)/bc2_sqrt + eps)
__device__ inline void adam_update(float g, float& p, float& m, float& v, const AdamScalars& s) {
    // Moments: every operation pinned (no context-dependent fma contraction), so the fused
    // epilogue and the flat multi-tensor kernel stay bit-identical.  Step: v_sqrt_f32 / v_rcp_f32
    // (1 ulp) instead of the correctly rounded sequences (~10x the instructions); the term they
    // feed is scaled by lr/(1-b1^t) ~ 5e-4 before it meets p, so p moves by < 0.1 ulp of itself.
    if (s.weight_decay != 0.f) g = __fmaf_rn(s.weight_decay, p, g);
    m = __fmaf_rn(__fsub_rn(g, m), s.one_minus_beta1, m);
    v = __fmaf_rn(v, s.beta2, __fmul_rn(__fmul_rn(s.one_minus_beta2, g), g));
    const float denom = __fmaf_rn(__builtin_amdgcn_sqrtf(v), s.inv_bc2_sqrt, s.eps);
    p = __fmaf_rn(-s.step_size, __fmul_rn(m, __builtin_amdgcn_rcpf(denom)), p);
}

__device__ inline void adam_update4(const v4f& g, v4f& p, v4f& m, v4f& v, const AdamScalars& s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_update(g[e], pe, me, ve, s);
        p[e] = pe; m[e] = me; v[e] = ve;
    }
}

// Deferred Adam: a layer whose weight gradient went to the gradient arena in the PREVIOUS launch is
// updated by `kAdamBlocks` extra workgroups of the current one (p, g, m, v streamed while the other
// workgroups contract).  With Adam in the weight-gradient epilogue every workgroup of the launch does
// its update at the same moment, after the matrix pipe has gone idle: 15.7 us for the fused pair
// against 12.7 with a plain gradient store; store + 256 co-resident update workgroups: 14.2
// (tools/pair_probe.hip).  Same arithmetic as the epilogue and as adam_flat_kernel: bit-identical.
struct AdamSeg {
    float* p = nullptr;
    const float* g = nullptr;
    float* m = nullptr;
    float* v = nullptr;
    long long n4 = 0;          // float4 count (0: nothing pending)
    AdamScalars s{};
};
constexpr int kAdamBlocks = 256;    // workgroups per deferred-Adam segment (128 / 512 measured slower: profiles/r03_ab_adam_blocks.txt)
__device__ inline void adam_seg_body(const AdamSeg& a, int blk) {
    // (one float4 per thread in flight: two or four, and a delayed start, measured no better -- docs/experiments.md round 2)
    constexpr long long kStride = kAdamBlocks * 256ll;
    for (long long i = blk * 256ll + threadIdx.x; i < a.n4; i += kStride) {
        v4f pp = reinterpret_cast<v4f*>(a.p)[i];
        const v4f gg = reinterpret_cast<const v4f*>(a.g)[i];
        v4f mm = reinterpret_cast<v4f*>(a.m)[i];
        v4f vv = reinterpret_cast<v4f*>(a.v)[i];
        adam_update4(gg, pp, mm, vv, a.s);
        store_stream(a.p + 4 * i, pp);
        store_stream(a.m + 4 * i, mm);
        store_stream(a.v + 4 * i, vv);
    }
}
inline int adam_blocks(const AdamSeg* a) { return a && a->n4 > 0 ? kAdamBlocks : 0; }
// What one launch carries: up to two pending updates (a launch with little work of its own passes a big segment on to
// the next wide one and takes a small one instead: see take_pending in pvae.hip), kAdamBlocks workgroups each.
struct AdamPair {
    AdamSeg s[2];
    AdamPair() {}
    AdamPair(const AdamSeg& a) { s[0] = a; }                  // (probes under tools/ pass a single segment)
};
inline int adam_blocks(const AdamPair* p) { return p ? adam_blocks(&p->s[0]) + adam_blocks(&p->s[1]) : 0; }
__device__ inline void adam_pair_body(const AdamPair& p, int blk) {
    if (p.s[0].n4 > 0) {
        if (blk < kAdamBlocks) { adam_seg_body(p.s[0], blk); return; }
        blk -= kAdamBlocks;
    }
    adam_seg_body(p.s[1], blk);
}

// Bias gradient of a weight-gradient problem, db[q] = sum over the K rows of Q[k][q], for 32
// columns per workgroup (fixed summation order), handed to the epilogue's bias() (store, or Adam on
// the bias).  These few light workgroups are appended to every weight-gradient launch: summing the
// fragments inside the contraction loop instead put two VALU adds beside every four MFMAs of EVERY
// wave (only 1 tile column in 16 needs them) and cost 2 % of the step.
// Shape: a launch ends with its last workgroup, and these are latency chains sharing their CU with two
// contraction workgroups -- 16 workgroups of 64 columns x 16 row groups walked K = 256 rows in 8
// dependent round trips and kept the fused backward launch open 0.8 us longer (tools/pair_lab.hip:
// 14.28 vs 13.47 us).  32 columns x 32 row groups with all K / 32 loads of a thread in flight at once
// (full 128-byte lines per row) is ONE round trip for batches up to 256 rows.
__host__ __device__ inline int bias_tiles(const GemmArgs& ga) { return ga.tile32 ? ga.tiles_q : 2 * ga.tiles_q; }
template <class Epi>
__device__ inline void bias_grad_body(float* lds, int tile, const GemmArgs& ga, Epi& epi) {
    if (!epi.has_bias() || tile >= bias_tiles(ga)) return;
    const int tid = threadIdx.x, c4 = (tid & 7) * 4, r0 = tid >> 3;       // 32 rows x 32 columns per pass
    const float* __restrict__ src = ga.Q + (size_t)r0 * ga.ldq + tile * 32 + c4;
    const size_t step = (size_t)32 * ga.ldq;
    v4f s = v4f{0.f, 0.f, 0.f, 0.f};
    int k = r0;
    for (; k + 7 * 32 < ga.K; k += 8 * 32) {      // 8 rows per thread in flight (256 batch rows per trip)
        v4f t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = *reinterpret_cast<const v4f*>(src + i * step);
        s += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
        src += 8 * step;
    }
    for (; k < ga.K; k += 32) {
        s += *reinterpret_cast<const v4f*>(src);
        src += step;
    }
    *reinterpret_cast<v4f*>(lds + r0 * 32 + c4) = s;
    __syncthreads();
    if (tid < 32) {
        float v = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) v += lds[r * 32 + tid];
        epi.bias(tile * 32 + tid, v);
    }
}

template <class Epi, int ABL = 0>
__global__ void __launch_bounds__(256)
gemm_wgrad_reg_kernel(PVAE_GA_PARAMS(a_), int nw, Epi epi, AdamPair ad) {
    const GemmArgs ga = PVAE_GA_OF(a_);
    __shared__ __attribute__((aligned(16))) float lds[kPairLdsFloats];
    const int b = blockIdx.x, nb = bias_tiles(ga);
    if (b < nw) wgrad_body<Epi, ABL>(lds, b, ga, epi);
    else if (b < nw + nb) bias_grad_body(lds, b - nw, ga, epi);
    else adam_pair_body(ad, b - nw - nb);
}

// Horizontal fusion of two independent backward contractions in ONE launch: blocks [0, nd) run
// the input-gradient tiles of layer l-1, blocks [nd, nd + nw) the weight-gradient(+Adam) tiles of
// layer l.  The dgrad blocks are dispatched first (one per CU), the wgrad blocks land beside
// them (2 waves per SIMD), so one workgroup's load / Adam-traffic phases hide under the other's
// MFMA phases, and a launch boundary disappears.
// The next tiles_q blocks of each problem sum its bias gradient (bias_grad_body); the blocks past
// those (sa.rows_pad of them) stage the NEXT minibatch into the
// alternate input panels (StageArgs / stage_row below): the gather rides in the last launch of the
// step that precedes it instead of being a launch of its own.
// What the step's LAST launch does for the NEXT minibatch of a direct run: nothing is staged, but the rows the next
// step's first layers will gather are cold in HBM (an epoch walks the whole set once), and their first touch would sit on
// the critical path of those launches (+2 us each, measured).  A few workgroups of this launch read one dword per
// 128-byte line of those rows -- up to four contiguous runs: the state rows and the action rows of the (at most two)
// episode segments of the minibatch -- so that the Infinity Cache holds them when the next step starts.
struct TouchRuns {
    const float* p[4];
    int lines[4];                 // 128-byte lines per run (0: none)
    int blocks;                   // workgroups that do the touching
};
__device__ inline void touch_body(const TouchRuns& t, int blk) {
    float sink = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        for (int i = blk * 256 + (int)threadIdx.x; i < t.lines[r]; i += t.blocks * 256)
            asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(t.p[r] + (size_t)i * 32) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink)::"memory");
}
// The step's trailing weight gradient on the gathered first-layer input (no second problem, no staging workgroups).  The
// whole operand descriptor travels in the 14 preloadable leading dwords -- the s0 and s1 runs, their strides and widths,
// the batch's rows as two (base, start) runs -- so that the first tile fetch waits for no s_load (`rest` is only read for
// the index array of a minibatch with more than one episode jump).
template <class EpiW>
__global__ void __launch_bounds__(256, 3)       // (three waves per SIMD, as the staged kernel gets on its own: <= 168 VGPRs)
wgrad_pair_gather_kernel(const float* aQ, const float* g_s0, const float* g_s1, unsigned a_ld, unsigned a_t, unsigned a_k,
                         unsigned g_ld0n0, unsigned g_ld1n1, unsigned g_rq1, int g_b0, int g_b1, int na, EpiW e1, AdamPair ad,
                         TouchRuns touch, XSrc rest) {
    __shared__ __attribute__((aligned(16))) float lds[kPairLdsFloats];
    GemmArgs g1 = ga_unpack(aQ, nullptr, a_ld, a_t, a_k);
    const int gfl = g1.krot;                     // (the rotation bits carry: bit 0 = two-run row map, bit 1 = s1 row-indirect)
    g1.krot = 0;
    PGather ps;
    ps.x = rest;
    ps.x.s0 = g_s0; ps.x.ld0 = (int)(g_ld0n0 & 0xffffu); ps.x.n0 = (int)(g_ld0n0 >> 16);
    ps.x.s1 = g_s1; ps.x.ld1 = (int)(g_ld1n1 & 0xffffu); ps.x.n1 = (int)(g_ld1n1 >> 16);
    ps.x.ind1 = (gfl >> 1) & 1;
    ps.x.rows = (int)(g_rq1 & 0xffffu) + 1;
    ps.x.rm.seg = gfl & 1; ps.x.rm.q1 = (int)(g_rq1 >> 16) + 1; ps.x.rm.b0 = g_b0; ps.x.rm.b1 = g_b1;
    const int n1 = ga_grid(g1), nb1 = bias_tiles(g1);
    const int b = blockIdx.x;
    if (b < n1) wgrad_body<EpiW, 0, PGather>(lds, b, g1, e1, ps);
    else if (b < n1 + nb1) bias_grad_body(lds, b - n1, g1, e1);
    else if (b < n1 + nb1 + na) adam_pair_body(ad, b - n1 - nb1);
    else touch_body(touch, b - n1 - nb1 - na);
}
template <class EpiD, class EpiW, class PS>
__global__ void __launch_bounds__(256, 3)
bwd_pair_gather_kernel(PVAE_GA2_PARAMS, EpiD ed, EpiW ew, AdamPair ad, PS ps) {
    __shared__ __attribute__((aligned(16))) float lds[kPairLdsFloats];
    const GemmArgs gd = PVAE_GA2_A, gw = PVAE_GA2_B;
    const int nd = ga_grid(gd), nw = ga_grid(gw);
    const int b = blockIdx.x;
    if (b < nd) {
        if (gd.tile16) splitk_reg16_body<false, EpiD>(lds, b, gd, ed);
        else splitk_reg_body<false, EpiD, 0>(lds, b, gd, ed);
    } else if (b < nd + nw) wgrad_body<EpiW, 0, PS>(lds, b - nd, gw, ew, ps);
    else if (b < nd + nw + bias_tiles(gw)) bias_grad_body(lds, b - nd - nw, gw, ew);
    else adam_pair_body(ad, b - nd - nw - bias_tiles(gw));
}

template <class EpiW>
__global__ void __launch_bounds__(256)
wgrad_pair_kernel(PVAE_GA2_PARAMS, int na, EpiW e1, EpiW e2, StageArgs sa, AdamPair ad) {
    __shared__ __attribute__((aligned(16))) float lds[kPairLdsFloats];
    const GemmArgs g1 = PVAE_GA2_A, g2 = PVAE_GA2_B;
    const int n1 = ga_grid(g1), n12 = n1 + ga_grid(g2);
    const int b = blockIdx.x;
    const int nb1 = bias_tiles(g1), nb2 = bias_tiles(g2);
    if (b < n1) wgrad_body<EpiW>(lds, b, g1, e1);
    else if (b < n12) wgrad_body<EpiW>(lds, b - n1, g2, e2);
    else if (b < n12 + nb1) bias_grad_body(lds, b - n12, g1, e1);
    else if (b < n12 + nb1 + nb2) bias_grad_body(lds, b - n12 - nb1, g2, e2);
    else if (b < n12 + nb1 + nb2 + na) adam_pair_body(ad, b - n12 - nb1 - nb2);
    else stage_row(sa, b - n12 - nb1 - nb2 - na, 0, sa.rows_pad);
}

// (The dgrad half stays on the register-staged body here: with the wave-specialised body the
// pair needs 512-thread blocks and 64 KB of LDS per workgroup and measured 15 % slower.)
template <class EpiD, class EpiW, int ABL = 0>          // ABL: ablation bits of the two bodies (probes only)
__global__ void __launch_bounds__(256)
bwd_pair_kernel(PVAE_GA2_PARAMS, EpiD ed, EpiW ew, AdamPair ad) {
    __shared__ __attribute__((aligned(16))) float lds[kPairLdsFloats];
    const GemmArgs gd = PVAE_GA2_A, gw = PVAE_GA2_B;
    const int nd = ga_grid(gd), nw = ga_grid(gw);          // (a body with tiles_q = 0 is absent: ablation probes)
    PVAE_MARK(0, 0);
    PVAE_MARK_HW();
    // (dispatch order matters: input-gradient workgroups first.  Weight-gradient workgroups first: world step
    //  90.0 -> 93.5 us; the two kinds interleaved: 91.8 us -- the older waves of a SIMD win issue arbitration,
    //  and it is the input gradient whose reduction + store epilogue can hide under the partner's MFMAs)
    const int b = blockIdx.x;
    if (b < nd) {
        if (gd.tile16) splitk_reg16_body<false, EpiD>(lds, b, gd, ed);
        else splitk_reg_body<false, EpiD, ABL>(lds, b, gd, ed);
    } else if (b < nd + nw) wgrad_body<EpiW, ABL>(lds, b - nd, gw, ew);
    else if (b < nd + nw + bias_tiles(gw)) bias_grad_body(lds, b - nd - nw, gw, ew);
    else adam_pair_body(ad, b - nd - nw - bias_tiles(gw));
    PVAE_MARK(0, 3);
}

// The same fused launch with the input gradient on 64x32 tiles (splitk_reg64_body; 48 KB of LDS per workgroup): 512 rows
// and more.
template <class EpiD, class EpiW>
__global__ void __launch_bounds__(256)
bwd_pair64_kernel(PVAE_GA2_PARAMS, EpiD ed, EpiW ew, AdamPair ad) {
    __shared__ __attribute__((aligned(16))) float lds[kReg64RingFloats];
    const GemmArgs gd = PVAE_GA2_A, gw = PVAE_GA2_B;
    const int nd = ga_grid(gd), nw = ga_grid(gw);
    const int b = blockIdx.x;
    if (b < nd) splitk_reg64_body<EpiD>(lds, b, gd, ed);
    else if (b < nd + nw) wgrad_body<EpiW>(lds, b - nd, gw, ew);
    else if (b < nd + nw + bias_tiles(gw)) bias_grad_body(lds, b - nd - nw, gw, ew);
    else adam_pair_body(ad, b - nd - nw - bias_tiles(gw));
}

// ---------------------------------------------------------------------------------------
// epilogues: (q, p, 4 consecutive p values)
// ---------------------------------------------------------------------------------------
// sum over a block of 256 or 512 threads (fixed order: deterministic)
__device__ inline float block_sum_256(float v, float* scratch) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    if (blockDim.x > 256) s += (scratch[4] + scratch[5]) + (scratch[6] + scratch[7]);
    return s;
}

// Hidden-layer activation (get_activation_fn rmt:30-46 via gen_layers' act_hidden, tpv:180-192):
// internal codes 0 linear, 1 relu (the trainer's "act_fn"), 2 tanh, 3 sigmoid, 4 elu (alpha 1).  The backward
// pass needs only the layer's OUTPUT a: relu' = [a > 0], tanh' = 1 - a^2, sigmoid' = a (1 - a), elu' = a > 0 ? 1 : a + 1.
__device__ inline float act_apply(float x, int act) {
    switch (act) {
        case 1: return fmaxf(x, 0.f);
        case 2: return tanhf(x);
        case 3: return 1.f / (1.f + expf(-x));
        case 4: return x > 0.f ? x : expm1f(x);
        default: return x;
    }
}
__device__ inline float act_grad(float a, int act) {
    switch (act) {
        case 2: return 1.f - a * a;
        case 3: return a * (1.f - a);
        case 4: return a > 0.f ? 1.f : a + 1.f;
        default: return 1.f;
    }
}

struct EpiBiasAct {           // forward layer: out = act(acc + bias)
    float* out;
    int ldo;
    const float* bias;        // may be null
    int act;                  // act_apply code (0: linear output layer)
    float* out2 = nullptr;    // optional second destination for columns [0, n2): out2[q][off2 + p]
    int ld2 = 0, off2 = 0, n2 = 0;   // (motor-decoder output -> action columns of the world-model input)
    int n_valid = 1 << 30;    // real width of the layer: pad columns stay 0 also where act(0) != 0 (sigmoid),
                              // so that padding never reaches a weight gradient
    // operands of the epilogue that do not depend on the contraction are fetched BEFORE the main
    // loop (`preload`) and handed back at the end: their latency hides under the loop instead of
    // sitting between the last MFMA and the stores
    struct Pre { v4f b; };
    __device__ inline Pre preload(int, int p) const {
        Pre r;
        r.b = bias ? *reinterpret_cast<const v4f*>(bias + p) : v4f{0.f, 0.f, 0.f, 0.f};
        return r;
    }
    __device__ inline void operator()(int q, int p, v4f v, const Pre& pre) const {
        v += pre.b;
        if (act == 1) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (act > 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = p + e < n_valid ? act_apply(v[e], act) : 0.f;
        }
        store_stream(out + (size_t)q * ldo + p, v);
        if (out2) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (p + e < n2) out2[(size_t)q * ld2 + off2 + p + e] = v[e];
        }
    }
    __device__ inline void finish(float*, int, int) const {}
};

// Output layer fused with nn.MSELoss (tm:99) and its gradient (world-model MSE tpv:411-414,
// cycle loss tpv:417-419): pred = acc + bias is stored, dz = grad_scale * (pred - target) for
// valid rows/columns (zeros elsewhere, so padded tiles stay inert downstream), and the
// workgroup's squared-error sum goes to partial[tile] (summed later in a fixed order).
struct EpiMse {
    float* out;
    int ldo;
    const float* bias;
    const float* target;
    int ldt;
    float* dz;                // may be null (forward only)
    int ldz;
    int rows, D;
    float grad_scale;
    float* partial;
    int l1;                   // 0: nn.MSELoss (sum d^2, grad 2d/n), 1: nn.L1Loss (sum |d|, grad sign(d)/n)
    float sq = 0.f;
    int tind = 0;                    // 1: the target of batch row q is row trm.at(q) of `target` (s_{t+1} read from `states`
    RowMap trm{};                    // where it lies, 4-byte aligned; columns >= D are never used)
    struct Pre { v4f b, t; };
    __device__ inline Pre preload(int q, int p) const {
        Pre r;
        r.b = bias ? *reinterpret_cast<const v4f*>(bias + p) : v4f{0.f, 0.f, 0.f, 0.f};
        const size_t tq = tind ? (size_t)trm.at(q < rows ? q : 0) : (size_t)q;
        r.t = *reinterpret_cast<const v4f*>(target + tq * ldt + (tind && p >= D ? 0 : p));
        return r;
    }
    __device__ inline void operator()(int q, int p, v4f v, const Pre& pre) {
        v += pre.b;
        *reinterpret_cast<v4f*>(out + (size_t)q * ldo + p) = v;
        const v4f t = pre.t;
        v4f g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool valid = q < rows && p + e < D;
            const float d = v[e] - t[e];
            if (valid) sq += l1 ? fabsf(d) : d * d;
            const float gd = l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d;
            g[e] = valid ? grad_scale * gd : 0.f;
        }
        if (dz) *reinterpret_cast<v4f*>(dz + (size_t)q * ldz + p) = g;
    }
    __device__ inline void finish(float* scratch, int tile, int tid) {
        const float s = block_sum_256(sq, scratch);
        if (tid == 0) partial[tile] = s;
    }
};

struct EpiMask {              // input gradient: out = acc * act'(a)   (relu: acc where a > 0, else 0)
    float* out;
    int ldo;
    const float* mask;        // activation a of the producing layer (its output), or null
    int ldm;
    int act = 1;              // act_grad code of that layer (0: linear, the gradient passes unchanged)
    struct Pre { v4f m; };
    __device__ inline Pre preload(int q, int p) const {
        Pre r;
        r.m = mask ? *reinterpret_cast<const v4f*>(mask + (size_t)q * ldm + p) : v4f{1.f, 1.f, 1.f, 1.f};
        return r;
    }
    __device__ inline void operator()(int q, int p, v4f v, const Pre& pre) const {
        const v4f m = pre.m;
        if (act == 0) {                         // no activation behind that layer (rmt:32-33 "linear"): dz = acc
        } else if (act == 1 || !mask) {
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
            v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        } else {
            v.x *= act_grad(m.x, act); v.y *= act_grad(m.y, act);
            v.z *= act_grad(m.z, act); v.w *= act_grad(m.w, act);
        }
        store_stream(out + (size_t)q * ldo + p, v);
    }
    __device__ inline void finish(float*, int, int) const {}
};

// Input gradient of the WORLD MODEL's first layer (joint phase): its columns [c0, c0+n) are
// d(loss)/d(a_hat) arriving through the frozen world model (cycle loss, tpv:417-419).  Instead of
// storing the panel for a separate kernel, this epilogue forms the motor decoder's output gradient
// right here (tpv:381-382 + chain rule):
//     dz[q][c] = grad_scale * f(a_hat[q][c] - a[q][c]) + acc[q][c0+c]        (f: MSE d, L1 sign d)
// and the action-reconstruction loss partial of its workgroup.  The other columns (gradient wrt the
// state) have no consumer at lookahead 1 and are not stored.  Rows >= `rows` get zeros.
struct EpiActionSeed {
    const float* pred;  int ldp;      // a_hat: motor decoder output
    const float* target; int ldt;     // demonstrated action
    float* dz; int ldz;               // -> gradient wrt the decoder's output layer
    int c0, n, rows;
    float grad_scale;
    int l1;
    float* partial;
    float sq = 0.f;
    int tind = 0;                     // 1: the demonstrated action of batch row q is row trm.at(q) of `target` (`actions`)
    RowMap trm{};
    struct Pre {};
    __device__ inline Pre preload(int, int) const { return Pre(); }
    __device__ inline void operator()(int q, int p, v4f v, const Pre&) {
        if (p + 3 < c0 || p >= c0 + n) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = p + e - c0;
            if (c < 0 || c >= n) continue;
            float g = 0.f;
            if (q < rows) {
                const float d = pred[(size_t)q * ldp + c] - target[(size_t)(tind ? trm.at(q) : q) * ldt + c];
                sq += l1 ? fabsf(d) : d * d;
                g = grad_scale * (l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d) + v[e];
            }
            dz[(size_t)q * ldz + c] = g;
        }
    }
    __device__ inline void finish(float* scratch, int tile, int tid) {
        const float s = block_sum_256(sq, scratch);
        if (tid == 0) partial[tile] = s;
    }
};

// Input gradient of the MOTOR DECODER's first layer: its columns [c0, c0+Z) are d(loss)/dz.  This
// epilogue is the backward of the sampler + KL (autograd of rmt:734-740 and tpv:388) and writes the
// task encoder's output gradient directly:
//     dmu = dz + (beta/B) mu ;  dlogvar = dz * eps * 0.5 exp(0.5 lv) + (beta/B) 0.5 (exp(lv) - 1)
// Nothing else of the panel has a consumer at lookahead 1.
struct EpiSamplerSeed {
    const float* te_out; int ldte;    // [mu | logvar]
    const float* eps;                 // [rows_pad][Z] draws actually used
    float* dz; int ldz;               // -> gradient wrt the encoder's output layer [.. | dmu | dlogvar]
    int c0, Z, rows;
    float kl_scale;
    // learned prior mean (PVAE_PRIOR_STATE_MEAN; null otherwise): KL(N(mu,s^2) || N(mu_p,1)) pulls mu and
    // mu_p together -- dmu gets +kl (mu - mu_p), the prior stack's output gradient is the negative of it
    const float* mu_p = nullptr; int ldmp = 0;
    float* dz_p = nullptr; int ldzp = 0;
    struct Pre {};
    __device__ inline Pre preload(int, int) const { return Pre(); }
    __device__ inline void operator()(int q, int p, v4f v, const Pre&) const {
        if (p + 3 < c0 || p >= c0 + Z) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j = p + e - c0;
            if (j < 0 || j >= Z) continue;
            float gm = 0.f, gl = 0.f, gp = 0.f;
            if (q < rows) {
                const float mu = te_out[(size_t)q * ldte + j];
                const float lv = te_out[(size_t)q * ldte + Z + j];
                const float ep = eps[(size_t)q * Z + j];
                if (mu_p) {
                    gp = kl_scale * (mu - mu_p[(size_t)q * ldmp + j]);
                    gm = v[e] + gp;
                } else {
                    gm = v[e] + kl_scale * mu;
                }
                gl = v[e] * ep * 0.5f * expf(0.5f * lv) + kl_scale * 0.5f * (expf(lv) - 1.0f);
            }
            dz[(size_t)q * ldz + j] = gm;
            dz[(size_t)q * ldz + Z + j] = gl;
            if (dz_p) dz_p[(size_t)q * ldzp + j] = -gp;
        }
    }
    __device__ inline void finish(float*, int, int) const {}
};

// Fixed-order sum of the per-workgroup loss partials -> {total, loss_a, loss_kl, loss_s,
// loss_cyc} (tpv:430-435 weighting).  Runs in one wave: as the tail of the step's last
// weight-gradient launch (training) or as its own tiny kernel (evaluation).
struct LossFinal {
    const float* part[4];     // per-term partial arrays (a, kl, s, cyc)
    float* out;               // null = nothing to do
    float scale[4];           // a, kl, s, cyc: 1/(B*Da), 1/B, 1/(B*Db), 1/(B*Db)
    float coeff[4];
    int nparts[4];            // 0 = term inactive
};
__device__ inline void finalize_loss_wave(const LossFinal& f, int lane) {
    float total = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float s = 0.f;
        for (int i = lane; i < f.nparts[t]; i += 64) s += f.part[t][i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        s *= f.scale[t];
        if (lane == 0) f.out[1 + t] = s;
        total += f.coeff[t] * s;
    }
    if (lane == 0) f.out[0] = total;
}

struct EpiGradStore {         // weight gradient -> gradient arena (data-parallel path)
    float* g;
    int ld;
    float* gb = nullptr;      // bias gradient destination (null = none)
    LossFinal loss{};
    struct Pre {};
    __device__ inline Pre load(int, int) const { return Pre{}; }
    __device__ inline void apply(int q, int p, v4f v, const Pre&) const { (*this)(q, p, v); }
    __device__ inline void operator()(int q, int p, v4f v) const { store_stream(g + (size_t)q * ld + p, v); }
    __device__ inline bool has_bias() const { return gb != nullptr; }
    __device__ inline void bias(int q, float v) const { gb[q] = v; }
};

struct EpiGradAdam {          // weight gradient consumed in registers by Adam (1-GPU path)
    float* w;
    float* m;
    float* v;
    int ld;
    AdamScalars s;
    float* b = nullptr;       // bias / its moments (null = none)
    float* bm = nullptr;
    float* bv = nullptr;
    LossFinal loss{};
    struct Pre { v4f w, m, v; };
    // the Adam operands of a tile are fetched while the contraction runs (load) and consumed
    // in registers afterwards (apply)
    __device__ inline Pre load(int q, int p) const {
        const size_t o = (size_t)q * ld + p;
        return Pre{*reinterpret_cast<const v4f*>(w + o), *reinterpret_cast<const v4f*>(m + o),
                   *reinterpret_cast<const v4f*>(v + o)};
    }
    __device__ inline void apply(int q, int p, v4f g, Pre pre) const {
        const size_t o = (size_t)q * ld + p;
        adam_update4(g, pre.w, pre.m, pre.v, s);
        store_stream(w + o, pre.w);
        store_stream(m + o, pre.m);
        store_stream(v + o, pre.v);
    }
    __device__ inline void operator()(int q, int p, v4f g) const { apply(q, p, g, load(q, p)); }
    __device__ inline bool has_bias() const { return b != nullptr; }
    __device__ inline void bias(int q, float g) const {
        float pb = b[q], pm = bm[q], pv = bv[q];
        adam_update(g, pb, pm, pv, s);
        b[q] = pb; bm[q] = pm; bv[q] = pv;
    }
};

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// When the profiler arms a pair of events, the next launch goes through hipExtLaunchKernelGGL, which
// stamps them with the kernel's own start and end on the device (what rocprofv3 reports as the kernel
// duration); events recorded around a plain launch would include the launch seam.
inline hipEvent_t g_kernel_ev[2] = {nullptr, nullptr};
#define PVAE_LAUNCH(kernel, grid, block, st, ...)                                                          \
    do {                                                                                                   \
        if (g_kernel_ev[0]) {                                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, st, g_kernel_ev[0], g_kernel_ev[1], 0, __VA_ARGS__); \
            g_kernel_ev[0] = g_kernel_ev[1] = nullptr;                                                     \
        } else                                                                                               \
            hipLaunchKernelGGL(kernel, grid, block, 0, st, __VA_ARGS__);                                   \
    } while (0)
struct GemmGrid {
    int tiles_q, tiles_p, p_per_xcd, grid;
};
inline GemmGrid make_grid(int rows_q, int cols_p, int bq, int bp) {
    GemmGrid g;
    g.tiles_q = rows_q / bq;
    g.tiles_p = cols_p / bp;
    g.p_per_xcd = (g.tiles_p + 7) / 8;
    g.grid = 8 * g.p_per_xcd * g.tiles_q;
    return g;
}

// 512 rows and more: 64x32 tiles (splitk_ws64_body) whenever they still give every CU a workgroup; PVAE_WS64=0: off (A/B)
inline int g_ws64 = 1;                       // pvae_set_option(NULL, "ws64", 0): off (tests compare the tilings bit for bit)
inline bool uses_64x32(int M, int N) { return g_ws64 && M >= 512 && M % 64 == 0 && (M / 64) * (N / 32) >= 256; }
// 1024 rows and more: 64x64 tiles whenever THEY still give every CU a workgroup; PVAE_WS6464=0: off (A/B)
inline int g_ws6464 = 1;                     // option "ws6464"
inline bool uses_64x64(int M, int N) { return g_ws6464 && uses_64x32(M, N) && N % 64 == 0 && (M / 64) * (N / 64) >= 256; }
// ... with the XCDs partitioning the ROW blocks (see splitk_ws64_body); PVAE_WS6464_ROWS=0: column ranges as elsewhere (A/B)
inline int g_ws6464_rows = 1;                // option "ws6464_rows"
inline GemmGrid make_grid_6464(int M, int N, GemmArgs& ga) {
    GemmGrid g = make_grid(M, N, 64, 64);
    ga.tiles_q = g.tiles_q; ga.tiles_p = g.tiles_p; ga.p_per_xcd = g.p_per_xcd;
    if (g_ws6464_rows) {
        ga.rowxcd = 1;
        g.grid = 8 * ((g.tiles_q + 7) / 8) * g.tiles_p;
    }
    return g;
}
// ... and the input-gradient half of the fused backward pairs (PVAE_PAIR64=0: off, A/B)
inline int g_pair64 = 1;                     // option "pair64"
inline bool pair_uses_64x32(int M, int N) { return g_pair64 && uses_64x32(M, N); }
// narrow outputs: under 128 workgroups of 32x32 -> use 16x16 tiles (4x the workgroups)
inline bool forward_uses_16x16(int M, int N) { return (M / 32) * (N / 32) < 128; }
inline int forward_tiles(int M, int N) {
    return forward_uses_16x16(M, N) ? (M / 16) * (N / 16) : (M / 32) * (N / 32);
}

// forward: out[M][N] = act(X[M][K] W[N][K]^T + b)
template <class Epi>
inline hipError_t gemm_forward_epi(const float* X, int ldx, const float* W, int ldw, int M, int N, int K,
                                   const Epi& e, hipStream_t st) {
    if (forward_uses_16x16(M, N)) {
        const GemmGrid g = make_grid(M, N, 16, 16);
        const GemmArgs ga{X, ldx, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
        PVAE_LAUNCH((gemm_splitk_reg16_kernel<true, Epi>), dim3(g.grid), dim3(256), st, PVAE_GA_PASS(ga), e);
        return hipGetLastError();
    }
    if constexpr (std::is_same<Epi, EpiBiasAct>::value) {
        if (uses_64x64(M, N)) {
            GemmArgs ga{X, ldx, W, ldw, K, 0, 0, 0};
            const GemmGrid g = make_grid_6464(M, N, ga);
            PVAE_LAUNCH((gemm_splitk_ws64_kernel<true, Epi, 64>), dim3(g.grid), dim3(512), st, PVAE_GA_PASS(ga), e);
            return hipGetLastError();
        }
        if (uses_64x32(M, N)) {
            const GemmGrid g = make_grid(M, N, 64, 32);
            const GemmArgs ga{X, ldx, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
            PVAE_LAUNCH((gemm_splitk_ws64_kernel<true, Epi>), dim3(g.grid), dim3(512), st, PVAE_GA_PASS(ga), e);
            return hipGetLastError();
        }
    }
    const GemmGrid g = make_grid(M, N, 32, 32);
    const GemmArgs ga{X, ldx, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
    PVAE_LAUNCH((gemm_splitk_ws_kernel<true, Epi>), dim3(g.grid), dim3(kWsThreads), st, PVAE_GA_PASS(ga), e);
    return hipGetLastError();
}
// forward layer on 32x32 tiles whose launch forms some of its own input columns (Pro, see splitk_ws_body)
inline bool forward_pro_ok(int M, int N) { return !forward_uses_16x16(M, N) && !uses_64x32(M, N) && !g_krot && !g_rowxcd; }
template <class Epi, class Pro>
inline hipError_t gemm_forward_pro(const float* X, int ldx, const float* W, int ldw, int M, int N, int K, const Epi& e,
                                   const Pro& pro, hipStream_t st) {
    const GemmGrid g = make_grid(M, N, 32, 32);
    const GemmArgs ga{X, ldx, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
    PVAE_LAUNCH((gemm_splitk_ws_pro_kernel<Epi, Pro>), dim3(g.grid), dim3(kWsThreads), st, PVAE_GA_PASS(ga), e, pro);
    return hipGetLastError();
}
// a stack's FIRST layer on the gathered operand (XSrc): same tile geometries as gemm_forward_epi, never the 16x16 kernel
inline bool forward_gather_ok(int M, int N) { return !forward_uses_16x16(M, N) && !g_krot && !g_rowxcd; }
#define PVAE_GG_PASS(xs, ga) (xs).s0, (ga).P, (int)((unsigned)(xs).ld0 | ((unsigned)(xs).n0 << 16)), (ga).ldp, (ga).K, (ga).tiles_q, (ga).tiles_p, \
                             (ga).p_per_xcd, ga_flags(ga) | ((xs).rm.seg ? 64 : 0), (int)((unsigned)((xs).rows - 1) | ((unsigned)((xs).rm.q1 - 1) << 16)), \
                             (xs).rm.b0, (xs).rm.b1
inline bool gather_packable(const XSrc& xs) {
    return xs.ld0 > 0 && xs.ld0 < 65536 && xs.n0 >= 64 && xs.n0 < 65536 && xs.rows >= 1 && xs.rows <= 65536 && xs.rm.q1 >= 1 && xs.rm.q1 <= 65536;
}
template <class Epi>
inline hipError_t gemm_forward_gather(const XSrc& xs, const float* W, int ldw, int M, int N, int K, const Epi& e, hipStream_t st) {
    if (!gather_packable(xs)) return hipErrorInvalidValue;
    if constexpr (std::is_same<Epi, EpiBiasAct>::value) {
        if (uses_64x64(M, N)) {
            GemmArgs ga{nullptr, 0, W, ldw, K, 0, 0, 0};
            const GemmGrid g = make_grid_6464(M, N, ga);
            PVAE_LAUNCH((gemm_splitk_ws64_gather_kernel<Epi, 64>), dim3(g.grid), dim3(512), st, PVAE_GG_PASS(xs, ga), e, xs);
            return hipGetLastError();
        }
        if (uses_64x32(M, N)) {
            const GemmGrid g = make_grid(M, N, 64, 32);
            const GemmArgs ga{nullptr, 0, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
            PVAE_LAUNCH((gemm_splitk_ws64_gather_kernel<Epi, 32>), dim3(g.grid), dim3(512), st, PVAE_GG_PASS(xs, ga), e, xs);
            return hipGetLastError();
        }
    }
    const GemmGrid g = make_grid(M, N, 32, 32);
    const GemmArgs ga{nullptr, 0, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
    PVAE_LAUNCH((gemm_splitk_ws_gather_kernel<Epi>), dim3(g.grid), dim3(kWsThreads), st, PVAE_GG_PASS(xs, ga), e, xs);
    return hipGetLastError();
}
template <class Epi, class Pro>
inline hipError_t gemm_forward_pro_gather(const XSrc& xs, const float* W, int ldw, int M, int N, int K, const Epi& e,
                                          const Pro& pro, hipStream_t st) {
    if (!gather_packable(xs)) return hipErrorInvalidValue;
    const GemmGrid g = make_grid(M, N, 32, 32);
    const GemmArgs ga{nullptr, 0, W, ldw, K, g.tiles_q, g.tiles_p, g.p_per_xcd};
    PVAE_LAUNCH((gemm_splitk_ws_pro_gather_kernel<Epi, Pro>), dim3(g.grid), dim3(kWsThreads), st, PVAE_GG_PASS(xs, ga), e, pro, xs);
    return hipGetLastError();
}
inline hipError_t gemm_forward(const float* X, int ldx, const float* W, int ldw, const float* bias,
                               float* out, int ldo, int M, int N, int K, int act, hipStream_t st) {
    EpiBiasAct e{out, ldo, bias, act};
    return gemm_forward_epi(X, ldx, W, ldw, M, N, K, e, st);
}
// dgrad: dX[M][Kin] = (dZ[M][N] W[N][Kin]) .* (mask > 0)
// Tile geometry: 32x32, or 16x16 when that leaves fewer than 128 workgroups (narrow first layers;
// PVAE_DGRAD16=0 switches it off: A/B).
inline int g_dgrad16 = 1;                    // option "dgrad16"
inline bool dgrad_uses_16x16(int M, int Kin) { return g_dgrad16 && (M / 32) * (Kin / 32) < 128; }
// workgroups whose epilogue sees a tile (= loss partials a seed epilogue writes)
inline int dgrad_tiles(int M, int Kin) {
    return dgrad_uses_16x16(M, Kin) ? (M / 16) * (Kin / 16) : (M / 32) * (Kin / 32);
}
struct DgradPlan {
    GemmArgs ga;
    int grid;
};
inline DgradPlan plan_dgrad(const float* dZ, int ldz, const float* W, int ldw, int M, int Kin, int N) {
    const bool t16 = dgrad_uses_16x16(M, Kin);
    const GemmGrid g = t16 ? make_grid(M, Kin, 16, 16) : make_grid(M, Kin, 32, 32);
    DgradPlan d{GemmArgs{dZ, ldz, W, ldw, N, g.tiles_q, g.tiles_p, g.p_per_xcd}, g.grid};
    d.ga.tile16 = t16 ? 1 : 0;
    return d;
}
// same contraction with a caller-supplied epilogue (gradient seeds of the producing stack)
template <class EpiD>
inline hipError_t gemm_dgrad_epi(const float* dZ, int ldz, const float* W, int ldw, int M, int Kin, int N,
                                 const EpiD& e, hipStream_t st) {
    if constexpr (std::is_same<EpiD, EpiMask>::value) {
        if (uses_64x64(M, Kin)) {                               // ... at >= 1024 rows
            GemmArgs ga{dZ, ldz, W, ldw, N, 0, 0, 0};
            const GemmGrid g = make_grid_6464(M, Kin, ga);
            PVAE_LAUNCH((gemm_splitk_ws64_kernel<false, EpiD, 64>), dim3(g.grid), dim3(512), st, PVAE_GA_PASS(ga), e);
            return hipGetLastError();
        }
        if (uses_64x32(M, Kin)) {                               // stand-alone input gradient of a hidden layer at >= 512 rows
            const GemmGrid g = make_grid(M, Kin, 64, 32);
            const GemmArgs ga{dZ, ldz, W, ldw, N, g.tiles_q, g.tiles_p, g.p_per_xcd};
            PVAE_LAUNCH((gemm_splitk_ws64_kernel<false, EpiD>), dim3(g.grid), dim3(512), st, PVAE_GA_PASS(ga), e);
            return hipGetLastError();
        }
    }
    const DgradPlan d = plan_dgrad(dZ, ldz, W, ldw, M, Kin, N);
    if (d.ga.tile16) PVAE_LAUNCH((gemm_splitk_reg16_kernel<false, EpiD>), dim3(d.grid), dim3(256), st, PVAE_GA_PASS(d.ga), e);
    else PVAE_LAUNCH((gemm_splitk_ws_kernel<false, EpiD>), dim3(d.grid), dim3(kWsThreads), st, PVAE_GA_PASS(d.ga), e);
    return hipGetLastError();
}
inline hipError_t gemm_dgrad(const float* dZ, int ldz, const float* W, int ldw, const float* mask,
                             int ldm, float* dX, int ldo, int M, int Kin, int N, hipStream_t st, int act = 1) {
    const EpiMask e{dX, ldo, mask, ldm, act};
    return gemm_dgrad_epi(dZ, ldz, W, ldw, M, Kin, N, e, st);
}
// wgrad: G[N][Kin] = dZ[M][N]^T X[M][Kin]  (+ bias gradient, + optional loss finalisation)
// Tile geometry per problem: 64x64, or 32x32 when that leaves at most half the CUs with a tile
// (PVAE_WGRAD32=0 switches the small geometry off: A/B).
inline int g_wgrad32 = 1;                    // option "wgrad32": 0 never, 1 where it pays, 2 always
inline bool wgrad_uses_32x32(int N, int Kin, int M) {
    return g_wgrad32 && ((N / 64) * (Kin / 64) <= 128 || g_wgrad32 == 2) && M % 64 == 0;
}
struct WgradPlan {
    GemmArgs ga;
    int grid, nbias;          // contraction workgroups, bias-gradient workgroups (32 columns each)
};
inline WgradPlan plan_wgrad(const float* dZ, int ldz, const float* X, int ldx, int N, int Kin, int M) {
    const bool t32 = wgrad_uses_32x32(N, Kin, M);
    const GemmGrid g = t32 ? make_grid(N, Kin, 32, 32) : make_grid(N, Kin, 64, 64);
    WgradPlan p{GemmArgs{dZ, ldz, X, ldx, M, g.tiles_q, g.tiles_p, g.p_per_xcd}, g.grid, N / 32};
    p.ga.tile32 = t32 ? 1 : 0;
    return p;
}
template <class Epi>
inline hipError_t gemm_wgrad(const float* dZ, int ldz, const float* X, int ldx, int N, int Kin, int M,
                             const Epi& e, hipStream_t st, const AdamPair* ad = nullptr) {
    const WgradPlan w = plan_wgrad(dZ, ldz, X, ldx, N, Kin, M);
    PVAE_LAUNCH((gemm_wgrad_reg_kernel<Epi>), dim3(w.grid + w.nbias + adam_blocks(ad)), dim3(256), st, PVAE_GA_PASS(w.ga),
                w.grid, e, ad ? *ad : AdamPair());
    return hipGetLastError();
}
// one launch, two independent weight gradients (the two last layers of a backward pass)
template <class EpiW>
inline hipError_t gemm_wgrad_pair(const float* dZ1, int ldz1, const float* X1, int ldx1, int N1, int Kin1,
                                  const EpiW& e1, const float* dZ2, int ldz2, const float* X2, int ldx2, int N2,
                                  int Kin2, const EpiW& e2, int M, hipStream_t st, const StageArgs* next = nullptr,
                                  const AdamPair* ad = nullptr) {
    const WgradPlan w1 = plan_wgrad(dZ1, ldz1, X1, ldx1, N1, Kin1, M);
    const WgradPlan w2 = plan_wgrad(dZ2, ldz2, X2, ldx2, N2, Kin2, M);
    StageArgs sa;
    memset(&sa, 0, sizeof(sa));
    if (next) sa = *next;                     // rows_pad extra blocks gather the next minibatch
    if (!ga_packable(w1.ga) || !ga_packable(w2.ga) || ga_grid(w1.ga) != w1.grid || ga_grid(w2.ga) != w2.grid) return hipErrorInvalidValue;
    PVAE_LAUNCH((wgrad_pair_kernel<EpiW>), dim3(w1.grid + w2.grid + w1.nbias + w2.nbias + adam_blocks(ad) + sa.rows_pad),
                dim3(256), st, PVAE_GA2_PASS(w1.ga, w2.ga), adam_blocks(ad), e1, e2, sa, ad ? *ad : AdamPair());
    return hipGetLastError();
}
// the same with problem 1's X gathered (XSrc) -- a stack's first layer; nothing to stage then
template <class EpiW>
inline hipError_t gemm_wgrad_pair_gather(const float* dZ1, int ldz1, const XSrc& xs, int N1, int Kin1, const EpiW& e1, int M,
                                         hipStream_t st, const AdamPair* ad = nullptr, const TouchRuns* touch = nullptr) {
    WgradPlan w1 = plan_wgrad(dZ1, ldz1, nullptr, 0, N1, Kin1, M);
    if (!ga_packable(w1.ga) || ga_grid(w1.ga) != w1.grid || !gather_packable(xs) || xs.ld1 < 0 || xs.ld1 >= 65536 || xs.n1 < 0 ||
        xs.n1 >= 65536)
        return hipErrorInvalidValue;
    w1.ga.krot = (xs.rm.seg ? 1 : 0) | (xs.ind1 ? 2 : 0);
    TouchRuns tr;
    memset(&tr, 0, sizeof(tr));
    if (touch) tr = *touch;
    const float* s1 = xs.n1 > 0 ? xs.s1 : xs.s0;                 // (never dereferenced for a value when n1 == 0, but always an address)
    PVAE_LAUNCH((wgrad_pair_gather_kernel<EpiW>), dim3(w1.grid + w1.nbias + adam_blocks(ad) + tr.blocks), dim3(256), st,
                w1.ga.Q, xs.s0, s1, ga_pack_ld(w1.ga), ga_pack_t(w1.ga), ga_pack_k(w1.ga), (unsigned)xs.ld0 | ((unsigned)xs.n0 << 16),
                (unsigned)xs.ld1 | ((unsigned)xs.n1 << 16), (unsigned)(xs.rows - 1) | ((unsigned)(xs.rm.q1 - 1) << 16), xs.rm.b0, xs.rm.b1,
                adam_blocks(ad), e1, ad ? *ad : AdamPair(), tr, xs);
    return hipGetLastError();
}
// input gradient (16x16 / 32x32 tiles, caller's epilogue) || weight gradient on the gathered X
template <class EpiD, class EpiW>
inline hipError_t gemm_bwd_pair_epi_gather(const float* dZd, int ldzd, const float* Wd, int ldwd, int Md, int Kind, int Nd,
                                           const EpiD& ed, const float* dZw, int ldzw, const XSrc& xs, int Nw, int Kinw,
                                           int Mw, const EpiW& ew, hipStream_t st, const AdamPair* ad = nullptr) {
    const DgradPlan d = plan_dgrad(dZd, ldzd, Wd, ldwd, Md, Kind, Nd);
    const WgradPlan w = plan_wgrad(dZw, ldzw, nullptr, 0, Nw, Kinw, Mw);
    if (!ga_packable(d.ga) || !ga_packable(w.ga) || ga_grid(d.ga) != d.grid || ga_grid(w.ga) != w.grid) return hipErrorInvalidValue;
    const PGather ps{xs};
    PVAE_LAUNCH((bwd_pair_gather_kernel<EpiD, EpiW, PGather>), dim3(d.grid + w.grid + w.nbias + adam_blocks(ad)), dim3(256), st,
                PVAE_GA2_PASS(d.ga, w.ga), ed, ew, ad ? *ad : AdamPair(), ps);
    return hipGetLastError();
}
// one launch: dX'[M][Kin'] = (dZ'[M][N'] W'[N'][Kin']) .* mask   ||   G[N][Kin] = dZ[M][N]^T X[M][Kin]
template <class EpiD, class EpiW>
inline hipError_t gemm_bwd_pair_epi(const float* dZd, int ldzd, const float* Wd, int ldwd, int Md, int Kind, int Nd,
                                    const EpiD& ed, const float* dZw, int ldzw, const float* Xw, int ldxw, int Nw,
                                    int Kinw, int Mw, const EpiW& ew, hipStream_t st, const AdamPair* ad = nullptr) {
    const DgradPlan d = plan_dgrad(dZd, ldzd, Wd, ldwd, Md, Kind, Nd);
    const WgradPlan w = plan_wgrad(dZw, ldzw, Xw, ldxw, Nw, Kinw, Mw);
    if constexpr (std::is_same<EpiD, EpiMask>::value) {
        if (pair_uses_64x32(Md, Kind)) {                        // hidden-layer input gradient at >= 512 rows
            const GemmGrid g = make_grid(Md, Kind, 64, 32);
            const GemmArgs gd64{dZd, ldzd, Wd, ldwd, Nd, g.tiles_q, g.tiles_p, g.p_per_xcd};
            if (!ga_packable(gd64) || !ga_packable(w.ga) || ga_grid(w.ga) != w.grid) return hipErrorInvalidValue;
            PVAE_LAUNCH((bwd_pair64_kernel<EpiD, EpiW>), dim3(g.grid + w.grid + w.nbias + adam_blocks(ad)), dim3(256), st,
                        PVAE_GA2_PASS(gd64, w.ga), ed, ew, ad ? *ad : AdamPair());
            return hipGetLastError();
        }
    }
    if (!ga_packable(d.ga) || !ga_packable(w.ga) || ga_grid(d.ga) != d.grid || ga_grid(w.ga) != w.grid) return hipErrorInvalidValue;
    PVAE_LAUNCH((bwd_pair_kernel<EpiD, EpiW>), dim3(d.grid + w.grid + w.nbias + adam_blocks(ad)), dim3(256), st,
                       PVAE_GA2_PASS(d.ga, w.ga), ed, ew, ad ? *ad : AdamPair());
    return hipGetLastError();
}
template <class EpiW>
inline hipError_t gemm_bwd_pair(const float* dZd, int ldzd, const float* Wd, int ldwd, const float* mask, int ldm,
                                float* dXd, int ldod, int Md, int Kind, int Nd,
                                const float* dZw, int ldzw, const float* Xw, int ldxw, int Nw, int Kinw, int Mw,
                                const EpiW& ew, hipStream_t st, const AdamPair* ad = nullptr, int act = 1) {
    const EpiMask ed{dXd, ldod, mask, ldm, act};
    return gemm_bwd_pair_epi(dZd, ldzd, Wd, ldwd, Md, Kind, Nd, ed, dZw, ldzw, Xw, ldxw, Nw, Kinw, Mw, ew, st, ad);
}

}  // namespace pvae
