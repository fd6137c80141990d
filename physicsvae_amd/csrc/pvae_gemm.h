// pvae_gemm.h -- fp32 MFMA tile kernel for the three contractions of an MLP layer at
// minibatch scale (M = 32..512 rows).  gfx950 only.
//
//   C[q][p] = sum_k Q(q,k) * P(p,k)            C row-major, p contiguous
//
//   forward  : q = batch row, p = out feature, k = in feature   Q = X  (k-contig)  P = W  (k-contig)
//   dgrad    : q = batch row, p = in feature,  k = out feature  Q = dZ (k-contig)  P = W  (p-contig)
//   wgrad    : q = out feat,  p = in feature,  k = batch row    Q = dZ (q-contig)  P = X  (p-contig)
//
// An operand is "ROW" when its reduction index is the contiguous one in memory and "COL"
// when its output index is.  Either way tiles are fetched with full-line float4 loads along
// the contiguous index and kept in that orientation in LDS, so no transposes are needed:
//   ROW tile  [rows][BK+4]  -> fragment = one ds_read_b64  (2 k-values, conflict-free)
//   COL tile  [BK][cols+8]  -> fragment = two ds_read_b32  (conflict-free)
// One 8-deep k-chunk feeds two v_mfma_f32_16x16x4_f32: lane (i = l&15, h = l>>4) supplies
// k = kk + 2h + s for step s in {0,1}; the k-order inside a chunk is a permutation, which a
// sum does not care about as long as both operands use the same one.
//
// The MFMA "A" slot takes the P fragment and the "B" slot the Q fragment, so D = C^T-tile:
// each lane ends up with 4 *consecutive p* of one q  ->  float4 epilogue loads/stores.
//
// 256 threads = 4 waves in a 2x2 arrangement; each wave owns (BQ/2)x(BP/2) of the tile as
// TQ x TP MFMA tiles.  With a 1x1 wave tile two accumulators alternate over k so the
// dependent-issue latency of the 16x16x4 MFMA (40 cycles vs 32 issue) is covered.
//
// Pipeline: global->register prefetch of tile t+1 is issued before the MFMAs of tile t, the
// registers are written to the other LDS stage afterwards, one barrier per k-tile.
//
// Block -> tile mapping is XCD-aware: block b runs on XCD b%8 (observed dispatch order,
// used for L2 locality only); each XCD is given a contiguous range of p-tiles and all
// q-tiles of it, so the P panel it streams is fetched once per XCD and the (small) Q
// operand stays resident in that XCD's L2.
#pragma once
#include <hip/hip_runtime.h>

namespace pvae {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int BX, int BK, bool ROW>
struct OperandTile {
    static constexpr int kContig = ROW ? BK : BX;       // floats per tile row in memory order
    static constexpr int kRows = ROW ? BX : BK;
    static constexpr int kStride = ROW ? (BK + 4) : (BX + 8);
    static constexpr int kFloats = kRows * kStride;
    static constexpr int kVecs = kRows * kContig / 4;   // float4 per tile
    static constexpr int kPerThread = kVecs / 256;
    static_assert(kVecs % 256 == 0, "tile must split evenly over 256 threads");

    // x0: tile origin along the operand's output index; k0: along the reduction index
    __device__ static inline void load(const float* __restrict__ g, int ld, int x0, int k0, int tid,
                                       v4f (&r)[kPerThread]) {
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int v = tid + 256 * u;
            const int row = v / (kContig / 4);
            const int c4 = v % (kContig / 4);
            const size_t off = ROW ? ((size_t)(x0 + row) * ld + k0 + c4 * 4)
                                   : ((size_t)(k0 + row) * ld + x0 + c4 * 4);
            r[u] = *reinterpret_cast<const v4f*>(g + off);
        }
    }
    __device__ static inline void store(float* s, int tid, const v4f (&r)[kPerThread]) {
#pragma unroll
        for (int u = 0; u < kPerThread; ++u) {
            const int v = tid + 256 * u;
            const int row = v / (kContig / 4);
            const int c4 = v % (kContig / 4);
            *reinterpret_cast<v4f*>(s + row * kStride + c4 * 4) = r[u];
        }
    }
    // fragment for the 16 outputs starting at x (tile-local), k-chunk kk, lane (i,h)
    __device__ static inline v2f frag(const float* s, int x, int kk, int i, int h) {
        if (ROW) {
            return *reinterpret_cast<const v2f*>(s + (x + i) * kStride + kk + 2 * h);
        } else {
            v2f f;
            f.x = s[(kk + 2 * h) * kStride + x + i];
            f.y = s[(kk + 2 * h + 1) * kStride + x + i];
            return f;
        }
    }
};

template <int BQ, int BP, int BK, bool Q_ROW, bool P_ROW, class Epi>
__global__ void __launch_bounds__(256)
gemm_tile_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ P, int ldp, int K,
                 int tiles_q, int tiles_p, int p_per_xcd, Epi epi) {
    using QT = OperandTile<BQ, BK, Q_ROW>;
    using PT = OperandTile<BP, BK, P_ROW>;
    constexpr int TQ = BQ / 32, TP = BP / 32;
    constexpr int NACC = (TQ * TP == 1) ? 2 : 1;

    const int bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3;
    const int tile_p = xcd * p_per_xcd + loc / tiles_q;
    const int tile_q = loc % tiles_q;
    if (tile_p >= tiles_p) return;
    const int q0 = tile_q * BQ, p0 = tile_p * BP;

    __shared__ __attribute__((aligned(16))) float lds[2 * (QT::kFloats + PT::kFloats)];
    constexpr int kStage = QT::kFloats + PT::kFloats;     // one pipeline stage: [Q tile | P tile]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lh = lane >> 4;
    const int wq = (wave >> 1) * (BQ / 2), wp = (wave & 1) * (BP / 2);

    v4f acc[TQ][TP][NACC];
#pragma unroll
    for (int a = 0; a < TQ; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[a][b][c] = v4f{0.f, 0.f, 0.f, 0.f};

    v4f rq[QT::kPerThread], rp[PT::kPerThread];
    const int nk = K / BK;

    QT::load(Q, ldq, q0, 0, tid, rq);
    PT::load(P, ldp, p0, 0, tid, rp);
    QT::store(lds, tid, rq);
    PT::store(lds + QT::kFloats, tid, rp);
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) {
            QT::load(Q, ldq, q0, (t + 1) * BK, tid, rq);
            PT::load(P, ldp, p0, (t + 1) * BK, tid, rp);
        }
        const float* cq = lds + cur * kStage;
        const float* cp = cq + QT::kFloats;
#pragma unroll
        for (int kk = 0; kk < BK; kk += 8) {
            v2f fq[TQ], fp[TP];
#pragma unroll
            for (int a = 0; a < TQ; ++a) fq[a] = QT::frag(cq, wq + a * 16, kk, li, lh);
#pragma unroll
            for (int b = 0; b < TP; ++b) fp[b] = PT::frag(cp, wp + b * 16, kk, li, lh);
#pragma unroll
            for (int a = 0; a < TQ; ++a)
#pragma unroll
                for (int b = 0; b < TP; ++b) {
                    acc[a][b][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fp[b].x, fq[a].x,
                                                                         acc[a][b][0], 0, 0, 0);
                    acc[a][b][NACC - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        fp[b].y, fq[a].y, acc[a][b][NACC - 1], 0, 0, 0);
                }
        }
        if (t + 1 < nk) {
            float* nq = lds + (cur ^ 1) * kStage;
            QT::store(nq, tid, rq);
            PT::store(nq + QT::kFloats, tid, rp);
        }
        __syncthreads();
    }

    // D[i = p-local][j = q-local]: lane holds q = li, p = 4*lh + r (r = 0..3)
#pragma unroll
    for (int a = 0; a < TQ; ++a)
#pragma unroll
        for (int b = 0; b < TP; ++b) {
            v4f v = acc[a][b][0];
            if (NACC == 2) v += acc[a][b][NACC - 1];
            epi(q0 + wq + a * 16 + li, p0 + wp + b * 16 + 4 * lh, v);
        }
}

// ---------------------------------------------------------------------------------------
// epilogues: (q, p, 4 consecutive p values)
// ---------------------------------------------------------------------------------------
struct EpiBiasAct {           // forward layer: out = act(acc + bias)
    float* out;
    int ldo;
    const float* bias;        // may be null
    int relu;
    __device__ inline void operator()(int q, int p, v4f v) const {
        if (bias) v += *reinterpret_cast<const v4f*>(bias + p);
        if (relu) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
            v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        *reinterpret_cast<v4f*>(out + (size_t)q * ldo + p) = v;
    }
};

struct EpiMask {              // input gradient: out = acc * (act > 0)
    float* out;
    int ldo;
    const float* mask;        // post-ReLU activation of the producing layer, or null
    int ldm;
    __device__ inline void operator()(int q, int p, v4f v) const {
        if (mask) {
            const v4f m = *reinterpret_cast<const v4f*>(mask + (size_t)q * ldm + p);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f;
            v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<v4f*>(out + (size_t)q * ldo + p) = v;
    }
};

struct AdamScalars {
    float step_size;          // lr / (1 - beta1^t)
    float bc2_sqrt;           // sqrt(1 - beta2^t)
    float beta1, beta2, eps;
};

// torch.optim.Adam single-tensor update (amsgrad False, weight_decay 0), tm:119-122,143:
//   m <- m + (g - m)(1 - b1) ; v <- v b2 + (1 - b2) g g ; p <- p - step_size * m / (sqrt(v)/bc2_sqrt + eps)
__device__ inline void adam_update(float g, float& p, float& m, float& v, const AdamScalars& s) {
    // every operation is pinned (no context-dependent fma contraction), so the fused
    // epilogue and the flat multi-tensor kernel produce bit-identical parameters
    m = __fmaf_rn(__fsub_rn(g, m), __fsub_rn(1.0f, s.beta1), m);
    v = __fmaf_rn(v, s.beta2, __fmul_rn(__fmul_rn(__fsub_rn(1.0f, s.beta2), g), g));
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), s.bc2_sqrt), s.eps);
    p = __fmaf_rn(-s.step_size, __fdiv_rn(m, denom), p);
}

__device__ inline void adam_update4(const v4f& g, v4f& p, v4f& m, v4f& v, const AdamScalars& s) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        adam_update(g[e], pe, me, ve, s);
        p[e] = pe; m[e] = me; v[e] = ve;
    }
}

struct EpiGradStore {         // weight gradient -> gradient arena (data-parallel path)
    float* g;
    int ld;
    __device__ inline void operator()(int q, int p, v4f v) const {
        *reinterpret_cast<v4f*>(g + (size_t)q * ld + p) = v;
    }
};

struct EpiGradAdam {          // weight gradient consumed in registers by Adam (1-GPU path)
    float* w;
    float* m;
    float* v;
    int ld;
    AdamScalars s;
    __device__ inline void operator()(int q, int p, v4f g) const {
        const size_t o = (size_t)q * ld + p;
        v4f pw = *reinterpret_cast<v4f*>(w + o);
        v4f pm = *reinterpret_cast<v4f*>(m + o);
        v4f pv = *reinterpret_cast<v4f*>(v + o);
        adam_update4(g, pw, pm, pv, s);
        *reinterpret_cast<v4f*>(w + o) = pw;
        *reinterpret_cast<v4f*>(m + o) = pm;
        *reinterpret_cast<v4f*>(v + o) = pv;
    }
};

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
template <int BQ, int BP, int BK, bool Q_ROW, bool P_ROW, class Epi>
inline hipError_t launch_gemm(const float* Q, int ldq, const float* P, int ldp, int rows_q, int cols_p,
                              int K, const Epi& epi, hipStream_t st) {
    const int tiles_q = rows_q / BQ, tiles_p = cols_p / BP;
    const int p_per_xcd = (tiles_p + 7) / 8;
    const int grid = 8 * p_per_xcd * tiles_q;
    hipLaunchKernelGGL((gemm_tile_kernel<BQ, BP, BK, Q_ROW, P_ROW, Epi>), dim3(grid), dim3(256), 0, st,
                       Q, ldq, P, ldp, K, tiles_q, tiles_p, p_per_xcd, epi);
    return hipGetLastError();
}

// forward: out[M][N] = act(X[M][K] W[N][K]^T + b)
inline hipError_t gemm_forward(const float* X, int ldx, const float* W, int ldw, const float* bias,
                               float* out, int ldo, int M, int N, int K, int relu, hipStream_t st) {
    EpiBiasAct e{out, ldo, bias, relu};
    return launch_gemm<32, 32, 64, true, true>(X, ldx, W, ldw, M, N, K, e, st);
}
// dgrad: dX[M][Kin] = (dZ[M][N] W[N][Kin]) .* (mask > 0)
inline hipError_t gemm_dgrad(const float* dZ, int ldz, const float* W, int ldw, const float* mask,
                             int ldm, float* dX, int ldo, int M, int Kin, int N, hipStream_t st) {
    EpiMask e{dX, ldo, mask, ldm};
    return launch_gemm<32, 32, 64, true, false>(dZ, ldz, W, ldw, M, Kin, N, e, st);
}
// wgrad: G[N][Kin] = dZ[M][N]^T X[M][Kin]
template <class Epi>
inline hipError_t gemm_wgrad(const float* dZ, int ldz, const float* X, int ldx, int N, int Kin, int M,
                             const Epi& e, hipStream_t st) {
    return launch_gemm<64, 64, 32, false, false>(dZ, ldz, X, ldx, N, Kin, M, e, st);
}

}  // namespace pvae
