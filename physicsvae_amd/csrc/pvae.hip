// pvae.hip -- libpvae_gfx950.so: C-ABI (include/pvae.h) + glue kernels around the MFMA
// tile kernel of pvae_gemm.h.  gfx950 (MI355X) only; built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared pvae.hip -o libpvae_gfx950.so
//
// Reference lines restated by each kernel are cited at the kernel (tpv / tm / rmt as in
// include/pvae.h).
#include "pvae_internal.h"

// ---------------------------------------------------------------------------------------
// glue kernels
// ---------------------------------------------------------------------------------------

// Minibatch staging as a launch of its own: one block per (padded) batch row and time step
// (blockIdx.y = t < L); the work is stage_row (pvae_gemm.h).
// (four rows per workgroup, one wave each: the row's sources land in LDS by LDS-DMA and the panels are written with whole
//  16-byte stores, stage_row_lds; rows too wide for the wave's LDS: stage_row_wave, branch-free dword-granular buffer
//  accesses.  Round 3's scalar-load form measured 14.2 / 30.3 us per 8192 windows against 14.0 / 20.0: docs/experiments.md)
// per-wave LDS of the gather: the smallest power of two that holds a row's sources + one DMA group of slack (2 KB at the
// configs[2] dims, 4 KB at configs[4]'s: all eight workgroups a CU can hold are resident at once); 0: rows too wide
static inline int stage_lds_floats(int Db, int Da) {
    const int need = 2 * Db + Da + 64;
    if (need > kStageLdsFloats) return 0;
    int n = 256;
    while (n < need) n <<= 1;
    return n;
}
__global__ void __launch_bounds__(256) stage_batch_kernel(StageArgs a, int lds_floats) {
    extern __shared__ float stage_lds[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                      // (provably wave-uniform: the row's
    const int r = blockIdx.x * 4 + w;                                                    //  descriptors / LDS base live in SGPRs)
    if (r >= a.rows_pad) return;
    if (lds_floats > 0) stage_row_lds(a, r, blockIdx.y, a.rows_pad, threadIdx.x & 63, stage_lds + w * lds_floats, lds_floats - 1);
    else stage_row_wave(a, r, blockIdx.y, a.rows_pad, threadIdx.x & 63);
}

// s1 of step t+1 = world-model prediction of step t (tpv:421): copy the first Db columns of the
// valid rows of `src` into the current-state columns of up to four input panels.
__global__ void __launch_bounds__(256)
scatter_state_kernel(const float* __restrict__ src, int lds_, int rows, int Db, float* __restrict__ d0, int ld0,
                     float* __restrict__ d1, int ld1, float* __restrict__ d2, int ld2, float* __restrict__ d3,
                     int ld3) {
    const int total = rows * Db;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / Db, c = idx - r * Db;
        const float v = src[(size_t)r * lds_ + c];
        if (d0) d0[(size_t)r * ld0 + c] = v;
        if (d1) d1[(size_t)r * ld1 + c] = v;
        if (d2) d2[(size_t)r * ld2 + c] = v;
        if (d3) d3[(size_t)r * ld3 + c] = v;
    }
}

// dst[r][c] += s0[r][c] (+ s1 + s2 + s3), c < n, r < rows: the gradient wrt the state handed from
// step t to step t+1 is the sum of what came back through every consumer of that state (encoder,
// decoder and the world-model invocations of step t+1).  Fixed summation order.
__global__ void __launch_bounds__(256)
add_cols_kernel(float* __restrict__ dst, int ldd, int rows, int n, const float* __restrict__ s0, int l0,
                const float* __restrict__ s1, int l1, const float* __restrict__ s2, int l2,
                const float* __restrict__ s3, int l3) {
    const int total = rows * n;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / n, c = idx - r * n;
        float v = dst[(size_t)r * ldd + c];
        if (s0) v += s0[(size_t)r * l0 + c];
        if (s1) v += s1[(size_t)r * l1 + c];
        if (s2) v += s2[(size_t)r * l2 + c];
        if (s3) v += s3[(size_t)r * l3 + c];
        dst[(size_t)r * ldd + c] = v;
    }
}

__device__ inline float block_sum_256(float v) {
    __shared__ float red[4];
    return block_sum_256(v, red);
}

// The motor decoder's helper (rmt:833-835): a_hat[:, :Da] += range * h, h = the helper stack's tanh output.  The decoder's
// output layer has already left a second copy of its own a_hat in the action columns of the world model's input panel.
__global__ void __launch_bounds__(256)
helper_add_kernel(float* __restrict__ a_hat, int lda, const float* __restrict__ h, int ldh, float* __restrict__ wm_in,
                  int ldw, int col0, int rows, int Da, float range) {
    const int total = rows * Da;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / Da, c = idx - r * Da;
        const float v = __fmaf_rn(range, h[(size_t)r * ldh + c], a_hat[(size_t)r * lda + c]);
        a_hat[(size_t)r * lda + c] = v;
        if (wm_in) wm_in[(size_t)r * ldw + col0 + c] = v;
    }
}
// ... and its backward: the gradient wrt the action reaches the helper's pre-activation through range * tanh'
// (dz_h = range * (1 - h^2) * d a_hat); pad rows / columns of the panel are written as zeros.
__global__ void __launch_bounds__(256)
helper_seed_kernel(const float* __restrict__ dz_a, int lda, const float* __restrict__ h, float* __restrict__ dz_h, int ldh,
                   int rows, int rows_pad, int Da, float range) {
    const int total = rows_pad * ldh;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / ldh, c = idx - r * ldh;
        float g = 0.f;
        if (r < rows && c < Da) {
            const float t = h[idx];
            g = range * (1.0f - t * t) * dz_a[(size_t)r * lda + c];
        }
        dz_h[idx] = g;
    }
}

// nn.MSELoss (tm:99; or nn.L1Loss, tm:100-101, when l1) of pred vs target over rows x D, plus its
// gradient:
//   partial[b] = sum (pred - target)^2 over this block's rows      (finalize scales by 1/(B*D))
//   dz = grad_scale * (pred - target) [+ extra]                    grad_scale = coeff*2/(B*D)
//   L1: partial = sum |pred - target|, dz = grad_scale * sign(pred - target), grad_scale = coeff/(B*D)
// Used for the world-model MSE (tpv:411-414), the cycle loss (tpv:417-419) and the action
// reconstruction loss (tpv:381-382; `extra` = gradient arriving through the frozen world
// model, columns [Db, Db+Da) of d(wm_in)).
__global__ void __launch_bounds__(256)
mse_grad_kernel(const float* __restrict__ pred, int ldp, const float* __restrict__ target, int ldt,
                float* __restrict__ dz, int ldz, int rows, int rows_pad, int D, float grad_scale,
                const float* __restrict__ extra, int lde, int extra_col0, float* __restrict__ partial, int l1) {
    float acc = 0.f;
    for (int r = blockIdx.x; r < rows_pad; r += gridDim.x) {
        const bool valid = r < rows;
        for (int c = threadIdx.x; c < ldz; c += 256) {
            float g = 0.f;
            if (valid && c < D) {
                const float d = pred[(size_t)r * ldp + c] - target[(size_t)r * ldt + c];
                acc += l1 ? fabsf(d) : d * d;
                g = grad_scale * (l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : d);
                if (extra) g += extra[(size_t)r * lde + extra_col0 + c];
            }
            if (dz) dz[(size_t)r * ldz + c] = g;
        }
    }
    const float s = block_sum_256(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// Reparameterisation sampler + KL-to-N(0,I) partial sums (rmt:734-740, 795-800; tpv:384-389):
//   z = mu + eps * exp(0.5 logvar)      written into md_in[:, Db:Db+Z]
//   partial[b] = sum -0.5 (1 + logvar - mu^2 - exp(logvar))     (finalize scales by 1/B)
__global__ void __launch_bounds__(256)
reparam_kernel(const float* __restrict__ te_out, int ldte, const float* __restrict__ eps_in,
               float* __restrict__ eps_used, float* __restrict__ md_in, int ld_md, int Db, int Z, int rows,
               int rows_pad, int noise, unsigned long long seed, unsigned long long offset,
               float* __restrict__ partial, float* __restrict__ z_dense,
               const float* __restrict__ mu_p = nullptr, int ldmp = 0) {
    float acc = 0.f;
    const int total = rows_pad * Z;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / Z, c = idx - r * Z;
        float z = 0.f, e = 0.f;
        if (r < rows) {
            const float mu = te_out[(size_t)r * ldte + c];
            const float lv = te_out[(size_t)r * ldte + Z + c];
            if (noise) e = eps_in ? eps_in[(size_t)r * Z + c] : philox_normal(seed, offset, r, c);
            z = __fmaf_rn(e, expf(0.5f * lv), mu);
            if (mu_p) {               // KL(N(mu, s^2) || N(mu_p, 1)), oracle/refpath.py PRIORS
                const float d = mu - mu_p[(size_t)r * ldmp + c];
                acc += 0.5f * (expf(lv) + d * d - 1.0f - lv);
            } else {
                acc += -0.5f * (1.0f + lv - mu * mu - expf(lv));
            }
        }
        md_in[(size_t)r * ld_md + Db + c] = z;
        eps_used[(size_t)r * Z + c] = e;
        if (z_dense && r < rows) z_dense[(size_t)r * Z + c] = z;      // caller's [rows][Z] copy (rollout path)
    }
    const float s = block_sum_256(acc);
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = s;
}

// The same sampler as the PROLOGUE of the decoder's first-layer launch (pvae_gemm.h, splitk_ws_body / NoPro): every
// workgroup of that launch forms z for its own 32 batch rows while its first k-tile is in flight and patches it over
// the z columns of its input tile in LDS; the workgroups of column tile 0 also store z (the decoder's first-layer
// weight gradient reads it from the input panel), the draws actually used, and the KL partial of their row block.
// One launch less per joint step (the sampler launch was ~4.5 us of fixed cost for 8 K elements).
struct ProSampler {
    static constexpr bool kActive = true;
    static constexpr int kMaxZ = 64, kScratchFloats = 8;
    static constexpr int kPer = 32 * kMaxZ / 4 / 256;     // work items per thread at Z = kMaxZ
    const float* te_out; int ldte;       // encoder output [mu | logvar]
    const float* eps_in;                 // supplied draws [rows][Z], or null: Philox
    float* eps_used;                     // [rows_pad][Z]
    float* md_in; int ld_md;             // the decoder's input panel: z columns written by column tile 0
    int c0, Z, rows, noise;              // z columns = [c0, c0 + Z), Z % 4 == 0
    unsigned long long seed, offset;
    float* partial;                      // KL partial per row block (tiles_q of them), or null
    // work item = 4 consecutive z columns of one row (one Philox call, 16-byte accesses): item e of the workgroup is
    // row e / (Z/4), columns 4 (e % (Z/4)) ..; thread tid owns items tid, tid + 256, ...
    struct State { v4f mu[kPer], lv[kPer], ep[kPer], z[kPer]; bool formed; };
    __device__ inline bool needs(int t) const { return t == (c0 >> 6) || t == ((c0 + Z - 1) >> 6); }
    // request the inputs (the encoder's output was written through by the previous launch: cold fetches, which now
    // travel while the k-tiles in front of the z columns are contracted)
    __device__ inline State prepare(int q0, int tid) const {
        State st;
        const int G = Z >> 2;
        const float* __restrict__ te = te_out;
        const float* __restrict__ ei = eps_in;
        const v4f zero = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + 256 * u;
            const int r = e / G, j = (e - r * G) * 4, q = q0 + r;
            const bool live = e < 32 * G && q < rows;
            st.mu[u] = live ? *reinterpret_cast<const v4f*>(te + (size_t)q * ldte + j) : zero;
            st.lv[u] = live ? *reinterpret_cast<const v4f*>(te + (size_t)q * ldte + Z + j) : zero;
            st.ep[u] = (live && noise && ei) ? *reinterpret_cast<const v4f*>(ei + (size_t)q * Z + j) : zero;
            st.z[u] = zero;
        }
        st.formed = false;
        return st;
    }
    // tile = the swizzled [32][64] image of k-tile t (chunk ^= row & 15, as the loaders write it)
    __device__ inline void patch(float* tile, float* scratch, State& st, int t, int q0, int tile_p, int, int tid) const {
        const int G = Z >> 2;
        if (!st.formed) {                 // first patched tile: form z, KL partial, and (column tile 0) store z and the draws
            float acc = 0.f;
#pragma unroll
            for (int u = 0; u < kPer; ++u) {
                const int e = tid + 256 * u;
                if (e >= 32 * G) break;
                const int r = e / G, j = (e - r * G) * 4, q = q0 + r;
                v4f ee = v4f{0.f, 0.f, 0.f, 0.f};
                if (q < rows) {
                    if (noise) ee = eps_in ? st.ep[u] : philox_normal4(seed, offset, q, j >> 2);
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        st.z[u][x] = __fmaf_rn(ee[x], expf(0.5f * st.lv[u][x]), st.mu[u][x]);
                        acc += -0.5f * (1.0f + st.lv[u][x] - st.mu[u][x] * st.mu[u][x] - expf(st.lv[u][x]));
                    }
                }
                if (tile_p == 0) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) md_in[(size_t)q * ld_md + c0 + j + x] = st.z[u][x];      // (c0 = dim_body: unaligned)
                    *reinterpret_cast<v4f*>(eps_used + (size_t)q * Z + j) = ee;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if ((tid & 63) == 0) scratch[tid >> 6] = acc;
            st.formed = true;
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int e = tid + 256 * u;
            if (e >= 32 * G) break;
            const int r = e / G, j = (e - r * G) * 4;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int c = c0 + j + x;
                if ((c >> 6) != t) continue;
                const int kc = c & 63;
                tile[r * 64 + ((((kc >> 2) ^ (r & 15))) << 2) + (kc & 3)] = st.z[u][x];
            }
        }
    }
    // behind the barrier that follows the first patch: the four compute waves' KL sums are in the scratch
    __device__ inline void publish(const float* scratch, int t, int tile_p, int tile_q, int tid) const {
        if (t == (c0 >> 6) && tile_p == 0 && tid == 0 && partial)
            partial[tile_q] = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    }
};

// Backward of the sampler + KL (autograd of rmt:734-740 and tpv:388):
//   dmu = dz + (beta/B) mu ;  dlogvar = dz * eps * 0.5 exp(0.5 lv) + (beta/B) 0.5 (exp(lv) - 1)
__global__ void __launch_bounds__(256)
reparam_bwd_kernel(const float* __restrict__ d_md_in, int ld_md, int Db, const float* __restrict__ te_out,
                   int ldte, const float* __restrict__ eps_used, float* __restrict__ dz_te, int ld_dz,
                   int rows, int rows_pad, int Z, float kl_scale, const float* __restrict__ mu_p = nullptr,
                   int ldmp = 0, float* __restrict__ dz_p = nullptr, int ldzp = 0) {
    const int total = rows_pad * ld_dz;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / ld_dz, c = idx - r * ld_dz;
        float g = 0.f;
        if (r < rows && c < 2 * Z) {
            const int cz = c < Z ? c : c - Z;
            const float dzv = d_md_in[(size_t)r * ld_md + Db + cz];
            const float mu = te_out[(size_t)r * ldte + cz];
            const float lv = te_out[(size_t)r * ldte + Z + cz];
            if (c < Z) {
                if (mu_p) {
                    const float gp = kl_scale * (mu - mu_p[(size_t)r * ldmp + cz]);
                    g = dzv + gp;
                    if (dz_p) dz_p[(size_t)r * ldzp + cz] = -gp;
                } else {
                    g = dzv + kl_scale * mu;
                }
            } else {
                const float e = eps_used[(size_t)r * Z + cz];
                g = dzv * e * 0.5f * expf(0.5f * lv) + kl_scale * 0.5f * (expf(lv) - 1.0f);
            }
        }
        dz_te[idx] = g;
        if (dz_p && c < Z && r >= rows) dz_p[(size_t)r * ldzp + c] = 0.f;
    }
}

// PVAE_PRIOR_HYPERSPHERE (oracle/refpath.py PRIORS; rmt:810-814, tpv:404-407): the encoder's Z outputs
// e are projected onto the unit sphere, z = e / max(|e|, 1e-12) (F.normalize), z goes to the decoder;
// the prior sample of this forward is u = n / max(|n|, 1e-12), n ~ N(0, I) (the supplied eps, or Philox),
// and the KL slot of the loss is mean_i <z_i, u_i>.  One wave per row.
//   md_in[:, Db:Db+Z] = z     eps_used = u (zeros without noise)     partial[b] = sum over its rows of <z, u>
__global__ void __launch_bounds__(256)
sphere_kernel(const float* __restrict__ te_out, int ldte, const float* __restrict__ eps_in,
              float* __restrict__ eps_used, float* __restrict__ md_in, int ld_md, int Db, int Z, int rows,
              int rows_pad, int noise, unsigned long long seed, unsigned long long offset,
              float* __restrict__ partial, float* __restrict__ z_dense, int normalize) {
    // normalize == 0: latent_prior_type = False (rmt:815-816) -- z = e, nothing sampled, no loss term
    if (!normalize) noise = 0;
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    float dot = 0.f;
    if (r < rows_pad) {
        float e2 = 0.f, n2 = 0.f;
        for (int c = lane; c < Z; c += 64) {
            if (r < rows) {
                const float e = te_out[(size_t)r * ldte + c];
                e2 += e * e;
                if (noise) {
                    const float nz = eps_in ? eps_in[(size_t)r * Z + c] : philox_normal(seed, offset, r, c);
                    n2 += nz * nz;
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { e2 += __shfl_xor(e2, o, 64); n2 += __shfl_xor(n2, o, 64); }
        const float ie = normalize ? 1.0f / fmaxf(sqrtf(e2), 1e-12f) : 1.0f, in_ = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        for (int c = lane; c < Z; c += 64) {
            float z = 0.f, u = 0.f;
            if (r < rows) {
                z = te_out[(size_t)r * ldte + c] * ie;
                if (noise) u = (eps_in ? eps_in[(size_t)r * Z + c] : philox_normal(seed, offset, r, c)) * in_;
                dot += z * u;
                if (z_dense) z_dense[(size_t)r * Z + c] = z;
            }
            md_in[(size_t)r * ld_md + Db + c] = z;
            eps_used[(size_t)r * Z + c] = u;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
    }
    if (lane == 0) part[wave] = dot;
    __syncthreads();
    if (threadIdx.x == 0 && partial) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// its backward: g = dL/dz = (what came back through the decoder) + (beta/B) u;  dL/de = (g - z <z, g>) / |e|
__global__ void __launch_bounds__(256)
sphere_bwd_kernel(const float* __restrict__ d_md_in, int ld_md, int Db, const float* __restrict__ te_out, int ldte,
                  const float* __restrict__ u_used, float* __restrict__ dz_te, int ld_dz, int rows, int rows_pad,
                  int Z, float kl_scale, int normalize) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + wave;
    if (r >= rows_pad) return;
    float e2 = 0.f, zg = 0.f;
    if (r < rows)
        for (int c = lane; c < Z; c += 64) {
            const float e = te_out[(size_t)r * ldte + c];
            e2 += e * e;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) e2 += __shfl_xor(e2, o, 64);
    const float ie = 1.0f / fmaxf(sqrtf(e2), 1e-12f);
    if (!normalize) {                          // z = e: the decoder's input gradient is the encoder's output gradient
        for (int c = lane; c < ld_dz; c += 64)
            dz_te[(size_t)r * ld_dz + c] = (r < rows && c < Z) ? d_md_in[(size_t)r * ld_md + Db + c] : 0.f;
        return;
    }
    if (r < rows)
        for (int c = lane; c < Z; c += 64) {
            const float g = d_md_in[(size_t)r * ld_md + Db + c] + kl_scale * u_used[(size_t)r * Z + c];
            zg += te_out[(size_t)r * ldte + c] * ie * g;
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) zg += __shfl_xor(zg, o, 64);
    for (int c = lane; c < ld_dz; c += 64) {
        float d = 0.f;
        if (r < rows && c < Z) {
            const float g = d_md_in[(size_t)r * ld_md + Db + c] + kl_scale * u_used[(size_t)r * Z + c];
            d = (g - te_out[(size_t)r * ldte + c] * ie * zg) * ie;
        }
        dz_te[(size_t)r * ld_dz + c] = d;
    }
}

// dst[r][dst_col0 + c] = src[r][src_col0 + c]
__global__ void __launch_bounds__(256)
copy_cols_kernel(const float* __restrict__ src, int lds_, int src_col0, float* __restrict__ dst, int ldd,
                 int dst_col0, int rows, int ncols) {
    const int total = rows * ncols;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / ncols, c = idx - r * ncols;
        dst[(size_t)r * ldd + dst_col0 + c] = src[(size_t)r * lds_ + src_col0 + c];
    }
}

// dst[rows_pad][ld] = zero-padded copy of dense src[rows][n]; only the columns [c0, c0 + nw) of src are taken (the
// window a first layer with an input subset reads: whatever the caller put in the other columns meets structural-zero
// weights, and the panel keeps exact zeros there like a staged one)
__global__ void __launch_bounds__(256)
pad_copy_kernel(const float* __restrict__ src, int n, int rows, float* __restrict__ dst, int ld, int rows_pad, int c0, int nw) {
    const int total = rows_pad * ld;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / ld, c = idx - r * ld;
        dst[idx] = (r < rows && c < n && c >= c0 && c < c0 + nw) ? src[(size_t)r * n + c] : 0.f;
    }
}

// Evaluation-only finalisation (training folds it into the last weight-gradient launch).
__global__ void finalize_loss_kernel(LossFinal f) { finalize_loss_wave(f, threadIdx.x); }

// Multi-tensor Adam over one contiguous arena segment (data-parallel path, after the
// gradient all-reduce).  28 B/param of traffic: read p,g,m,v, write p,m,v.
__global__ void __launch_bounds__(256)
adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                 float* __restrict__ v, long long n4, AdamScalars s) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += gridDim.x * 256ll) {
        v4f pp = reinterpret_cast<v4f*>(p)[i];
        const v4f gg = reinterpret_cast<const v4f*>(g)[i];
        v4f mm = reinterpret_cast<v4f*>(m)[i];
        v4f vv = reinterpret_cast<v4f*>(v)[i];
        adam_update4(gg, pp, mm, vv, s);
        reinterpret_cast<v4f*>(p)[i] = pp;
        reinterpret_cast<v4f*>(m)[i] = mm;
        reinterpret_cast<v4f*>(v)[i] = vv;
    }
}

// Rollout-batch forward layer (rows <= 4; rmt:742-771 runs at B = 1 inside the 30 Hz control
// loop): out[r][n] = act(sum_k x[r][k] W[n][k] + b[n]).  One wave per output feature streams its
// weight row once with float4 loads (all 256 CUs busy: n_out/4 blocks of 4 waves), the R input
// rows come from L1/L2, lanes split K and combine with a shuffle tree.  HBM/L2-bound: 4 B per
// weight, ~2 flops per byte -- the tile kernels would push the same panel through 32 workgroups.
template <int R>
__global__ void __launch_bounds__(256)
gemv_rows_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ W, int ldw,
                 const float* __restrict__ bias, float* __restrict__ out, int ldo, int K, int relu,
                 float* __restrict__ out2, int ld2, int off2, int n2, int n_valid) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const float* wrow = W + (size_t)n * ldw;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {
        const v4f wv = *reinterpret_cast<const v4f*>(wrow + k);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const v4f xv = *reinterpret_cast<const v4f*>(x + (size_t)r * ldx + k);
            acc[r] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[r]))));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) {
            v += bias[n];
            v = (relu > 1 && n >= n_valid) ? 0.f : act_apply(v, relu);
            out[(size_t)r * ldo + n] = v;
            if (out2 && n < n2) out2[(size_t)r * ld2 + off2 + n] = v;
        }
    }
}

// Rollout forward with fewer launches (pvae_infer at <= 4 rows): the layer kernel assembles its R input
// rows in LDS itself, so the staging launch, the sampler launch and the copy-out launches disappear --
// 7 launches for observation -> action (TE 3, MD 4 at the trainer's default sizes) instead of 9, 10 with
// the world model's prediction instead of 14.  The input of a layer is
//   kind 0: rows of a padded activation panel (hidden layers)
//   kind 1: the caller's dense observation rows obs[r][0:Ka]                       (first encoder layer)
//   kind 2: [obs[r][0:Ka] | z_r],  z = mu + eps * exp(logvar / 2) from the encoder's output   (first decoder layer:
//           the sampler of rmt:734-740 runs here; workgroup 0 also records z and the draws)
//   kind 3: [obs[r][0:Ka] | src_b[r][0:Kb]]                                        (first world-model layer: a_hat)
//   kinds 4 / 5: [obs[r][0:Ka] | e_r] resp. [obs | e_r / |e_r|], e = the encoder's Z outputs (latent_prior_type False /
//           hypersphere_uniform: what sphere_kernel computes on the training path)
// One wave per output feature streams its weight row once (as gemv_rows_kernel); rows >= `rows` of the
// R-row template are computed on zeros and never stored.
struct RolloutIn {
    int kind;
    const float* a; int lda, Ka;      // panel (kind 0: Ka = padded width) or dense observation
    const float* b; int ldb, Kb;      // kind 2: encoder output [mu | logvar] (Kb = Z); kind 3: second source
    const float* eps; int noise;      // kind 2: supplied draws [rows][Z] or null (Philox) / noise off
    unsigned long long seed, offset;
    float* z_out; float* eps_used;    // kind 2, written by workgroup 0 (z_out may be null)
    float* keep;                      // kind 1: workgroup 0 copies the observation rows here ([rows][Ka]; may be null)
};
template <int R>
__global__ void __launch_bounds__(256)
gemv_rollout_kernel(RolloutIn in, int rows, const float* __restrict__ W, int ldw, const float* __restrict__ bias,
                    float* __restrict__ out, int ldo, int K, int relu, float* __restrict__ out2, int ld2, int n2,
                    int n_valid, const float* __restrict__ ls) {
    extern __shared__ __attribute__((aligned(16))) float xs[];         // [R][K], K = ld of the layer (multiple of 64)
    const int tid = threadIdx.x;
    // this wave's weight row: the first 1024 columns are requested BEFORE the input rows are assembled,
    // so that the two memory latencies of a layer (inputs, weights) overlap instead of adding up
    const int n = blockIdx.x * 4 + (tid >> 6);
    const int lane = tid & 63;
    const float* wrow = W + (size_t)n * ldw;
    v4f wpre[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane * 4 + 256 * j;
        wpre[j] = k < K ? *reinterpret_cast<const v4f*>(wrow + k) : v4f{0.f, 0.f, 0.f, 0.f};
    }
    if (in.kind == 0) {                   // hidden layers: whole padded panel rows, 16 bytes per load
        const int kq = K >> 2;
        for (int i = tid; i < R * kq; i += 256) {
            const int r = i / kq, k = (i - r * kq) * 4;
            *reinterpret_cast<v4f*>(xs + r * K + k) =
                r < rows ? *reinterpret_cast<const v4f*>(in.a + (size_t)r * in.lda + k) : v4f{0.f, 0.f, 0.f, 0.f};
        }
    }
    for (int i = tid; in.kind != 0 && i < R * K; i += 256) {
        const int r = i / K, k = i - r * K;
        float v = 0.f;
        if (r < rows) {
            if (k < in.Ka) {
                v = in.a[(size_t)r * in.lda + k];
                if (in.kind == 1 && in.keep && blockIdx.x == 0) in.keep[(size_t)r * in.Ka + k] = v;
            } else if (k < in.Ka + in.Kb) {
                const int j = k - in.Ka;
                if (in.kind == 2) {
                    const float mu = in.b[(size_t)r * in.ldb + j], lv = in.b[(size_t)r * in.ldb + in.Kb + j];
                    float e = 0.f;
                    if (in.noise) e = in.eps ? in.eps[(size_t)r * in.Kb + j] : philox_normal(in.seed, in.offset, r, j);
                    v = mu + e * expf(0.5f * lv);
                    if (blockIdx.x == 0) {
                        if (in.z_out) in.z_out[(size_t)r * in.Kb + j] = v;
                        in.eps_used[(size_t)r * in.Kb + j] = e;
                    }
                } else if (in.kind == 3) {
                    v = in.b[(size_t)r * in.ldb + j];
                } else if (in.kind == 4) {          // latent_prior_type False: the encoder's outputs are the code
                    v = in.b[(size_t)r * in.ldb + j];
                    if (blockIdx.x == 0) {
                        if (in.z_out) in.z_out[(size_t)r * in.Kb + j] = v;
                        in.eps_used[(size_t)r * in.Kb + j] = 0.f;
                    }
                } else if (in.kind == 5) {          // hypersphere: z = e / max(|e|, 1e-12) (sphere_kernel)
                    float e2 = 0.f;
                    for (int q = 0; q < in.Kb; ++q) { const float e = in.b[(size_t)r * in.ldb + q]; e2 += e * e; }
                    v = in.b[(size_t)r * in.ldb + j] * (1.0f / fmaxf(sqrtf(e2), 1e-12f));
                    if (blockIdx.x == 0) {
                        float u = 0.f;
                        if (in.noise) {             // the prior sample of this forward, recorded only
                            float n2 = 0.f, mine = 0.f;
                            for (int q = 0; q < in.Kb; ++q) {
                                const float nz = in.eps ? in.eps[(size_t)r * in.Kb + q] : philox_normal(in.seed, in.offset, r, q);
                                n2 += nz * nz;
                                if (q == j) mine = nz;
                            }
                            u = mine * (1.0f / fmaxf(sqrtf(n2), 1e-12f));
                        }
                        if (in.z_out) in.z_out[(size_t)r * in.Kb + j] = v;
                        in.eps_used[(size_t)r * in.Kb + j] = u;
                    }
                }
            }
        }
        xs[i] = v;
    }
    __syncthreads();
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = lane * 4 + 256 * j;
        if (k < K) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const v4f xv = *reinterpret_cast<const v4f*>(xs + r * K + k);
                acc[r] = fmaf(wpre[j].x, xv.x, fmaf(wpre[j].y, xv.y, fmaf(wpre[j].z, xv.z, fmaf(wpre[j].w, xv.w, acc[r]))));
            }
        }
    }
    for (int k = lane * 4 + 1024; k < K; k += 256) {
        const v4f wv = *reinterpret_cast<const v4f*>(wrow + k);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const v4f xv = *reinterpret_cast<const v4f*>(xs + r * K + k);
            acc[r] = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc[r]))));
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && r < rows) {
            v += bias[n];
            v = (relu > 1 && n >= n_valid) ? 0.f : act_apply(v, relu);
            out[(size_t)r * ldo + n] = v;
            if (out2 && n < n2) {
                out2[(size_t)r * ld2 + n] = v;
                if (ls) out2[(size_t)r * ld2 + n2 + n] = ls[n];       // AppendLogStd (rmt:160-206): [a_hat | log_std]
            }
        }
    }
}

// logits[r][n .. 2n) = log_std[0 .. n) for the staged inference path (AppendLogStd, rmt:160-206)
__global__ void __launch_bounds__(256)
append_logstd_kernel(float* __restrict__ logits, int ld, int n, int rows, const float* __restrict__ ls) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rows * n; idx += gridDim.x * 256) {
        const int r = idx / n, c = idx - r * n;
        logits[(size_t)r * ld + n + c] = ls[c];
    }
}

// A stack of dense Linear layers on caller-owned row-major weights W_i[n_out][n_in] (any row stride, any
// alignment), hidden activation act_apply(code), linear output: pvae_mlp_forward.  One wave per output feature
// and chunk of R rows; made for the value branch at rollout batch sizes (rmt:846-853: 2*Db -> 256 -> 256 -> 1).
template <int R>
__global__ void __launch_bounds__(256)
gemv_dense_kernel(const float* __restrict__ x, int ldx, int rows, const float* __restrict__ W, int ldw,
                  const float* __restrict__ bias, int K, int N, int act, float* __restrict__ out, int ldo) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, r0 = blockIdx.y * R;
    if (n >= N) return;
    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)n * ldw + k];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r0 + r < rows) acc[r] = fmaf(w, x[(size_t)(r0 + r) * ldx + k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && r0 + r < rows) out[(size_t)(r0 + r) * ldo + n] = act_apply(v + (bias ? bias[n] : 0.f), act);
    }
}

static XSrc xsrc_of(const pvae_ctx* c, int net, int phase, bool with_s1, int rows);
struct FwdTail {              // what the output layer's epilogue does besides bias
    const EpiMse* mse = nullptr;      // fused MSE loss + gradient
    float* out2 = nullptr;            // or: copy the first n2 output columns to out2[:, off2:]
    int ld2 = 0, off2 = 0, n2 = 0;
    const ProSampler* pro0 = nullptr; // layer 0 forms the sampler's z columns of its input itself (decoder, joint step)
    const XSrc* xs0 = nullptr;        // layer 0 gathers its input rows from the demonstration set (direct steps)
    const ProCols* cols0 = nullptr;   // ... and copies these columns over its input tile (world model: a_t / a_hat)
};

// `row0`: first row of the time-step block to run on (0 unless lookahead > 1)
static int forward_net(pvae_ctx* c, int n, int rows_pad, hipStream_t st, const FwdTail& tail = FwdTail(),
                       int64_t row0 = 0) {
    const NetLayout& N = c->L.net[n];
    const float* x = c->ws + c->W.net[n].in + row0 * N.layers[0].ld;
    int ldx = N.layers[0].ld;
    for (const Layer& l : N.layers) {
        float* out = c->ws + c->W.net[n].act[l.index] + row0 * l.n_out_pad;
        const int rows = (int)c->staged_rows_f;
        // (category 5: the narrow layers that run on 16x16 tiles -- another kernel, gemm_splitk_reg16_kernel)
        const int ps = g_prof.begin(forward_uses_16x16(pad32(rows), l.n_out_pad) && rows > 4 ? 5 : 0,
                                    2.0 * c->staged_rows_f * l.n_in * l.n_out, st);
        if (rows <= 4 && !tail.mse) {            // rollout batch: stream W once over all CUs
            float* o2 = (l.last && tail.out2) ? tail.out2 : nullptr;
            const dim3 grid(l.n_out_pad / 4), block(256);
#define PVAE_GEMV(R)                                                                                        \
    PVAE_LAUNCH((gemv_rows_kernel<R>), grid, block, st, x, ldx, c->params + l.w_off, l.ld,           \
                       c->params + l.b_off, out, l.n_out_pad, l.ld, l.act, o2, tail.ld2, tail.off2, tail.n2, l.n_out)
            if (rows == 1) PVAE_GEMV(1);
            else if (rows == 2) PVAE_GEMV(2);
            else PVAE_GEMV(4);
#undef PVAE_GEMV
            HIP_TRY(hipGetLastError());
        } else if (l.last && tail.mse) {
            EpiMse e = *tail.mse;
            e.out = out; e.ldo = l.n_out_pad; e.bias = c->params + l.b_off;
            HIP_TRY(gemm_forward_epi(x, ldx, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, st));
        } else {
            EpiBiasAct e{out, l.n_out_pad, c->params + l.b_off, l.act};
            e.n_valid = l.n_out;
            if (l.last && tail.out2) { e.out2 = tail.out2; e.ld2 = tail.ld2; e.off2 = tail.off2; e.n2 = tail.n2; }
            if (l.index == 0 && tail.xs0 && tail.pro0)
                HIP_TRY(gemm_forward_pro_gather(*tail.xs0, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, *tail.pro0, st));
            else if (l.index == 0 && tail.xs0 && tail.cols0)
                HIP_TRY(gemm_forward_pro_gather(*tail.xs0, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, *tail.cols0, st));
            else if (l.index == 0 && tail.xs0)
                HIP_TRY(gemm_forward_gather(*tail.xs0, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, st));
            else if (l.index == 0 && tail.pro0)
                HIP_TRY(gemm_forward_pro(x, ldx, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, *tail.pro0, st));
            else
                HIP_TRY(gemm_forward_epi(x, ldx, c->params + l.w_off, l.ld, rows_pad, l.n_out_pad, l.ld, e, st));
        }
        g_prof.end(ps, st);
        x = out;
        ldx = l.n_out_pad;
    }
    return 0;
}

// A backward pass is a list of stages = launches in stream order.  `ready_*` names the slice of
// the gradient arena that is final once the stage has run (data-parallel callers start that
// slice's all-reduce right away, while later stages execute).
struct Stage {
    std::function<int()> run;
    int64_t ready_off = 0, ready_cnt = 0;
    int net = -1;
};
typedef std::vector<Stage> Plan;

// the pending deferred-Adam segment, handed to the launch that is about to go out
// What the launch that is about to go out carries.  A hidden-layer pair absorbs the 28 B/param of a 1024x1024 update
// at ~1 us; a launch with little work of its own (a stack's first / last layer) is as long as the update it
// carries (the sampler-seed pair: 9.6 us, 40 MB).  So such a NARROW launch passes a big pending segment on (it stays
// `held` for the next WIDE launch) and takes only what is small; wide launches and the step's last launch take all.
enum { kTakeAll = 0, kTakeSmall = 1 };
constexpr long long kBigAdamSeg = 150000;                       // float4 elements (a 1024x256 layer: 65.8 K, 1024x1024: 262 K)
static AdamPair take_pending(pvae_ctx* c, int how = kTakeAll) {
    AdamPair p;
    if (how == kTakeSmall && c->pending_adam.n4 >= kBigAdamSeg && c->held_adam.n4 <= 0) {
        c->held_adam = c->pending_adam;                         // pass it on
        c->pending_adam = AdamSeg();
        return p;
    }
    if (how == kTakeSmall && c->held_adam.n4 > 0) {             // still holding one: take the recent one if it is small
        if (c->pending_adam.n4 < kBigAdamSeg) { p.s[0] = c->pending_adam; c->pending_adam = AdamSeg(); }
        else { p.s[0] = c->held_adam; c->held_adam = c->pending_adam; c->pending_adam = AdamSeg(); }   // (two big ones: oldest goes)
        return p;
    }
    p.s[0] = c->held_adam;
    p.s[1] = c->pending_adam;
    if (p.s[0].n4 <= 0) { p.s[0] = p.s[1]; p.s[1] = AdamSeg(); }
    c->held_adam = AdamSeg();
    c->pending_adam = AdamSeg();
    return p;
}
// nothing left to carry them: their own launches
static int flush_pending_adam(pvae_ctx* c, hipStream_t st) {
    const AdamPair p = take_pending(c);
    for (const AdamSeg& a : p.s) {
        if (a.n4 <= 0) continue;
        int grid = (int)((a.n4 + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, st, a.p, a.g, a.m, a.v, a.n4, a.s);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// dz[last] must be filled.  Layer by layer, last to first: the input gradient of layer i reads
// W_i; a weight gradient of layer i with Adam in its epilogue overwrites W_i.  Two schedules:
//  * same layer (the default whenever the update can be deferred, and for the gradient-store path
//    of the data-parallel exchange): wgrad_i only stores its gradient, shares ONE horizontally
//    fused launch with dgrad_i, and Adam_i runs as extra workgroups of the next launch:
//        dgrad_L + wgrad_L | dgrad_{L-1} + wgrad_{L-1} + Adam_L | ... | wgrad_0 + Adam_1
//  * one behind (Adam in the epilogue; no gradient arena, PVAE_SAME_LAYER=0 / PVAE_DEFER_ADAM=0):
//    dgrad_{i-1} (needs dz_{i-1}, W_{i-1}) and wgrad_i (needs dz_i, x_i; writes W_i) are independent:
//        dgrad_L | dgrad_{L-1} + wgrad_L | ... | dgrad_1 + wgrad_2 | [dgrad_0] + wgrad_1 | wgrad_0
//    (without an input gradient the two last weight gradients share a launch).
// `fold` (optional) is executed by the blocks of the last launch.
// What the input-gradient launch of a stack's FIRST layer does with its result (lookahead 1):
// nothing special (store the panel), or form the gradient seed of the stack that produced those
// input columns in its epilogue (pvae_gemm.h: EpiActionSeed / EpiSamplerSeed).
// Columns [c0, c0 + n) of a first-layer input gradient, widened to whole 32-column tiles: the only
// part of that panel a gradient seed reads, so the only part its launch contracts.
struct SeedWindow { int lo, width; };
static inline SeedWindow seed_window(int c0, int n) {
    const int lo = c0 & ~31;
    return SeedWindow{lo, pad32(c0 + n) - lo};
}
struct InputSeed {
    int kind = 0;                      // 0 none, 1 action seed (world model -> decoder), 2 sampler seed (decoder -> encoder)
    EpiActionSeed a;
    EpiSamplerSeed s;
};
// A weight-gradient launch handed from one stack's plan to the next one's first input-gradient
// launch, so the two go out as ONE horizontally fused launch across the stack boundary.
struct DgradArgs {
    const float* dZ; int ldz; const float* W; int ldw; const float* mask; int ldm; float* dX; int ldo;
    int M, Kin, Nd;
    double flops;
    int act = 1;          // act_grad code of the layer behind `mask`
};
struct CarriedWgrad {
    bool valid = false;
    std::function<int(const DgradArgs&)> run_with_dgrad;
    int64_t ready_off = 0, ready_cnt = 0;
    int net = -1;
};

static void plan_backward_net(pvae_ctx* c, int n, int rows_pad, bool train, bool input_grad,
                              const pvae_step_params* sp, bool fused, hipStream_t st, const LossFinal* fold,
                              Plan& plan, const InputSeed* seed = nullptr, CarriedWgrad* carry_out = nullptr,
                              const CarriedWgrad* carry_in = nullptr, bool wide_follows_layer0 = false) {
    const NetLayout* N = &c->L.net[n];
    const NetWork* w = &c->W.net[n];
    const AdamScalars as = adam_scalars(sp, n);
    const int last = (int)N->layers.size() - 1;
    const bool pair = train && c->pair_launch;
    const double rowsf = c->staged_rows_f;
    // act_grad code of the layer whose output masks the input gradient of layer i (layer i - 1; none for i == 0)
    auto mask_act = [=](int i) { return i > 0 ? N->layers[i - 1].act : 1; };
    const int need = n == PVAE_NET_WM ? c->L.cfg.dim_action : c->L.cfg.latent;
    // direct step: the weight gradient of layer 0 contracts over the gathered input (XSrc), not over a staged panel
    const bool dx0 = c->dx.on && train && n != PVAE_NET_PR;
    const XSrc xs0 = dx0 ? xsrc_of(c, n, n == PVAE_NET_WM ? PVAE_PHASE_WORLD : PVAE_PHASE_JOINT, true, (int)c->staged_rows_f) : XSrc();
    LossFinal foldv;
    memset(&foldv, 0, sizeof(foldv));
    if (fold) foldv = *fold;
    InputSeed seedv;
    if (seed) seedv = *seed;

    auto has_dgrad = [=](int i) { return i > 0 || input_grad; };
    auto seg_of = [=](int lo, int hi, Stage& s) {       // layers lo..hi (lo <= hi) of this net
        s.ready_off = N->layers[lo].w_off;
        s.ready_cnt = N->layers[hi].b_off + N->layers[hi].n_out_pad - N->layers[lo].w_off;
        s.net = n;
    };
    // dgrad of layer i: dz[i] (.) W_i -> dz[i-1] (masked) or d_in (i == 0, unmasked)
    auto dgrad = [=](int i) -> int {
        const Layer& l = N->layers[i];
        const float* xin = i == 0 ? c->ws + w->in : c->ws + w->act[i - 1];
        const int ps = g_prof.begin(1, 2.0 * rowsf * (i > 0 ? l.n_in : need) * l.n_out, st);
        if (i == 0 && seedv.kind == 1) {
            // only the input columns the seed consumes are contracted (a 32-aligned window of W_0)
            const SeedWindow sw = seed_window(seedv.a.c0, seedv.a.n);
            EpiActionSeed e = seedv.a;
            e.c0 -= sw.lo;
            HIP_TRY(gemm_dgrad_epi(c->ws + w->dz[0], l.n_out_pad, c->params + l.w_off + sw.lo, l.ld, rows_pad, sw.width,
                                   l.n_out_pad, e, st));
        } else if (i == 0 && seedv.kind == 2) {
            const SeedWindow sw = seed_window(seedv.s.c0, seedv.s.Z);
            EpiSamplerSeed e = seedv.s;
            e.c0 -= sw.lo;
            HIP_TRY(gemm_dgrad_epi(c->ws + w->dz[0], l.n_out_pad, c->params + l.w_off + sw.lo, l.ld, rows_pad, sw.width,
                                   l.n_out_pad, e, st));
        } else {
            HIP_TRY(gemm_dgrad(c->ws + w->dz[i], l.n_out_pad, c->params + l.w_off, l.ld, i > 0 ? xin : nullptr, l.ld,
                               i > 0 ? c->ws + w->dz[i - 1] : c->ws + w->d_in, l.ld, rows_pad, l.ld, l.n_out_pad, st, mask_act(i)));
        }
        g_prof.end(ps, st);
        return 0;
    };
    auto adam_epi = [=](const Layer& l) {
        EpiGradAdam e{c->params + l.w_off, c->m + l.w_off, c->v + l.w_off, l.ld, as};
        e.b = c->params + l.b_off; e.bm = c->m + l.b_off; e.bv = c->v + l.b_off;
        return e;
    };
    auto store_epi = [=](const Layer& l) {
        EpiGradStore e{c->grads + l.w_off, l.ld};
        e.gb = c->grads + l.b_off;
        return e;
    };
    // wgrad of layer i, optionally fused with the dgrad of layer j (j < 0: alone).  In a fused pair
    // that is not the step's last launch the gradient is stored and Adam deferred to workgroups of
    // the next weight-gradient launch (AdamSeg); every launch carries whatever is pending.
    const bool can_defer = fused && c->defer_adam && c->grads != nullptr;
    auto wgrad = [=](int i, int j, bool with_fold) -> int {
        const Layer& l = N->layers[i];
        const float* dz = c->ws + w->dz[i];
        const float* xin = i == 0 ? c->ws + w->in : c->ws + w->act[i - 1];
        // (j == i: the launch also reads W_i, so the update MUST wait for the next one)
#ifdef PVAE_DIAG_EPI_ADAM
        // TIMING-ONLY diagnostic build (docs/experiments.md, round 5): Adam in the epilogue of the same-layer pair, as a
        // second ("ping-pong") parameter arena would allow -- here it overwrites the W_i that the pair's input-gradient half
        // is reading, so the results are wrong; launches, traffic and epilogues are those of the ping-pong schedule.
        const bool defer = can_defer && j >= 0 && j != i && !with_fold;
#else
        const bool defer = can_defer && j >= 0 && (!with_fold || j == i);
#endif
        // (narrow launches -- a stack's last and first layer -- hand a big pending update on to the next hidden-layer
        //  pair of the step, when there is one: take_pending)
        const bool narrow = j == i && !with_fold && ((i == last && last >= 2) || (i == 0 && wide_follows_layer0));
        auto go = [&](auto e) -> int {
            if (with_fold) e.loss = foldv;
            const AdamPair ad = take_pending(c, narrow ? kTakeSmall : kTakeAll);
            if (j >= 0) {
                const Layer& d = N->layers[j];
                const float* dx_in = j == 0 ? c->ws + w->in : c->ws + w->act[j - 1];
                const int pp = g_prof.begin(3, 2.0 * rowsf * ((double)l.n_in * l.n_out +
                                               (double)(j > 0 ? d.n_in : need) * d.n_out), st);
                if (j == 0 && seedv.kind == 2) {
                    const SeedWindow sw = seed_window(seedv.s.c0, seedv.s.Z);
                    EpiSamplerSeed es = seedv.s;
                    es.c0 -= sw.lo;
                    if (dx0)                  // (i == 0 too: the decoder's first layer, X = [s_t | z] gathered)
                        HIP_TRY(gemm_bwd_pair_epi_gather(c->ws + w->dz[0], d.n_out_pad, c->params + d.w_off + sw.lo, d.ld, rows_pad,
                                                         sw.width, d.n_out_pad, es, dz, l.n_out_pad, xs0, l.n_out_pad, l.ld,
                                                         rows_pad, e, st, &ad));
                    else
                    HIP_TRY(gemm_bwd_pair_epi(c->ws + w->dz[0], d.n_out_pad, c->params + d.w_off + sw.lo, d.ld, rows_pad,
                                              sw.width, d.n_out_pad, es, dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld,
                                              rows_pad, e, st, &ad));
                } else {
                    HIP_TRY(gemm_bwd_pair(c->ws + w->dz[j], d.n_out_pad, c->params + d.w_off, d.ld,
                                          j > 0 ? dx_in : nullptr, d.ld, j > 0 ? c->ws + w->dz[j - 1] : c->ws + w->d_in,
                                          d.ld, rows_pad, d.ld, d.n_out_pad, dz, l.n_out_pad, xin, l.ld, l.n_out_pad,
                                          l.ld, rows_pad, e, st, &ad, mask_act(j)));
                }
                g_prof.end(pp, st);
            } else {
                const int pw = g_prof.begin(2, 2.0 * rowsf * l.n_in * l.n_out, st);
                HIP_TRY(gemm_wgrad(dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, rows_pad, e, st, &ad));
                g_prof.end(pw, st);
            }
            return 0;
        };
        if (!fused) return go(store_epi(l));
        if (!defer) return go(adam_epi(l));
        const int rc = go(store_epi(l));
        if (rc == 0) {
            AdamSeg a;
            a.p = c->params + l.w_off; a.g = c->grads + l.w_off; a.m = c->m + l.w_off; a.v = c->v + l.w_off;
            a.n4 = (l.b_off + l.n_out_pad - l.w_off) / 4;
            a.s = as;
            c->pending_adam = a;
        }
        return rc;
    };
    auto wgrad_pair10 = [=](bool with_fold) -> int {    // layers 1 and 0 in one launch
        const Layer& l1 = N->layers[1];
        const Layer& l0 = N->layers[0];
        const int pw2 = g_prof.begin(2, 2.0 * rowsf * ((double)l1.n_in * l1.n_out + (double)l0.n_in * l0.n_out), st);
        auto go = [&](auto e1, auto e0) -> int {
            if (with_fold) e1.loss = foldv;            // block 0 of the launch belongs to the first problem
            // the step's LAST launch also gathers the next minibatch into the alternate panels
            const bool carry = with_fold && c->next_stage.rows_pad > 0;
            const AdamPair ad = take_pending(c);
            HIP_TRY(gemm_wgrad_pair(c->ws + w->dz[1], l1.n_out_pad, c->ws + w->act[0], l1.ld, l1.n_out_pad, l1.ld, e1,
                                    c->ws + w->dz[0], l0.n_out_pad, c->ws + w->in, l0.ld, l0.n_out_pad, l0.ld, e0,
                                    rows_pad, st, carry ? &c->next_stage : nullptr, &ad));
            if (carry) c->next_carried = true;
            return 0;
        };
        const int rc = fused ? go(adam_epi(l1), adam_epi(l0)) : go(store_epi(l1), store_epi(l0));
        g_prof.end(pw2, st);
        return rc;
    };

    // layer 0 alone (same-layer schedule): the step's last launch of a stack without input gradient;
    // carries the loss finalisation, the pending update and the gather of the next minibatch
    auto wgrad_last0 = [=](bool with_fold) -> int {
        const Layer& l0 = N->layers[0];
        const int pw = g_prof.begin(2, 2.0 * rowsf * l0.n_in * l0.n_out, st);
        auto go = [&](auto e0) -> int {
            if (with_fold) e0.loss = foldv;
            const bool carry = with_fold && c->next_stage.rows_pad > 0;
            const AdamPair ad = take_pending(c);
            if (dx0) {                        // X gathered from the demonstration set: nothing was staged, nothing to stage
                HIP_TRY(gemm_wgrad_pair_gather(c->ws + w->dz[0], l0.n_out_pad, xs0, l0.n_out_pad, l0.ld, e0, rows_pad, st, &ad,
                                               with_fold && c->next_touch.blocks > 0 ? &c->next_touch : nullptr));
                return 0;
            }
            HIP_TRY(gemm_wgrad_pair(c->ws + w->dz[0], l0.n_out_pad, c->ws + w->in, l0.ld, l0.n_out_pad, l0.ld, e0,
                                    (const float*)nullptr, 0, (const float*)nullptr, 0, 0, l0.ld, e0,
                                    rows_pad, st, carry ? &c->next_stage : nullptr, &ad));
            if (carry) c->next_carried = true;
            return 0;
        };
        const int rc = fused ? go(adam_epi(l0)) : go(store_epi(l0));
        g_prof.end(pw, st);
        return rc;
    };

    auto push = [&](std::function<int()> f) -> Stage& {
        plan.emplace_back();
        plan.back().run = std::move(f);
        return plan.back();
    };
    if (!train) {
        for (int i = last; i >= 0; --i)
            if (has_dgrad(i)) push([=] { return dgrad(i); });
        return;
    }
    if (!pair) {
        for (int i = last; i >= 0; --i) {
            if (has_dgrad(i)) push([=] { return dgrad(i); });
            const bool f = fold && i == 0;
            seg_of(i, i, push([=] { return wgrad(i, -1, f); }));
        }
        return;
    }
    if ((!fused || can_defer) && c->same_layer_pairs) {
        // Same-layer schedule: with the update deferred (or no update at all: gradient store for the
        // data-parallel exchange) wgrad_i no longer writes W_i, so it shares a launch with dgrad_i
        // instead of trailing one launch behind it:
        //     dgrad_L + wgrad_L | dgrad_{L-1} + wgrad_{L-1} [+ Adam_L] | ... | wgrad_0 [+ Adam_1]
        // The short first launch (K = output width) and the short last one (narrow layer 0) each get
        // a partner of their own size, instead of a lone short launch at one end and two narrow
        // problems in one launch at the other.
        for (int i = last; i >= 0; --i) {
            const bool f = fold && i == 0;
            if (has_dgrad(i)) seg_of(i, i, push([=] { return wgrad(i, i, f); }));
            else seg_of(0, 0, push([=] { return wgrad_last0(f); }));
        }
        return;
    }
    if (has_dgrad(last)) {
        if (carry_in && carry_in->valid) {
            // the previous stack's trailing weight gradient rides with this stack's first input gradient
            const CarriedWgrad cw = *carry_in;
            const Layer& l = N->layers[last];
            DgradArgs da{c->ws + w->dz[last], l.n_out_pad, c->params + l.w_off, l.ld,
                         last > 0 ? c->ws + w->act[last - 1] : nullptr, l.ld,
                         last > 0 ? c->ws + w->dz[last - 1] : c->ws + w->d_in, l.ld, rows_pad, l.ld, l.n_out_pad,
                         2.0 * rowsf * l.n_in * l.n_out, mask_act(last)};
            Stage& sref = push([=] { return cw.run_with_dgrad(da); });
            sref.ready_off = cw.ready_off; sref.ready_cnt = cw.ready_cnt; sref.net = cw.net;
        } else {
            push([=] { return dgrad(last); });
        }
    }
    for (int i = last; i >= 0; --i) {
        const int j = i - 1;                       // dgrad_{i-1} rides with wgrad_i
        if (j >= 0 && has_dgrad(j)) {
            const bool f = fold && i == 0;
            seg_of(i, i, push([=] { return wgrad(i, j, f); }));
        } else if (i == 1 && !has_dgrad(0)) {
            const bool f = fold != nullptr;
            seg_of(0, 1, push([=] { return wgrad_pair10(f); }));
            return;
        } else if (i == 0 && carry_out && !fold) {
            // hand the lone trailing weight gradient to the next stack's plan
            const Layer& l = N->layers[0];
            const float* dz = c->ws + w->dz[0];
            const float* xin = c->ws + w->in;
            carry_out->valid = true;
            carry_out->ready_off = l.w_off;
            carry_out->ready_cnt = l.b_off + l.n_out_pad - l.w_off;
            carry_out->net = n;
            carry_out->run_with_dgrad = [=](const DgradArgs& d) -> int {
                const int pp = g_prof.begin(3, d.flops + 2.0 * rowsf * l.n_in * l.n_out, st);
                hipError_t he;
                const AdamPair ad = take_pending(c);
                if (fused) {
                    he = gemm_bwd_pair(d.dZ, d.ldz, d.W, d.ldw, d.mask, d.ldm, d.dX, d.ldo, d.M, d.Kin, d.Nd, dz,
                                       l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, rows_pad, adam_epi(l), st, &ad, d.act);
                } else {
                    he = gemm_bwd_pair(d.dZ, d.ldz, d.W, d.ldw, d.mask, d.ldm, d.dX, d.ldo, d.M, d.Kin, d.Nd, dz,
                                       l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, rows_pad, store_epi(l), st, &ad, d.act);
                }
                g_prof.end(pp, st);
                if (he != hipSuccess) return fail(-10, "gemm_bwd_pair: %s", hipGetErrorString(he));
                return 0;
            };
        } else {
            const bool f = fold && i == 0;
            seg_of(i, i, push([=] { return wgrad(i, -1, f); }));
        }
    }
}

// ---------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------
extern "C" {

int pvae_abi_version(void) { return PVAE_ABI_VERSION; }
const char* pvae_last_error(void) { return g_err; }

int pvae_num_layers(const pvae_config* cfg) {
    if (!cfg) return fail(-1, "null cfg");
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    int n = 0;
    for (auto& N : L.net) n += (int)N.layers.size();
    return n;
}

int pvae_layer(const pvae_config* cfg, int i, pvae_layer_info* out) {
    if (!cfg || !out) return fail(-1, "null argument");
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    for (int n : kArenaOrder) {
        const NetLayout& N = L.net[n];
        if (i < (int)N.layers.size()) {
            const Layer& l = N.layers[i];
            out->net = l.net; out->index = l.index; out->n_in = l.n_in; out->n_out = l.n_out; out->col0 = l.col0;
            out->ld = l.ld; out->n_out_pad = l.n_out_pad; out->w_offset = l.w_off; out->b_offset = l.b_off;
            out->act = l.act == 0 ? PVAE_ACT_LINEAR : l.act - 1;
            return 0;
        }
        i -= (int)N.layers.size();
    }
    return fail(-1, "layer index out of range");
}

int64_t pvae_arena_floats(const pvae_config* cfg) {
    if (!cfg) return fail(-1, "null cfg");
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    return L.arena_floats;
}

int pvae_net_segment(const pvae_config* cfg, int net, int64_t* offset, int64_t* count) {
    if (!cfg || !offset || !count) return fail(-1, "null argument");
    if (net < 0 || net >= PVAE_NUM_NETS) return fail(-1, "bad net id %d", net);
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    *offset = L.net[net].off;
    *count = L.net[net].count;
    return 0;
}

size_t pvae_workspace_bytes(const pvae_config* cfg) {
    if (!cfg) return 0;
    Layout L = make_layout(*cfg);
    if (!L.ok) return 0;
    return (size_t)make_workspace(L).total_floats * sizeof(float);
}

int64_t pvae_workspace_offset(const pvae_config* cfg, int kind, int net, int layer) {
    if (!cfg) return fail(-1, "null cfg");
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    Workspace W = make_workspace(L);
    if (kind >= 0 && kind <= 3) {
        if (net < 0 || net >= PVAE_NUM_NETS || L.net[net].layers.empty()) return fail(-1, "bad net id %d", net);
        if (kind >= 2 && (layer < 0 || layer >= (int)L.net[net].layers.size())) return fail(-1, "bad layer %d", layer);
    }
    switch (kind) {
        case 0: return W.net[net].in;
        case 1: return W.net[net].d_in;
        case 2: return W.net[net].act[layer];
        case 3: return W.net[net].dz[layer];
        case 4: return W.s2;
        case 5: return W.act_t;
        case 6: return W.eps;
        case 7: return W.obs_keep;
        default: return fail(-1, "bad kind %d", kind);
    }
}

int pvae_create(const pvae_config* cfg, pvae_ctx** out) {
    if (!cfg || !out) return fail(-1, "null argument");
    Layout L = make_layout(*cfg);
    if (!L.ok) return fail(-1, "bad config: %s", L.why);
    pvae_ctx* c = new (std::nothrow) pvae_ctx();
    if (!c) return fail(-3, "out of host memory");
    c->L = L;
    c->W = make_workspace(L);
    memset(&c->next_stage, 0, sizeof(c->next_stage));
    *out = c;
    return 0;
}

// Switches of schedule and tile geometry (what used to be PVAE_* environment variables read inside the library): explicit,
// through the ABI.  ctx == NULL: process-wide kernel-geometry switches; else that context's schedule.  The production
// values are the defaults; the parity tests flip them to hold every variant to the same bits.
static bool g_look_pair = true, g_rollout_fused = true;
int pvae_set_option(pvae_ctx* c, const char* name, int64_t value) {
    if (!name) return fail(-1, "null option name");
    const std::string k(name);
    const int v = (int)value;
    if (!c) {
        if (k == "krot") g_krot = v;
        else if (k == "rowxcd") g_rowxcd = v;
        else if (k == "ws64") g_ws64 = v;
        else if (k == "ws6464") g_ws6464 = v;
        else if (k == "ws6464_rows") g_ws6464_rows = v;
        else if (k == "pair64") g_pair64 = v;
        else if (k == "dgrad16") g_dgrad16 = v;
        else if (k == "wgrad32") g_wgrad32 = v;
        else if (k == "look_pair") g_look_pair = v != 0;
        else if (k == "rollout_fused") g_rollout_fused = v != 0;
        else return fail(-1, "unknown process-wide option '%s'", name);
        return 0;
    }
    if (k == "pair") c->pair_launch = v != 0;
    else if (k == "defer_adam") c->defer_adam = v != 0;
    else if (k == "same_layer") c->same_layer_pairs = v != 0;
    else if (k == "fold_sampler") c->fold_sampler = v != 0;
    else if (k == "direct") return pvae_set_direct(c, v);
    else if (k == "p2p_timeout_ms") { if (value > 0) c->p2p.timeout_ticks = (long long)value * 100000ll; }
    else if (k == "p2p_selftest_flags_only") c->p2p_selftest_flags_only = v != 0;
    else if (k == "server_mailbox") c->server_mailbox = v;             // 0 auto (device memory with a large BAR), 1 host, 2 device
    else return fail(-1, "unknown context option '%s'", name);
    return 0;
}

int pvae_p2p_close(pvae_ctx* c);
void pvae_destroy(pvae_ctx* ctx) {
    if (ctx && ctx->comm && g_rccl.ok()) g_rccl.CommDestroy(ctx->comm);
    if (ctx) {
        pvae_p2p_close(ctx);
        if (ctx->p2p.flags) (void)hipFree(ctx->p2p.flags);
        if (ctx->p2p.staging) (void)hipFree(ctx->p2p.staging);
        if (ctx->p2p.self_buf) (void)hipFree(ctx->p2p.self_buf);
        server_free(ctx);
    }
    delete ctx;
}

int pvae_bind_arenas(pvae_ctx* c, float* params, float* grads, float* exp_avg, float* exp_avg_sq) {
    if (!c) return fail(-1, "null ctx");
    if (!params) return fail(-1, "params arena is null");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15)
        return fail(-1, "arenas must be 16-byte aligned");
    c->params = params; c->grads = grads; c->m = exp_avg; c->v = exp_avg_sq;
    params_touched(c, nullptr, false);
    return 0;
}

int pvae_bind_workspace(pvae_ctx* c, void* workspace, size_t bytes) {
    if (!c) return fail(-1, "null ctx");
    if (!workspace || ((uintptr_t)workspace & 255)) return fail(-1, "workspace must be 256-byte aligned");
    if (bytes < (size_t)c->W.total_floats * sizeof(float))
        return fail(-1, "workspace too small: %zu < %zu", bytes, (size_t)c->W.total_floats * sizeof(float));
    c->ws = (float*)workspace;
    c->seed_pads_clean = false;
    c->pf.valid = false;
    return 0;
}

int pvae_bind_dataset(pvae_ctx* c, const float* states, const float* actions, const int32_t* window_row,
                      int64_t n_rows, int64_t n_windows) {
    if (!c) return fail(-1, "null ctx");
    if (!states || !actions || !window_row) return fail(-1, "null dataset pointer");
    if (n_rows < 2 || n_windows < 1) return fail(-1, "empty dataset");
    if (n_rows > 2147483647ll) return fail(-1, "more than 2^31-1 rows");
    c->states = states; c->actions = actions; c->window_row = window_row;
    c->next_states = nullptr;
    c->n_rows = n_rows; c->n_windows = n_windows;
    c->pf.valid = false;         // a minibatch gathered ahead came from the previous binding
    // the gathered first layers fetch whole 16-byte chunks: the last one of a row may reach 12 bytes past it, i.e. past the
    // array for its very last row.  Only allocations with that much room behind them qualify (else: the panel path).
    auto roomy = [](const float* p, int64_t floats) {
        void* base = nullptr; size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)p) != hipSuccess) { (void)hipGetLastError(); return false; }
        return (const char*)(p + floats) + 16 <= (const char*)base + size;
    };
    c->data_slack = roomy(states, n_rows * c->L.cfg.dim_body) && roomy(actions, n_rows * c->L.cfg.dim_action);
    // window -> row on the host (RowMap: a minibatch's rows as two runs in kernel arguments instead of an index load in
    // front of every first-layer launch).  The caller's array must be final when it is bound.
    c->window_row_host.clear();
    if (c->data_slack && c->direct) {                   // (pvae_set_direct after the bind: index loads instead -- still correct)
        c->window_row_host.resize((size_t)n_windows);
        if (hipMemcpy(c->window_row_host.data(), window_row, (size_t)n_windows * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            c->window_row_host.clear();
        }
    }
    return 0;
}

int pvae_set_direct(pvae_ctx* c, int on) {
    if (!c) return fail(-1, "null ctx");
    c->direct = on != 0;
    if (c->direct && c->data_slack && c->window_row && (int64_t)c->window_row_host.size() != c->n_windows) {
        c->window_row_host.resize((size_t)c->n_windows);
        if (hipMemcpy(c->window_row_host.data(), c->window_row, (size_t)c->n_windows * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess) {
            (void)hipGetLastError();
            c->window_row_host.clear();
        }
    }
    return 0;
}
// 1: the next training step on this binding would read the demonstration set directly (same arguments as the step)
int pvae_direct_active(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, int fused);

int pvae_bind_dataset_next(pvae_ctx* c, const float* next_states) {
    if (!c) return fail(-1, "null ctx");
    if (!c->states) return fail(-2, "dataset not bound");
    c->next_states = next_states;
    c->pf.valid = false;
    return 0;
}

// `steps`: time steps to stage (the ctx's lookahead for training batches, 1 for rollout inference)
// Arguments of a staging job into the CURRENT (alt == false) or the alternate set of input panels.
static StageArgs stage_args(const pvae_ctx* c, long long first_window, const float* x, const float* y, int rows,
                            bool from_set, int steps, bool alt) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action;
    float* w = c->ws;
    const int ld_wm = c->L.net[PVAE_NET_WM].layers[0].ld;
    StageArgs a;
    memset(&a, 0, sizeof(a));
    a.states = from_set ? c->states : nullptr;
    a.next_states = from_set ? c->next_states : nullptr;
    a.actions = from_set ? c->actions : nullptr;
    a.window_row = from_set ? c->window_row : nullptr;
    a.first_window = first_window;
    a.x = x; a.y = y;
    a.rows = rows; a.rows_pad = pad32(rows); a.Db = Db; a.Da = Da; a.L = steps;
    a.te_in = w + (alt ? c->W.alt_in[PVAE_NET_TE] : c->W.net[PVAE_NET_TE].in); a.ld_te = c->L.net[PVAE_NET_TE].layers[0].ld;
    a.md_in = w + (alt ? c->W.alt_in[PVAE_NET_MD] : c->W.net[PVAE_NET_MD].in); a.ld_md = c->L.net[PVAE_NET_MD].layers[0].ld;
    a.wm_in = w + (alt ? c->W.alt_in[PVAE_NET_WM] : c->W.net[PVAE_NET_WM].in); a.ld_wm = ld_wm;
    a.s2 = w + (alt ? c->W.alt_s2 : c->W.s2); a.ld_s2 = pad64(Db);
    a.act_t = w + (alt ? c->W.alt_act_t : c->W.act_t); a.ld_a = pad64(Da);
    a.wm_pred = (steps > 1 && !alt) ? a.wm_in + (int64_t)steps * a.rows_pad * ld_wm : nullptr;
    if (!c->L.net[PVAE_NET_PR].layers.empty()) {
        a.pr_in = w + (alt ? c->W.alt_in[PVAE_NET_PR] : c->W.net[PVAE_NET_PR].in);
        a.ld_pr = c->L.net[PVAE_NET_PR].layers[0].ld;
    }
    const int te = c->L.cfg.te_inputs, md = c->L.cfg.md_inputs;          // input subsets: the blocks left out are staged as zeros
    a.in_off = (te == PVAE_INPUT_TASK ? 1 : 0) | (te == PVAE_INPUT_BODY ? 2 : 0) | (md == PVAE_INPUT_TASK ? 4 : 0);
    return a;
}

// `steps`: time steps to stage (the ctx's lookahead for training batches, 1 for rollout inference)
static int stage(pvae_ctx* c, long long first_window, const float* x, const float* y, int rows, bool from_set,
                 hipStream_t st, int steps) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (rows < 1 || rows > c->L.cfg.max_batch) return fail(-1, "rows %d outside [1, %d]", rows, c->L.cfg.max_batch);
    c->dx.on = false;
    const StageArgs a = stage_args(c, first_window, x, y, rows, from_set, steps, false);
    const int lf = stage_lds_floats(a.Db, a.Da);
    hipLaunchKernelGGL(stage_batch_kernel, dim3((a.rows_pad + 3) / 4, steps), dim3(256), (size_t)4 * lf * sizeof(float), st, a, lf);
    HIP_TRY(hipGetLastError());
    c->staged_rows = rows;
    c->staged_rows_f = rows;
    return 0;
}

int pvae_invalidate_staging(pvae_ctx* c) {
    if (!c) return fail(-1, "null ctx");
    c->pf.valid = false;
    c->staged_rows = 0;
    return 0;
}

int pvae_gather(pvae_ctx* c, int64_t first_window, int32_t rows, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (!c->states) return fail(-2, "dataset not bound");
    if (first_window < 0 || first_window + rows > c->n_windows)
        return fail(-1, "windows [%lld, %lld) outside [0, %lld)", (long long)first_window,
                    (long long)(first_window + rows), (long long)c->n_windows);
    return stage(c, first_window, nullptr, nullptr, rows, true, (hipStream_t)stream, c->W.L);
}

int pvae_set_batch(pvae_ctx* c, const float* x, const float* y, int32_t rows, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (!x) return fail(-1, "x is null");
    return stage(c, 0, x, y, rows, false, (hipStream_t)stream, c->W.L);
}

}  // extern "C"

// The sampler of the configured prior kind (rmt:795-819): reparam_kernel (N(mu, s^2); KL to N(0, I) or to
// the learned prior mean mu_p) or sphere_kernel (unit-sphere encoder).  `partial` may be null (rollout).
static int sampler_grid(const pvae_ctx* c, int rows_pad) {
    if (c->L.cfg.prior_kind >= PVAE_PRIOR_HYPERSPHERE) return rows_pad / 4;
    const int Z = c->L.cfg.latent;
    return (rows_pad * Z + 255) / 256 < 64 ? (rows_pad * Z + 255) / 256 : 64;
}
// Where the sampler's z goes: columns [Db, Db + Z) of the decoder's input panel -- or, for a decoder that reads s_t only
// (motor_decoder_inputs = ["body"], rmt:822-829), of a side panel of the same shape: the code is still drawn, kept for
// pvae_read_tensor and priced by the KL term, but must not sit in the operand of the decoder's weight gradient (the
// weights of those columns are structural zeros and stay so because the operand is zero there).
static int64_t z_panel(const pvae_ctx* c) {
    return c->L.cfg.md_inputs == PVAE_INPUT_BODY ? c->W.z_side : c->W.net[PVAE_NET_MD].in;
}
static int launch_sampler(pvae_ctx* c, const float* te_out, int ldte, const float* eps, float* eps_used, float* md_in,
                          int ld_md, int rows, int rows_pad, int noise, unsigned long long seed,
                          unsigned long long offset, float* partial, float* z_dense, const float* mu_p, int ldmp,
                          hipStream_t st) {
    const int Db = c->L.cfg.dim_body, Z = c->L.cfg.latent;
    md_in += z_panel(c) - c->W.net[PVAE_NET_MD].in;
    if (c->L.cfg.prior_kind >= PVAE_PRIOR_HYPERSPHERE) {
        hipLaunchKernelGGL(sphere_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, st, te_out, ldte, eps, eps_used, md_in,
                           ld_md, Db, Z, rows, rows_pad, noise, seed, offset, partial, z_dense,
                           c->L.cfg.prior_kind == PVAE_PRIOR_HYPERSPHERE ? 1 : 0);
    } else {
        hipLaunchKernelGGL(reparam_kernel, dim3(sampler_grid(c, rows_pad)), dim3(256), 0, st, te_out, ldte, eps, eps_used,
                           md_in, ld_md, Db, Z, rows, rows_pad, noise, seed, offset, partial, z_dense, mu_p, ldmp);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// Everything a step needs that is a pure function of (phase, rows, step params).
// The sampler runs as the prologue of the decoder's first-layer launch (ProSampler) when that launch is the 32x32-tile
// kernel and the prior is the reference's default: joint training steps at lookahead 1, more than 4 rows.
static bool sampler_folds(const pvae_ctx* c, int rows) {
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    return c->fold_sampler && c->pair_launch && c->W.L == 1 && c->L.cfg.prior_kind == PVAE_PRIOR_ZERO_MEAN &&
           c->L.cfg.md_inputs != PVAE_INPUT_BODY &&
           c->L.net[PVAE_NET_PR].layers.empty() && c->L.cfg.latent <= ProSampler::kMaxZ && c->L.cfg.latent % 4 == 0 &&
           rows > 4 &&
           MD.layers.size() > 1 && forward_pro_ok(pad32(rows), MD.layers[0].n_out_pad);
}

// ---- first layers on the demonstration set (XSrc) ------------------------------------------------------------
// The gathered input of stack `net` in the step in flight.  `with_s1`: the second column block is part of the operand
// (weight gradients; forward layers on 64-row tiles) -- false when a Pro patch of the launch supplies those columns.
static XSrc xsrc_of(const pvae_ctx* c, int net, int phase, bool with_s1, int rows) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    XSrc x;
    memset(&x, 0, sizeof(x));
    x.s0 = c->states; x.rm = c->dx.rm; x.ld0 = Db; x.rows = rows;
    x.zero = c->ws + c->W.zero;
    x.s1 = x.zero;
    if (net == PVAE_NET_TE) { x.n0 = 2 * Db; return x; }               // [s_t | s_{t+1}]: one run of 2 Db floats of `states`
    x.n0 = Db;
    if (!with_s1) return x;
    if (net == PVAE_NET_MD) {                                          // [s_t | z]: z where the sampler stored it
        x.s1 = c->ws + c->W.net[PVAE_NET_MD].in + Db; x.ind1 = 0; x.ld1 = c->L.net[PVAE_NET_MD].layers[0].ld; x.n1 = Z;
    } else if (phase == PVAE_PHASE_WORLD) {                            // [s_t | a_t]
        x.s1 = c->actions; x.ind1 = 1; x.ld1 = Da; x.n1 = Da;
    } else {                                                           // [s_t | a_hat]: the decoder's output panel
        x.s1 = c->ws + c->W.net[PVAE_NET_MD].act.back(); x.ind1 = 0; x.ld1 = c->L.net[PVAE_NET_MD].layers.back().n_out_pad; x.n1 = Da;
    }
    return x;
}
static bool sampler_folds(const pvae_ctx* c, int rows);
// Can this training step read the demonstration set directly?  (Everything else keeps the staging launch.)
static bool direct_ok(const pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, bool fused) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    if (!c->direct || !c->data_slack || !c->states || c->next_states || c->W.L != 1 || !c->pair_launch || !c->same_layer_pairs)
        return false;
    if (rows <= 4 || c->L.cfg.prior_kind != PVAE_PRIOR_ZERO_MEAN || !c->L.net[PVAE_NET_PR].layers.empty()) return false;
    // input subsets: the staged panels carry the zeros.  Each field on its own -- BODY (1) on one stack and TASK (2) on
    // the other OR to 3, which is also what "both" is spelt as
    if (c->L.cfg.te_inputs % 3 != 0 || c->L.cfg.md_inputs % 3 != 0) return false;
    if (!c->L.net[PVAE_NET_MH].layers.empty()) return false;                     // the helper reads the staged decoder panel
    if (fused && !(c->defer_adam && c->grads)) return false;          // (the same-layer schedule of plan_backward_net)
    if (Da > ProCols::kMaxN || Z > ProCols::kMaxN || Db < 64 || 2 * Db >= 65536) return false;
    const int rp = pad32(rows);
    // a first layer on 64-row tiles has no Pro patch: its second column block is chunk-selected, which needs dim_body % 4 == 0
    auto layer0_ok = [&](int net, bool second_block) {
        const NetLayout& N = c->L.net[net];
        if (N.layers.size() < 2) return false;
        const int n = N.layers[0].n_out_pad;
        if (!forward_gather_ok(rp, n)) return false;
        return !(second_block && uses_64x32(rp, n) && (Db & 3));
    };
    if (phase == PVAE_PHASE_WORLD) return layer0_ok(PVAE_NET_WM, true);
    if (!(sp->cycle_coeff > 0.0f)) return false;                      // (the action loss sits in the world model's seed epilogue)
    if (!layer0_ok(PVAE_NET_TE, false) || !layer0_ok(PVAE_NET_MD, true) || !layer0_ok(PVAE_NET_WM, true)) return false;
    // decoder on 32x32 tiles: z comes from the sampler prologue of that very launch
    if (!uses_64x32(rp, c->L.net[PVAE_NET_MD].layers[0].n_out_pad) && !sampler_folds(c, rows)) return false;
    return true;
}

struct StepShape {
    bool fold_sampler;
    int rows_pad, wm_tiles, gridz, nparts_a;
    bool seed_action, seed_sampler;   // stack hand-overs fused into input-gradient epilogues (plan_backward)
    int l1;                    // loss_kind of the three reconstruction terms
    float gs;                  // d(mean loss)/d(residual) factor: 2 for MSE, 1 for L1
    float Bg;
    bool cyc_grad, kl_active;
    LossFinal lf;
};

static int step_shape(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, float* loss_out, bool backward,
                      StepShape& S) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    S.rows_pad = pad32(rows);
    S.l1 = sp->loss_kind == PVAE_LOSS_L1 ? 1 : 0;
    S.gs = S.l1 ? 1.0f : 2.0f;
    S.Bg = (float)(sp->global_rows > 0 ? sp->global_rows : rows);
    const int T = c->W.L;
    S.wm_tiles = forward_tiles(S.rows_pad, c->L.net[PVAE_NET_WM].layers.back().n_out_pad);
    if ((int64_t)S.wm_tiles * T > kLossParts)
        return fail(-1, "batch x dim_body x lookahead too large for the loss partial buffer");
    S.Bg *= (float)T;                          // every term is the mean over the L steps (tpv:423-428)
    S.gridz = sampler_grid(c, S.rows_pad);
    S.fold_sampler = phase == PVAE_PHASE_JOINT && sampler_folds(c, rows);
    if (S.fold_sampler) S.gridz = S.rows_pad / 32;          // one KL partial per row block
    (void)Z;
    S.nparts_a = S.rows_pad < 64 ? S.rows_pad : 64;
    S.cyc_grad = backward && phase == PVAE_PHASE_JOINT && sp->cycle_coeff > 0.0f;
    S.kl_active = phase == PVAE_PHASE_JOINT && sp->kl_coeff > 0.0f && sp->a_rec_coeff > 0.0f &&   // tpv:381-384
                  c->L.cfg.prior_kind != PVAE_PRIOR_NONE;                 // (`if self.latent_prior_type and ...`)
    // (the sphere's backward needs a dot product over a whole latent row, which no tile epilogue sees)
    // (a helper stack sits between the two hand-overs -- its seed reads the decoder's, its input gradient joins the
    //  decoder's before the sampler backward -- so a helper model takes the stand-alone glue kernels)
    S.seed_sampler = backward && phase == PVAE_PHASE_JOINT && c->W.L == 1 && c->pair_launch &&
                     c->L.cfg.prior_kind < PVAE_PRIOR_HYPERSPHERE && c->L.net[PVAE_NET_MH].layers.empty();
    S.seed_action = S.seed_sampler && S.cyc_grad;
    float* part = c->ws + c->W.loss_part;
    memset(&S.lf, 0, sizeof(S.lf));
    for (int t = 0; t < 4; ++t) S.lf.part[t] = part + (t + 1) * kLossParts;
    S.lf.out = loss_out;
    S.lf.scale[0] = 1.0f / (S.Bg * Da); S.lf.scale[1] = 1.0f / S.Bg;
    S.lf.scale[2] = 1.0f / (S.Bg * Db); S.lf.scale[3] = 1.0f / (S.Bg * Db);
    S.lf.coeff[0] = sp->a_rec_coeff; S.lf.coeff[1] = sp->kl_coeff;
    S.lf.coeff[2] = sp->s_rec_coeff; S.lf.coeff[3] = sp->cycle_coeff;
    if (phase == PVAE_PHASE_WORLD) {
        S.lf.nparts[2] = S.wm_tiles * T;
    } else {
        if (sp->a_rec_coeff > 0.0f)
            S.lf.nparts[0] = S.seed_action ? dgrad_tiles(S.rows_pad, seed_window(Db, Da).width) : S.nparts_a * T;
        if (S.kl_active) S.lf.nparts[1] = S.gridz * T;
        if (sp->cycle_coeff > 0.0f) S.lf.nparts[3] = S.wm_tiles * T;
    }
    return 0;
}

static int check_step(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, bool backward, bool fused) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!sp) return fail(-1, "null step params");
    if (phase != PVAE_PHASE_WORLD && phase != PVAE_PHASE_JOINT) return fail(-1, "unknown phase %d", phase);
    if (sp->loss_kind != PVAE_LOSS_MSE && sp->loss_kind != PVAE_LOSS_L1) return fail(-1, "unknown loss_kind %d", sp->loss_kind);
    if (rows < 1 || rows > c->L.cfg.max_batch) return fail(-1, "rows %d outside [1, %d]", rows, c->L.cfg.max_batch);
    if (rows != c->staged_rows) return fail(-2, "rows %d != staged rows %d", rows, c->staged_rows);
    if (backward && fused && (!c->m || !c->v)) return fail(-2, "Adam moment arenas not bound");
    if (backward && !fused && !c->grads) return fail(-2, "gradient arena not bound");
    if (phase == PVAE_PHASE_JOINT && sp->s_rec_coeff != 0.0f)
        return fail(-4, "joint phase with world_model_s_rec_coeff != 0 is not supported "
                        "(reference default is 0.0, tpv:284)");
    return 0;
}

static int run_forward_unrolled(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, const float* eps,
                                bool backward, const StepShape& S, hipStream_t st);
static void plan_backward_unrolled(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, bool backward,
                                   bool fused, const StepShape& S, hipStream_t st, Plan& plan);

// Forward launches + loss partials + the gradient seed of the world model's output layer.
static int run_forward(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, const float* eps, bool backward,
                       const StepShape& S, hipStream_t st) {
    if (c->W.L > 1) return run_forward_unrolled(c, phase, rows, sp, eps, backward, S, st);
    int rc;
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    float* w = c->ws;
    float* part = w + c->W.loss_part;
    const NetLayout& TE = c->L.net[PVAE_NET_TE];
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    const NetLayout& WM = c->L.net[PVAE_NET_WM];
    const NetWork& wte = c->W.net[PVAE_NET_TE];
    const NetWork& wmd = c->W.net[PVAE_NET_MD];
    const NetWork& wwm = c->W.net[PVAE_NET_WM];
    // world-model output layer fused with MSE(s2, .) and its gradient
    EpiMse mse;
    memset(&mse, 0, sizeof(mse));
    mse.target = w + c->W.s2; mse.ldt = pad64(Db);
    mse.dz = backward ? w + wwm.dz.back() : nullptr; mse.ldz = WM.layers.back().n_out_pad;
    mse.rows = rows; mse.D = Db; mse.l1 = S.l1;
    FwdTail wm_tail;
    wm_tail.mse = &mse;
    // direct step: every stack's first layer gathers its rows itself, s_{t+1} is read from `states` by the loss epilogue
    const bool dx = c->dx.on;
    XSrc xs_te, xs_md, xs_wm;
    ProCols wm_cols;
    memset(&wm_cols, 0, sizeof(wm_cols));
    if (dx) {
        mse.target = c->states + Db; mse.ldt = Db; mse.tind = 1; mse.trm = c->dx.rm;   // row + 1 of the window's s_t
        const bool wm64 = uses_64x32(S.rows_pad, WM.layers[0].n_out_pad);
        xs_wm = xsrc_of(c, PVAE_NET_WM, phase, wm64, rows);
        wm_tail.xs0 = &xs_wm;
        if (!wm64) {
            const XSrc full = xsrc_of(c, PVAE_NET_WM, phase, true, rows);
            wm_cols.src = full.s1; wm_cols.ind = full.ind1; wm_cols.rm = full.rm; wm_cols.ld = full.ld1; wm_cols.c0 = Db; wm_cols.n = Da;
            wm_cols.rows = rows;
            wm_tail.cols0 = &wm_cols;
        }
    }
    if (phase == PVAE_PHASE_WORLD) {
        // tpv:411-414: L = s_rec * MSE(s2, WM(s1, a_gt)); only the world model learns (tpv:326-329)
        mse.grad_scale = sp->s_rec_coeff * S.gs / (S.Bg * Db);
        mse.partial = part + 3 * kLossParts;
        return forward_net(c, PVAE_NET_WM, S.rows_pad, st, wm_tail);
    }
    const NetLayout& PR = c->L.net[PVAE_NET_PR];
    const NetWork& wpr = c->W.net[PVAE_NET_PR];
    const bool learned_prior = !PR.layers.empty();
    if (!c->seed_pads_clean) {
        // the seed epilogues (plan_backward) write only the real columns of these gradient
        // panels; their pad columns must be zero and nothing else ever writes them
        HIP_TRY(hipMemsetAsync(w + wmd.dz.back(), 0, (size_t)c->W.Bp * MD.layers.back().n_out_pad * sizeof(float), st));
        HIP_TRY(hipMemsetAsync(w + wte.dz.back(), 0, (size_t)c->W.Bp * TE.layers.back().n_out_pad * sizeof(float), st));
        if (learned_prior)
            HIP_TRY(hipMemsetAsync(w + wpr.dz.back(), 0, (size_t)c->W.Bp * PR.layers.back().n_out_pad * sizeof(float), st));
        c->seed_pads_clean = true;
    }
    // joint forward: [prior mean ->] TE -> sampler -> MD -> WM (rmt:742-771, 801-809)
    if (learned_prior && (rc = forward_net(c, PVAE_NET_PR, S.rows_pad, st))) return rc;
    FwdTail te_tail;
    if (dx) { xs_te = xsrc_of(c, PVAE_NET_TE, phase, false, rows); te_tail.xs0 = &xs_te; }
    if ((rc = forward_net(c, PVAE_NET_TE, S.rows_pad, st, te_tail))) return rc;
    ProSampler pro;
    memset(&pro, 0, sizeof(pro));
    if (S.fold_sampler) {                      // the sampler rides in the decoder's first-layer launch
        pro.te_out = w + wte.act.back(); pro.ldte = TE.layers.back().n_out_pad;
        pro.eps_in = eps; pro.eps_used = w + c->W.eps;
        pro.md_in = w + wmd.in; pro.ld_md = MD.layers[0].ld;
        pro.c0 = Db; pro.Z = Z; pro.rows = rows; pro.noise = 1;
        pro.seed = (unsigned long long)sp->rng_seed; pro.offset = (unsigned long long)sp->rng_offset;
        pro.partial = part + 2 * kLossParts;
    } else if ((rc = launch_sampler(c, w + wte.act.back(), TE.layers.back().n_out_pad, eps, w + c->W.eps, w + wmd.in,
                                    MD.layers[0].ld, rows, S.rows_pad, 1, (unsigned long long)sp->rng_seed,
                                    (unsigned long long)sp->rng_offset, part + 2 * kLossParts, (float*)nullptr,
                                    learned_prior ? w + wpr.act.back() : (const float*)nullptr,
                                    learned_prior ? PR.layers.back().n_out_pad : 0, st))) {
        return rc;
    }
    FwdTail md_tail;                           // a_hat also lands in the action columns of the WM input
    md_tail.out2 = w + wwm.in; md_tail.ld2 = WM.layers[0].ld; md_tail.off2 = Db; md_tail.n2 = Da;
    if (S.fold_sampler) md_tail.pro0 = &pro;
    if (dx) { xs_md = xsrc_of(c, PVAE_NET_MD, phase, !S.fold_sampler, rows); md_tail.xs0 = &xs_md; }
    if ((rc = forward_net(c, PVAE_NET_MD, S.rows_pad, st, md_tail))) return rc;
    const NetLayout& MH = c->L.net[PVAE_NET_MH];
    if (!MH.layers.empty()) {                  // rmt:833-835: the helper's term joins the action before anything reads it
        if ((rc = forward_net(c, PVAE_NET_MH, S.rows_pad, st))) return rc;
        const int grid = (rows * Da + 255) / 256 < 256 ? (rows * Da + 255) / 256 : 256;
        hipLaunchKernelGGL(helper_add_kernel, dim3(grid), dim3(256), 0, st, w + wmd.act.back(), MD.layers.back().n_out_pad,
                           w + c->W.net[PVAE_NET_MH].act.back(), MH.layers.back().n_out_pad, w + wwm.in, WM.layers[0].ld, Db,
                           rows, Da, c->L.cfg.mh_range);
        HIP_TRY(hipGetLastError());
    }
    // cycle loss (tpv:417-419) fused into the world model's output layer
    mse.grad_scale = sp->cycle_coeff * S.gs / (S.Bg * Db);
    mse.partial = part + 4 * kLossParts;
    return forward_net(c, PVAE_NET_WM, S.rows_pad, st, wm_tail);
}

// Everything after the forward pass, as stages.  (The action-reconstruction loss sits here: its
// gradient needs what came back through the frozen world model.)
static void plan_backward(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, bool backward, bool fused,
                          const StepShape& S, hipStream_t st, Plan& plan) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    if (c->W.L > 1) {
        plan_backward_unrolled(c, phase, rows, sp, backward, fused, S, st, plan);
        return;
    }
    float* w = c->ws;
    float* part = w + c->W.loss_part;
    const LossFinal* fold = S.lf.out ? &S.lf : nullptr;
    if (phase == PVAE_PHASE_WORLD) {
        if (backward) plan_backward_net(c, PVAE_NET_WM, S.rows_pad, true, false, sp, fused, st, fold, plan);
        return;
    }
    const NetLayout* TE = &c->L.net[PVAE_NET_TE];
    const NetLayout* MD = &c->L.net[PVAE_NET_MD];
    const NetLayout* WM = &c->L.net[PVAE_NET_WM];
    const NetWork* wte = &c->W.net[PVAE_NET_TE];
    const NetWork* wmd = &c->W.net[PVAE_NET_MD];
    const NetWork* wwm = &c->W.net[PVAE_NET_WM];
    const NetLayout* PR = &c->L.net[PVAE_NET_PR];
    const NetWork* wpr = &c->W.net[PVAE_NET_PR];
    const bool learned_prior = !PR->layers.empty();
    const bool sphere = c->L.cfg.prior_kind >= PVAE_PRIOR_HYPERSPHERE;     // (incl. NONE: the same kernel, not normalising)
    const int sphere_norm = c->L.cfg.prior_kind == PVAE_PRIOR_HYPERSPHERE ? 1 : 0;
    const int ldo_md = MD->layers.back().n_out_pad, ldo_te = TE->layers.back().n_out_pad;
    const float ga = sp->a_rec_coeff * S.gs / (S.Bg * Da);
    // The two gradient hand-overs between stacks live in the epilogue of the consuming stack's
    // first-layer input-gradient launch (InputSeed) whenever that launch exists and runs the paired
    // schedule; otherwise the stand-alone glue kernels do the same arithmetic.
    const bool seed_action = S.seed_action, seed_sampler = S.seed_sampler;
    if (S.cyc_grad) {                          // gradient through the frozen world model (dgrad only)
        InputSeed sd;
        if (seed_action) {
            sd.kind = 1;
            memset(&sd.a, 0, sizeof(sd.a));
            sd.a.pred = w + wmd->act.back(); sd.a.ldp = ldo_md;
            sd.a.target = w + c->W.act_t; sd.a.ldt = pad64(Da);
            if (c->dx.on) { sd.a.target = c->actions; sd.a.ldt = Da; sd.a.tind = 1; sd.a.trm = c->dx.rm; }     // a_t where it lies
            sd.a.dz = w + wmd->dz.back(); sd.a.ldz = ldo_md;
            sd.a.c0 = Db; sd.a.n = Da; sd.a.rows = rows;
            sd.a.grad_scale = ga; sd.a.l1 = S.l1;
            sd.a.partial = part + 1 * kLossParts;
        }
        plan_backward_net(c, PVAE_NET_WM, S.rows_pad, false, true, sp, fused, st, nullptr, plan, &sd);
    }
    // action reconstruction (tpv:381-382) + gradient arriving through the world model
    if (!seed_action) {
        const int nparts = S.nparts_a, rows_pad = S.rows_pad, l1 = S.l1;
        const bool cyc = S.cyc_grad;
        plan.emplace_back();
        plan.back().run = [=]() -> int {
            hipLaunchKernelGGL(mse_grad_kernel, dim3(nparts), dim3(256), 0, st, w + wmd->act.back(),
                               MD->layers.back().n_out_pad, w + c->W.act_t, pad64(Da),
                               backward ? w + wmd->dz.back() : nullptr, MD->layers.back().n_out_pad, rows, rows_pad,
                               Da, ga, cyc ? w + wwm->d_in : (const float*)nullptr, WM->layers[0].ld, Db,
                               part + 1 * kLossParts, l1);
            HIP_TRY(hipGetLastError());
            return 0;
        };
    }
    if (!backward) return;
    const NetLayout* MH = &c->L.net[PVAE_NET_MH];
    const NetWork* wmh = &c->W.net[PVAE_NET_MH];
    const bool helper = !MH->layers.empty();
    if (helper) {
        // d a_hat (just formed above: reconstruction + what came back through the world model) -> the helper's output layer,
        // then the helper's own backward: trained like the decoder (adam_t[PVAE_NET_MH] > 0) or passed through
        const int rows_pad = S.rows_pad, ldh = MH->layers.back().n_out_pad;
        const float range = c->L.cfg.mh_range;
        plan.emplace_back();
        plan.back().run = [=]() -> int {
            const int tot = rows_pad * ldh;
            hipLaunchKernelGGL(helper_seed_kernel, dim3((tot + 255) / 256 < 256 ? (tot + 255) / 256 : 256), dim3(256), 0, st,
                               w + wmd->dz.back(), ldo_md, w + wmh->act.back(), w + wmh->dz.back(), ldh, rows, rows_pad, Da, range);
            HIP_TRY(hipGetLastError());
            return 0;
        };
        plan_backward_net(c, PVAE_NET_MH, S.rows_pad, sp->adam_t[PVAE_NET_MH] > 0, true, sp, fused, st, nullptr, plan);
    }
    const float kls = S.kl_active ? sp->kl_coeff / S.Bg : 0.0f;
    InputSeed ss;
    if (seed_sampler) {
        ss.kind = 2;
        memset(&ss.s, 0, sizeof(ss.s));
        ss.s.te_out = w + wte->act.back(); ss.s.ldte = ldo_te;
        ss.s.eps = w + c->W.eps;
        ss.s.dz = w + wte->dz.back(); ss.s.ldz = ldo_te;
        ss.s.c0 = Db; ss.s.Z = Z; ss.s.rows = rows;
        ss.s.kl_scale = kls;
        if (learned_prior) {
            ss.s.mu_p = w + wpr->act.back(); ss.s.ldmp = PR->layers.back().n_out_pad;
            ss.s.dz_p = w + wpr->dz.back(); ss.s.ldzp = PR->layers.back().n_out_pad;
        }
    }
    CarriedWgrad carry;
    plan_backward_net(c, PVAE_NET_MD, S.rows_pad, true, true, sp, fused, st, nullptr, plan, &ss,
                      seed_sampler ? &carry : nullptr, nullptr,
                      /* hidden-layer pairs of the encoder follow the decoder's first-layer pair: */
                      !learned_prior && TE->layers.size() >= 3);
    if (!seed_sampler) {
        const int rows_pad = S.rows_pad;
        const int tot = rows_pad * TE->layers.back().n_out_pad;
        plan.emplace_back();
        plan.back().run = [=]() -> int {
            if (helper) {                      // z feeds the helper too: its input gradient joins the decoder's
                const int grid = (rows * Z + 255) / 256 < 256 ? (rows * Z + 255) / 256 : 256;
                hipLaunchKernelGGL(add_cols_kernel, dim3(grid), dim3(256), 0, st, w + wmd->d_in + Db, MD->layers[0].ld, rows, Z,
                                   (const float*)(w + wmh->d_in + Db), MH->layers[0].ld, (const float*)nullptr, 0,
                                   (const float*)nullptr, 0, (const float*)nullptr, 0);
                HIP_TRY(hipGetLastError());
            }
            if (sphere) {
                hipLaunchKernelGGL(sphere_bwd_kernel, dim3((rows_pad + 3) / 4), dim3(256), 0, st, w + wmd->d_in,
                                   MD->layers[0].ld, Db, w + wte->act.back(), TE->layers.back().n_out_pad, w + c->W.eps,
                                   w + wte->dz.back(), TE->layers.back().n_out_pad, rows, rows_pad, Z, kls, sphere_norm);
            } else {
                hipLaunchKernelGGL(reparam_bwd_kernel, dim3((tot + 255) / 256 < 256 ? (tot + 255) / 256 : 256), dim3(256), 0,
                                   st, w + wmd->d_in, MD->layers[0].ld, Db, w + wte->act.back(), TE->layers.back().n_out_pad,
                                   w + c->W.eps, w + wte->dz.back(), TE->layers.back().n_out_pad, rows, rows_pad, Z, kls,
                                   learned_prior ? w + wpr->act.back() : (const float*)nullptr,
                                   learned_prior ? PR->layers.back().n_out_pad : 0,
                                   learned_prior ? w + wpr->dz.back() : (float*)nullptr,
                                   learned_prior ? PR->layers.back().n_out_pad : 0);
            }
            HIP_TRY(hipGetLastError());
            return 0;
        };
    }
    // the learned prior mean trains through the KL term only (its output gradient was written beside the
    // encoder's by the sampler backward above); no input gradient
    if (learned_prior) plan_backward_net(c, PVAE_NET_PR, S.rows_pad, true, false, sp, fused, st, nullptr, plan);
    plan_backward_net(c, PVAE_NET_TE, S.rows_pad, true, false, sp, fused, st, fold, plan, nullptr, nullptr, &carry);
}


// ---------------------------------------------------------------------------------------
// lookahead > 1: the multi-step unroll of tpv:367-428
// ---------------------------------------------------------------------------------------
// Per step t: x_t = [s1_t | s2gt_t] -> encoder -> sampler -> decoder -> world model with the
// decoder's action (its output is both the cycle-loss prediction and s1_{t+1}, tpv:417-421) and,
// when world_model_s_rec_coeff > 0, the world model with the demonstrated action (tpv:411-414).
// All of it runs in BOTH phases (the world phase needs the chain because s1_{t+1} is a
// prediction), in row block t of every panel; the world model uses block t for the
// demonstrated-action invocation and block L+t for the predicted-action one.
struct Unroll {
    int T, rows, rows_pad;
    bool use_g;                        // demonstrated-action world-model invocations exist
    int64_t blk(int slot) const { return (int64_t)slot * rows_pad; }
};

static Unroll make_unroll(const pvae_ctx* c, int phase, int rows, const pvae_step_params* sp) {
    Unroll u;
    u.T = c->W.L; u.rows = rows; u.rows_pad = pad32(rows);
    u.use_g = phase == PVAE_PHASE_WORLD && sp->s_rec_coeff > 0.0f;
    return u;
}

static int run_forward_unrolled(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, const float* eps,
                                bool backward, const StepShape& S, hipStream_t st) {
    int rc;
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    const Unroll u = make_unroll(c, phase, rows, sp);
    float* w = c->ws;
    float* part = w + c->W.loss_part;
    const NetLayout& TE = c->L.net[PVAE_NET_TE];
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    const NetLayout& WM = c->L.net[PVAE_NET_WM];
    const NetWork& wte = c->W.net[PVAE_NET_TE];
    const NetWork& wmd = c->W.net[PVAE_NET_MD];
    const NetWork& wwm = c->W.net[PVAE_NET_WM];
    const int ld_te = TE.layers[0].ld, ld_md = MD.layers[0].ld, ld_wm = WM.layers[0].ld;
    const int ldo_te = TE.layers.back().n_out_pad, ldo_wm = WM.layers.back().n_out_pad;
    const bool joint = phase == PVAE_PHASE_JOINT;
    for (int t = 0; t < u.T; ++t) {
        const int64_t bt = u.blk(t), bp = u.blk(u.T + t);
        if ((rc = forward_net(c, PVAE_NET_TE, u.rows_pad, st, FwdTail(), bt))) return rc;
        hipLaunchKernelGGL(reparam_kernel, dim3(S.gridz), dim3(256), 0, st, w + wte.act.back() + bt * ldo_te, ldo_te,
                           eps ? eps + (size_t)t * rows * Z : (const float*)nullptr, w + c->W.eps + bt * Z,
                           w + z_panel(c) + bt * ld_md, ld_md, Db, Z, rows, u.rows_pad, 1,
                           (unsigned long long)sp->rng_seed, (unsigned long long)(sp->rng_offset + t),
                           part + 2 * kLossParts + t * S.gridz, (float*)nullptr);
        HIP_TRY(hipGetLastError());
        FwdTail md_tail;                       // a_hat -> action columns of the predicted-action WM input
        md_tail.out2 = w + wwm.in + bp * ld_wm; md_tail.ld2 = ld_wm; md_tail.off2 = Db; md_tail.n2 = Da;
        if ((rc = forward_net(c, PVAE_NET_MD, u.rows_pad, st, md_tail, bt))) return rc;
        if (!c->L.net[PVAE_NET_MH].layers.empty()) {        // rmt:833-835 in every unrolled step: a_hat_t += range * helper(x_t)
            const NetLayout& MH = c->L.net[PVAE_NET_MH];
            const int ldo_md = MD.layers.back().n_out_pad, ldo_mh = MH.layers.back().n_out_pad;
            if ((rc = forward_net(c, PVAE_NET_MH, u.rows_pad, st, FwdTail(), bt))) return rc;
            const int grid = (rows * Da + 255) / 256 < 256 ? (rows * Da + 255) / 256 : 256;
            hipLaunchKernelGGL(helper_add_kernel, dim3(grid), dim3(256), 0, st, w + wmd.act.back() + bt * ldo_md, ldo_md,
                               w + c->W.net[PVAE_NET_MH].act.back() + bt * ldo_mh, ldo_mh, w + wwm.in + bp * ld_wm, ld_wm, Db,
                               rows, Da, c->L.cfg.mh_range);
            HIP_TRY(hipGetLastError());
        }
        EpiMse mse;
        memset(&mse, 0, sizeof(mse));
        mse.target = w + c->W.s2 + bt * pad64(Db); mse.ldt = pad64(Db);
        mse.ldz = ldo_wm; mse.rows = rows; mse.D = Db; mse.l1 = S.l1;
        FwdTail wm_tail;
        wm_tail.mse = &mse;
        // predicted action: cycle loss (tpv:417-419) + the state of the next step
        mse.dz = backward ? w + wwm.dz.back() + bp * ldo_wm : nullptr;
        mse.grad_scale = joint ? sp->cycle_coeff * S.gs / (S.Bg * Db) : 0.0f;
        mse.partial = part + 4 * kLossParts + t * S.wm_tiles;
        if ((rc = forward_net(c, PVAE_NET_WM, u.rows_pad, st, wm_tail, bp))) return rc;
        if (u.use_g) {                         // demonstrated action: state reconstruction (tpv:411-414)
            mse.dz = backward ? w + wwm.dz.back() + bt * ldo_wm : nullptr;
            mse.grad_scale = sp->s_rec_coeff * S.gs / (S.Bg * Db);
            mse.partial = part + 3 * kLossParts + t * S.wm_tiles;
            if ((rc = forward_net(c, PVAE_NET_WM, u.rows_pad, st, wm_tail, bt))) return rc;
        }
        if (t + 1 < u.T) {                     // s1 of the next step (tpv:421)
            const int64_t nt = u.blk(t + 1), np = u.blk(u.T + t + 1);
            const int grid = (rows * Db + 255) / 256 < 256 ? (rows * Db + 255) / 256 : 256;
            hipLaunchKernelGGL(scatter_state_kernel, dim3(grid), dim3(256), 0, st,
                               w + wwm.act.back() + bp * ldo_wm, ldo_wm, rows, Db,
                               // (input subsets: a stack that does not read s_t keeps zeros there)
                               c->L.cfg.te_inputs == PVAE_INPUT_TASK ? (float*)nullptr : w + wte.in + nt * ld_te, ld_te,
                               c->L.cfg.md_inputs == PVAE_INPUT_TASK ? (float*)nullptr : w + wmd.in + nt * ld_md, ld_md,
                               w + wwm.in + np * ld_wm, ld_wm,
                               u.use_g ? w + wwm.in + nt * ld_wm : (float*)nullptr, ld_wm);
            HIP_TRY(hipGetLastError());
        }
    }
    return 0;
}

// Backward through the unroll, last step first.  Input gradients are needed in full here (the
// current-state columns of every consumer feed the previous step), weight gradients contract over
// ALL steps at once: the time-step blocks are stacked along the row axis, so one launch per layer
// with K = blocks * rows_pad yields sum_t X_t^T dZ_t (and Adam runs once, in its epilogue).
static void plan_backward_unrolled(pvae_ctx* c, int phase, int rows, const pvae_step_params* sp, bool backward,
                                   bool fused, const StepShape& S, hipStream_t st, Plan& plan) {
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    const Unroll u = make_unroll(c, phase, rows, sp);
    const int T = u.T;
    float* w = c->ws;
    float* part = w + c->W.loss_part;
    const bool joint = phase == PVAE_PHASE_JOINT;
    const NetLayout* NL = c->L.net;
    const NetWork* NW = c->W.net;
    const double rowsf = c->staged_rows_f;
    auto push = [&](std::function<int()> f) -> Stage& {
        plan.emplace_back();
        plan.back().run = std::move(f);
        return plan.back();
    };
    // which invocations carry gradient (evaluated last step first)
    const bool a_grad = joint && sp->a_rec_coeff > 0.0f;
    std::vector<char> p_act(T, 0), md_act(T, 0), any(T + 1, 0);
    for (int t = T - 1; t >= 0; --t) {
        p_act[t] = S.cyc_grad || (t + 1 < T && any[t + 1]);
        md_act[t] = a_grad || p_act[t];
        any[t] = md_act[t] || p_act[t] || u.use_g;
    }
    // full dgrad chain of net n over row block `slot`; layer 0 only when its input gradient is consumed
    auto dgrad_chain = [&](int n, int slot, bool layer0) {
        const NetLayout* N = &NL[n];
        const NetWork* nw = &NW[n];
        const int64_t b = u.blk(slot);
        const int rows_pad = u.rows_pad;
        for (int i = (int)N->layers.size() - 1; i >= (layer0 ? 0 : 1); --i) {
            push([=]() -> int {
                const Layer& l = N->layers[i];
                const float* mask = i > 0 ? w + nw->act[i - 1] + b * l.ld : nullptr;
                float* out = i > 0 ? w + nw->dz[i - 1] + b * l.ld : w + nw->d_in + b * l.ld;
                const int ps = g_prof.begin(1, 2.0 * rowsf * l.n_in * l.n_out, st);
                HIP_TRY(gemm_dgrad(w + nw->dz[i] + b * l.n_out_pad, l.n_out_pad, c->params + l.w_off, l.ld, mask, l.ld,
                                   out, l.ld, rows_pad, l.ld, l.n_out_pad, st, i > 0 ? N->layers[i - 1].act : 1));
                g_prof.end(ps, st);
                return 0;
            });
        }
    };
    const int ld_te = NL[PVAE_NET_TE].layers[0].ld, ld_md = NL[PVAE_NET_MD].layers[0].ld;
    const int ld_wm = NL[PVAE_NET_WM].layers[0].ld;
    const int ldo_te = NL[PVAE_NET_TE].layers.back().n_out_pad, ldo_md = NL[PVAE_NET_MD].layers.back().n_out_pad;
    const int ldo_wm = NL[PVAE_NET_WM].layers.back().n_out_pad;
    const NetWork* wte = &NW[PVAE_NET_TE];
    const NetWork* wmd = &NW[PVAE_NET_MD];
    const NetWork* wwm = &NW[PVAE_NET_WM];

    // Weight gradients contract over ALL steps at once (row blocks stacked): rows [0, krows) of a trainable stack.
    // Row blocks that receive no gradient are cut off the end or zero-filled; the fills go first (nothing writes
    // those blocks afterwards).
    // The motor decoder's helper (rmt:670-680, 833-835) sits in every step's a_hat_t.  Nothing in the trainer freezes it
    // (tpv:326-329, 347-350), and with lookahead > 1 the WORLD phase reaches it as well: the state the world model continues
    // from is its own prediction under the helped action (tpv:417-421).  So it is a trainable stack of both phases here
    // (adam_t[PVAE_NET_MH] == 0: frozen for this step), and the world phase's step 0 -- whose frozen decoder and encoder
    // lead nowhere -- still has to bring the action's gradient to it.
    const bool helper = !NL[PVAE_NET_MH].layers.empty();
    const bool mh_train = helper && sp->adam_t[PVAE_NET_MH] > 0;
    const NetWork* wmh = &NW[PVAE_NET_MH];
    std::vector<int> train_nets;
    if (mh_train) train_nets.push_back(PVAE_NET_MH);
    if (joint) { train_nets.push_back(PVAE_NET_MD); train_nets.push_back(PVAE_NET_TE); }
    else train_nets.push_back(PVAE_NET_WM);
    int krows_of[PVAE_NUM_NETS] = {};
    if (backward) {
        for (int n : train_nets) {
            const NetLayout* N = &NL[n];
            const NetWork* nw = &NW[n];
            std::vector<char> act;
            if (n == PVAE_NET_WM) {
                for (int t = 0; t < T; ++t) act.push_back(u.use_g);
                for (int t = 0; t < T; ++t) act.push_back(p_act[t]);
            } else {                               // (decoder, encoder, helper: one block per step that the action's gradient reaches)
                for (int t = 0; t < T; ++t) act.push_back(md_act[t]);
            }
            int blocks = (int)act.size();
            while (blocks > 0 && !act[blocks - 1]) --blocks;
            for (int b = 0; b < blocks; ++b) {
                if (act[b]) continue;
                const int64_t r0 = u.blk(b);
                const size_t nrows = (size_t)u.rows_pad;
                push([=]() -> int {
                    for (const Layer& l : N->layers)
                        HIP_TRY(hipMemsetAsync(w + nw->dz[l.index] + r0 * l.n_out_pad, 0, nrows * l.n_out_pad * sizeof(float), st));
                    return 0;
                });
            }
            krows_of[n] = blocks * u.rows_pad;
        }
    }
    const LossFinal* fold = S.lf.out ? &S.lf : nullptr;
    // Step 0's input-gradient launches of a trainable stack run LAST in the backward pass, so by the time layer i's
    // input gradient of step 0 is launched, dz[i] is final for every step: its weight gradient (over all steps) can
    // share that launch -- the same-layer pairing of the lookahead-1 schedule (gradient stored, Adam deferred to
    // workgroups of the next launch), instead of 3 weight-gradient launches per stack at the end (PVAE_LOOK_PAIR=0).
    const bool look_pair_env = g_look_pair;
    const bool can_defer = fused && c->defer_adam && c->grads != nullptr;
    const bool look_pair = backward && look_pair_env && c->pair_launch && c->same_layer_pairs && (!fused || can_defer);
    bool paired_done[PVAE_NUM_NETS] = {};
    // layers last .. lo of stack n: dgrad_i over row block `slot` || wgrad_i over rows [0, krows); then, when lo == 1,
    // layer 0's weight gradient on its own.  `with_fold`: the stack's last launch also finalises the losses.
    auto paired_chain = [&](int n, int slot, int lo, bool with_fold) {
        const NetLayout* N = &NL[n];
        const NetWork* nw = &NW[n];
        const int64_t b = u.blk(slot);
        const int rows_pad = u.rows_pad, krows = krows_of[n];
        const AdamScalars as = adam_scalars(sp, n);
        LossFinal foldv;
        memset(&foldv, 0, sizeof(foldv));
        if (with_fold && fold) foldv = *fold;
        for (int i = (int)N->layers.size() - 1; i >= 0; --i) {
            const bool has_d = i >= lo;
            const bool f = with_fold && fold && i == 0;
            Stage& sref = push([=]() -> int {
                const Layer& l = N->layers[i];
                const float* mask = i > 0 ? w + nw->act[i - 1] + b * l.ld : nullptr;
                float* out = i > 0 ? w + nw->dz[i - 1] + b * l.ld : w + nw->d_in + b * l.ld;
                const float* dz = w + nw->dz[i];
                const float* xin = i == 0 ? w + nw->in : w + nw->act[i - 1];
                EpiGradAdam ea{c->params + l.w_off, c->m + l.w_off, c->v + l.w_off, l.ld, as};
                ea.b = c->params + l.b_off; ea.bm = c->m + l.b_off; ea.bv = c->v + l.b_off;
                EpiGradStore es{c->grads + l.w_off, l.ld};
                es.gb = c->grads + l.b_off;
                if (f) { ea.loss = foldv; es.loss = foldv; }
                const AdamPair ad = take_pending(c);
                hipError_t he;
                if (has_d) {
                    const int pp = g_prof.begin(3, 2.0 * rowsf * l.n_in * l.n_out * (1.0 + (double)krows / rows_pad), st);
                    he = gemm_bwd_pair(dz + b * l.n_out_pad, l.n_out_pad, c->params + l.w_off, l.ld, mask, l.ld, out, l.ld,
                                       rows_pad, l.ld, l.n_out_pad, dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, krows, es,
                                       st, &ad, i > 0 ? N->layers[i - 1].act : 1);
                    g_prof.end(pp, st);
                    if (he == hipSuccess && fused) {                 // (this launch read W_i: its update waits for the next one)
                        AdamSeg a;
                        a.p = c->params + l.w_off; a.g = c->grads + l.w_off; a.m = c->m + l.w_off; a.v = c->v + l.w_off;
                        a.n4 = (l.b_off + l.n_out_pad - l.w_off) / 4;
                        a.s = as;
                        c->pending_adam = a;
                    }
                } else {
                    const int pw = g_prof.begin(2, 2.0 * rowsf * l.n_in * l.n_out * ((double)krows / rows_pad), st);
                    he = fused ? gemm_wgrad(dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, krows, ea, st, &ad)
                               : gemm_wgrad(dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, krows, es, st, &ad);
                    g_prof.end(pw, st);
                }
                if (he != hipSuccess) return fail(-10, "paired backward launch: %s", hipGetErrorString(he));
                return 0;
            });
            sref.ready_off = N->layers[i].w_off;
            sref.ready_cnt = N->layers[i].b_off + N->layers[i].n_out_pad - N->layers[i].w_off;
            sref.net = n;
        }
        paired_done[n] = true;
    };
    const int last_train = train_nets.back();
    for (int t = T - 1; t >= 0; --t) {
        const int64_t bt = u.blk(t), bp = u.blk(T + t);
        const int rows_pad = u.rows_pad;
        // step 0 in the WORLD phase: what flows back through the (frozen) decoder and encoder of step 0 reaches no
        // trainable parameter -- only the world model's own layers above layer 0 need their input gradients
        const bool upstream = joint || t > 0;
        const bool up_a = upstream || mh_train;   // the action's gradient of this step is wanted (by the helper, if by nobody else)
        const bool pair_wm = look_pair && t == 0 && !joint && krows_of[PVAE_NET_WM] > 0;
        if (backward && u.use_g) {
            if (pair_wm && !p_act[t]) paired_chain(PVAE_NET_WM, t, 1, true);    // (no predicted-action chain follows)
            else dgrad_chain(PVAE_NET_WM, t, t > 0);
        }
        if (backward && p_act[t]) {
            if (t + 1 < T && any[t + 1]) {        // + gradient wrt s1_{t+1}, from every consumer of it
                const int64_t nt = u.blk(t + 1), np = u.blk(T + t + 1);
                const float* s_te = md_act[t + 1] ? w + wte->d_in + nt * ld_te : nullptr;
                const float* s_md = md_act[t + 1] ? w + wmd->d_in + nt * ld_md : nullptr;
                const float* s_p = p_act[t + 1] ? w + wwm->d_in + np * ld_wm : nullptr;
                const float* s_g = u.use_g ? w + wwm->d_in + nt * ld_wm : nullptr;
                push([=]() -> int {
                    const int grid = (rows * Db + 255) / 256 < 256 ? (rows * Db + 255) / 256 : 256;
                    hipLaunchKernelGGL(add_cols_kernel, dim3(grid), dim3(256), 0, st, w + wwm->dz.back() + bp * ldo_wm,
                                       ldo_wm, rows, Db, s_te, ld_te, s_md, ld_md, s_p, ld_wm, s_g, ld_wm);
                    HIP_TRY(hipGetLastError());
                    return 0;
                });
            }
            if (pair_wm) paired_chain(PVAE_NET_WM, T + t, up_a ? 0 : 1, true);
            else dgrad_chain(PVAE_NET_WM, T + t, up_a);
        }
        if (!up_a) continue;
        // action reconstruction (tpv:381-382) + gradient arriving through the world model
        if (md_act[t] || (joint && sp->a_rec_coeff > 0.0f)) {
            const float ga = joint ? sp->a_rec_coeff * S.gs / (S.Bg * Da) : 0.0f;
            const int nparts = S.nparts_a, l1 = S.l1;
            const bool extra = p_act[t] && backward;
            push([=]() -> int {
                hipLaunchKernelGGL(mse_grad_kernel, dim3(nparts), dim3(256), 0, st, w + wmd->act.back() + bt * ldo_md,
                                   ldo_md, w + c->W.act_t + bt * pad64(Da), pad64(Da),
                                   backward ? w + wmd->dz.back() + bt * ldo_md : (float*)nullptr, ldo_md, rows, rows_pad,
                                   Da, ga, extra ? w + wwm->d_in + bp * ld_wm : (const float*)nullptr, ld_wm, Db,
                                   part + 1 * kLossParts + t * nparts, l1);
                HIP_TRY(hipGetLastError());
                return 0;
            });
        }
        if (!backward || !md_act[t]) continue;
        if (helper) {
            // d a_hat_t (just formed) -> the helper's output layer through range * tanh', then its own layers; its input
            // gradient matters where the decoder's does (z -> encoder, s1_t -> the previous step)
            const int ldh = NL[PVAE_NET_MH].layers.back().n_out_pad;
            const float range = c->L.cfg.mh_range;
            push([=]() -> int {
                const int tot = rows_pad * ldh;
                hipLaunchKernelGGL(helper_seed_kernel, dim3((tot + 255) / 256 < 256 ? (tot + 255) / 256 : 256), dim3(256), 0, st,
                                   w + wmd->dz.back() + bt * ldo_md, ldo_md, w + wmh->act.back() + bt * ldh,
                                   w + wmh->dz.back() + bt * ldh, ldh, rows, rows_pad, Da, range);
                HIP_TRY(hipGetLastError());
                return 0;
            });
            dgrad_chain(PVAE_NET_MH, t, upstream);
        }
        if (!upstream) continue;
        const bool pair_here = look_pair && t == 0 && joint;
        if (pair_here && krows_of[PVAE_NET_MD] > 0) paired_chain(PVAE_NET_MD, t, 0, false);
        else dgrad_chain(PVAE_NET_MD, t, true);
        if (helper) {                              // x_t = [s1_t | z_t] feeds the helper too: its input gradient joins the decoder's
            const int ld_mh = NL[PVAE_NET_MH].layers[0].ld;
            push([=]() -> int {
                const int n = Db + Z;
                const int grid = (rows * n + 255) / 256 < 256 ? (rows * n + 255) / 256 : 256;
                hipLaunchKernelGGL(add_cols_kernel, dim3(grid), dim3(256), 0, st, w + wmd->d_in + bt * ld_md, ld_md, rows, n,
                                   (const float*)(w + wmh->d_in + bt * ld_mh), ld_mh, (const float*)nullptr, 0,
                                   (const float*)nullptr, 0, (const float*)nullptr, 0);
                HIP_TRY(hipGetLastError());
                return 0;
            });
        }
        {
            const float kls = S.kl_active ? sp->kl_coeff / S.Bg : 0.0f;
            const int tot = rows_pad * ldo_te;
            push([=]() -> int {
                hipLaunchKernelGGL(reparam_bwd_kernel, dim3((tot + 255) / 256 < 256 ? (tot + 255) / 256 : 256), dim3(256),
                                   0, st, w + wmd->d_in + bt * ld_md, ld_md, Db, w + wte->act.back() + bt * ldo_te, ldo_te,
                                   w + c->W.eps + bt * Z, w + wte->dz.back() + bt * ldo_te, ldo_te, rows, rows_pad, Z,
                                   kls);
                HIP_TRY(hipGetLastError());
                return 0;
            });
        }
        if (pair_here && krows_of[PVAE_NET_TE] > 0) paired_chain(PVAE_NET_TE, t, 1, true);
        else dgrad_chain(PVAE_NET_TE, t, t > 0);
    }
    if (!backward) return;
    (void)last_train;

    // weight gradients of the stacks whose launches were not paired above: one contraction per layer over the
    // stacked blocks
    for (size_t k = 0; k < train_nets.size(); ++k) {
        const int n = train_nets[k];
        if (paired_done[n]) continue;
        const NetLayout* N = &NL[n];
        const NetWork* nw = &NW[n];
        const int blocks = krows_of[n] / u.rows_pad;
        const int krows = krows_of[n];
        const AdamScalars as = adam_scalars(sp, n);
        const bool last_net = k + 1 == train_nets.size();
        // two layers per launch (wgrad_pair_kernel), last layer first; an odd layer count leaves layer 0
        // on its own.  The launch that contains layer 0 of the last net also finalises the losses.
        const bool pairs = c->pair_launch && krows > 0;
        for (int i = (int)N->layers.size() - 1; i >= 0;) {
            const int j = (pairs && i >= 1) ? i - 1 : -1;             // second layer of this launch
            const int lo = j >= 0 ? j : i;
            const bool with_fold = fold && last_net && lo == 0;
            LossFinal foldv;
            memset(&foldv, 0, sizeof(foldv));
            if (with_fold) foldv = *fold;
            Stage& sref = push([=]() -> int {
                const Layer& l = N->layers[i];
                const float* dz = w + nw->dz[i];
                const float* xin = i == 0 ? w + nw->in : w + nw->act[i - 1];
                double fl = 2.0 * rowsf * blocks * l.n_in * l.n_out;
                if (j >= 0) fl += 2.0 * rowsf * blocks * N->layers[j].n_in * N->layers[j].n_out;
                const int pw = g_prof.begin(2, fl, st);
                int rc2 = 0;
                auto adam_of = [&](const Layer& y) {
                    EpiGradAdam e{c->params + y.w_off, c->m + y.w_off, c->v + y.w_off, y.ld, as};
                    e.b = c->params + y.b_off; e.bm = c->m + y.b_off; e.bv = c->v + y.b_off;
                    return e;
                };
                auto store_of = [&](const Layer& y) {
                    EpiGradStore e{c->grads + y.w_off, y.ld};
                    e.gb = c->grads + y.b_off;
                    return e;
                };
                auto go = [&](auto e1, auto e2) -> hipError_t {
                    if (with_fold) e1.loss = foldv;                   // block 0 of the launch runs e1's problem
                    if (j < 0) return gemm_wgrad(dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, krows, e1, st);
                    const Layer& l2 = N->layers[j];
                    const float* dz2 = w + nw->dz[j];
                    const float* xin2 = j == 0 ? w + nw->in : w + nw->act[j - 1];
                    return gemm_wgrad_pair(dz, l.n_out_pad, xin, l.ld, l.n_out_pad, l.ld, e1, dz2, l2.n_out_pad, xin2, l2.ld,
                                           l2.n_out_pad, l2.ld, e2, krows, st);
                };
                if (krows > 0) {                   // (0: nothing reached this net, its gradient stays as it is)
                    const Layer& l2 = N->layers[j >= 0 ? j : i];
                    const hipError_t he = fused ? go(adam_of(l), adam_of(l2)) : go(store_of(l), store_of(l2));
                    if (he != hipSuccess) rc2 = fail(-10, "weight-gradient launch: %s", hipGetErrorString(he));
                }
                g_prof.end(pw, st);
                return rc2;
            });
            sref.ready_off = N->layers[lo].w_off;
            sref.ready_cnt = N->layers[i].b_off + N->layers[i].n_out_pad - N->layers[lo].w_off;
            sref.net = n;
            i = lo - 1;
        }
    }
}

extern "C" {

int pvae_forward_backward(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, const float* eps,
                          float* loss_out, int flags, void* stream) {
    const bool backward = !(flags & PVAE_FLAG_NO_BACKWARD);
    const bool fused = (flags & PVAE_FLAG_FUSED_ADAM) != 0;
    int rc = check_step(c, phase, rows, sp, backward, fused);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (backward && fused) params_touched(c, st);
    StepShape S;
    if ((rc = step_shape(c, phase, rows, sp, loss_out, backward, S))) return rc;
    if ((rc = run_forward(c, phase, rows, sp, eps, backward, S, st))) return rc;
    Plan plan;
    plan_backward(c, phase, rows, sp, backward, fused, S, st, plan);
    c->pending_adam = c->held_adam = AdamSeg();
    for (Stage& s : plan)
        if ((rc = s.run())) { c->pending_adam = c->held_adam = AdamSeg(); return rc; }
    if ((rc = flush_pending_adam(c, st))) return rc;
    if (loss_out && !backward) {
        hipLaunchKernelGGL(finalize_loss_kernel, dim3(1), dim3(64), 0, st, S.lf);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int pvae_forward_seed(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, const float* eps,
                      void* stream) {
    int rc = check_step(c, phase, rows, sp, true, false);
    if (rc) return rc;
    StepShape S;
    if ((rc = step_shape(c, phase, rows, sp, nullptr, true, S))) return rc;
    return run_forward(c, phase, rows, sp, eps, true, S, (hipStream_t)stream);
}

int pvae_backward_stage(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, int stage,
                        float* loss_out, void* stream, int64_t* ready_offset, int64_t* ready_count,
                        int* ready_net, int* num_stages) {
    int rc = check_step(c, phase, rows, sp, true, false);
    if (rc) return rc;
    StepShape S;
    if ((rc = step_shape(c, phase, rows, sp, loss_out, true, S))) return rc;
    Plan plan;
    plan_backward(c, phase, rows, sp, true, false, S, (hipStream_t)stream, plan);
    if (num_stages) *num_stages = (int)plan.size();
    if (stage < 0 || stage >= (int)plan.size()) return fail(-1, "stage %d outside [0, %d)", stage, (int)plan.size());
    if (ready_offset) *ready_offset = plan[stage].ready_off;
    if (ready_count) *ready_count = plan[stage].ready_cnt;
    if (ready_net) *ready_net = plan[stage].net;
    return plan[stage].run();
}

int pvae_backward_plan(pvae_ctx* c, int phase, const pvae_step_params* sp, int64_t* offset, int64_t* count, int* net,
                       int max, int* num_stages) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!sp) return fail(-1, "null step params");
    if (phase != PVAE_PHASE_WORLD && phase != PVAE_PHASE_JOINT) return fail(-1, "unknown phase %d", phase);
    StepShape S;
    if ((rc = step_shape(c, phase, 1, sp, nullptr, true, S))) return rc;
    Plan plan;                                  // (stages are closures: building them launches nothing and changes no state)
    plan_backward(c, phase, 1, sp, true, false, S, (hipStream_t) nullptr, plan);
    if (num_stages) *num_stages = (int)plan.size();
    for (int k = 0; k < (int)plan.size() && k < max; ++k) {
        if (offset) offset[k] = plan[k].ready_off;
        if (count) count[k] = plan[k].ready_cnt;
        if (net) net[k] = plan[k].net;
    }
    return 0;
}

int pvae_adam_segment(pvae_ctx* c, int net, int64_t offset, int64_t count, const pvae_step_params* sp,
                      void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!sp) return fail(-1, "null step params");
    if (!c->grads || !c->m || !c->v) return fail(-2, "grads / Adam moment arenas not bound");
    if (net < 0 || net >= PVAE_NUM_NETS) return fail(-1, "bad net id %d", net);
    const NetLayout& N = c->L.net[net];
    if (offset < N.off || count < 0 || offset + count > N.off + N.count || (offset & 3) || (count & 3))
        return fail(-1, "segment [%lld, +%lld) not inside net %d or not float4-aligned", (long long)offset,
                    (long long)count, net);
    if (count == 0) return 0;
    params_touched(c, (hipStream_t)stream);
    const long long n4 = count / 4;
    int grid = (int)((n4 + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, c->params + offset,
                       c->grads + offset, c->m + offset, c->v + offset, n4, adam_scalars(sp, net));
    HIP_TRY(hipGetLastError());
    return 0;
}

int pvae_adam(pvae_ctx* c, int net_mask, const pvae_step_params* sp, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!sp) return fail(-1, "null step params");
    if (!c->grads || !c->m || !c->v) return fail(-2, "grads / Adam moment arenas not bound");
    params_touched(c, (hipStream_t)stream);
    for (int n = 0; n < PVAE_NUM_NETS; ++n) {
        if (!(net_mask & (1 << n))) continue;
        const NetLayout& N = c->L.net[n];
        if (N.count == 0) continue;
        const long long n4 = N.count / 4;      // segments are multiples of 64 floats
        int grid = (int)((n4 + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(adam_flat_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, c->params + N.off,
                           c->grads + N.off, c->m + N.off, c->v + N.off, n4, adam_scalars(sp, n));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

static void flip_stage_panels(pvae_ctx* c);

static RowMap row_map(const pvae_ctx* c, int64_t first_window, int rows) {
    RowMap rm;
    rm.row = c->window_row + first_window;
    rm.seg = 0; rm.q1 = rows; rm.b0 = rm.b1 = 0;
    if ((int64_t)c->window_row_host.size() == c->n_windows) {          // at most one jump inside the minibatch: two runs
        const int32_t* wr = c->window_row_host.data() + first_window;
        int jumps = 0, at = rows;
        for (int q = 1; q < rows && jumps < 2; ++q)
            if (wr[q] != wr[q - 1] + 1) { ++jumps; at = q; }
        if (jumps < 2) { rm.seg = 1; rm.q1 = at; rm.b0 = wr[0]; rm.b1 = at < rows ? wr[at] : 0; }
    }
    return rm;
}
// the rows the NEXT minibatch's first layers will gather, as runs of 128-byte lines for the last launch of this step to touch
static void plan_touch(pvae_ctx* c, int64_t next_first, int next_rows) {
    memset(&c->next_touch, 0, sizeof(c->next_touch));
    if (next_rows <= 0 || next_first < 0 || next_first + next_rows > c->n_windows) return;
    const RowMap rm = row_map(c, next_first, next_rows);
    if (!rm.seg) return;
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action;
    const int n[2] = {rm.q1, next_rows - rm.q1}, b[2] = {rm.b0, rm.b1};
    int total = 0;
    for (int s = 0; s < 2; ++s) {
        if (n[s] <= 0) continue;
        c->next_touch.p[2 * s] = c->states + (size_t)b[s] * Db;
        c->next_touch.lines[2 * s] = (int)(((size_t)(n[s] + 1) * Db * 4 + 127) / 128);     // (+ 1: s_{t+1} of the run's last window)
        c->next_touch.p[2 * s + 1] = c->actions + (size_t)b[s] * Da;
        c->next_touch.lines[2 * s + 1] = (int)(((size_t)n[s] * Da * 4 + 127) / 128);
        total += c->next_touch.lines[2 * s] + c->next_touch.lines[2 * s + 1];
    }
    c->next_touch.blocks = total > 0 ? (total + 255) / 256 : 0;
    if (c->next_touch.blocks > 64) c->next_touch.blocks = 64;
}
// A training step that reads the demonstration set directly: nothing is staged; `dx` tells run_forward /
// plan_backward_net to use the gathered first layers.  -> false: the step takes the staging launch as before.
static bool enter_direct(pvae_ctx* c, int phase, int64_t first_window, int rows, const pvae_step_params* sp, bool fused) {
    c->dx.on = false;
    if (!sp || !c->states || first_window < 0 || rows < 1 || rows > c->L.cfg.max_batch || first_window + rows > c->n_windows) return false;
    if (phase != PVAE_PHASE_WORLD && phase != PVAE_PHASE_JOINT) return false;
    if (!direct_ok(c, phase, rows, sp, fused)) return false;
    const RowMap rm = row_map(c, first_window, rows);
    if (!rm.seg) return false;             // (more than one episode jump inside the minibatch, or no host copy of window_row)
    c->dx.on = true;
    c->dx.rm = rm;
    memset(&c->next_touch, 0, sizeof(c->next_touch));
    c->staged_rows = rows;
    c->staged_rows_f = rows;
    c->pf.valid = false;
    c->next_stage.rows_pad = 0;
    c->next_carried = false;
    return true;
}
static void leave_direct(pvae_ctx* c) {
    c->dx.on = false;
    c->staged_rows = 0;          // the input panels do not hold this minibatch: a later pvae_forward_backward must stage first
}
int pvae_direct_active(pvae_ctx* c, int phase, int32_t rows, const pvae_step_params* sp, int fused) {
    if (!c || !sp) return fail(-1, "null argument");
    if (check_ready(c, true)) return 0;
    return c->states && direct_ok(c, phase, rows, sp, fused != 0) ? 1 : 0;
}

int pvae_dp_train_step(pvae_ctx* c, int phase, int64_t first_window, int32_t rows, const pvae_step_params* sp,
                       const float* eps, float* loss_out, int64_t next_first, int32_t next_rows, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!c->comm && !(c->p2p.open && (c->exchange_mode == PVAE_EXCHANGE_P2P || c->exchange_mode == PVAE_EXCHANGE_P2P_PUSH ||
                                      c->exchange_mode == PVAE_EXCHANGE_LOCAL)))
        return fail(-2, "no communicator (pvae_comm_init) and no peer-mapped exchange (pvae_p2p_open + pvae_comm_mode)");
    if (!sp) return fail(-1, "null step params");
    if (!c->grads || !c->m || !c->v) return fail(-2, "grads / Adam moment arenas not bound");
    if (phase != PVAE_PHASE_WORLD && phase != PVAE_PHASE_JOINT) return fail(-1, "unknown phase %d", phase);
    hipStream_t st = (hipStream_t)stream;
    params_touched(c, st);
    const bool learned_prior = !c->L.net[PVAE_NET_PR].layers.empty();
    const bool helper = !c->L.net[PVAE_NET_MH].layers.empty() && sp->adam_t[PVAE_NET_MH] > 0;
    // (the helper trains with the decoder in the joint phase and, with lookahead > 1, in the world phase too: plan_backward_unrolled)
    const int nets[4] = {(phase == PVAE_PHASE_WORLD && c->W.L == 1) || !helper ? -1 : PVAE_NET_MH,
                         phase == PVAE_PHASE_WORLD ? PVAE_NET_WM : PVAE_NET_MD,
                         phase == PVAE_PHASE_WORLD ? -1 : (learned_prior ? PVAE_NET_PR : PVAE_NET_TE),
                         phase == PVAE_PHASE_WORLD || !learned_prior ? -1 : PVAE_NET_TE};      // backward order
    c->bucket_bytes_now = auto_bucket_bytes(c, phase);
    hipStream_t cs = (c->bucket_bytes_now > 0 && c->comm_stream) ? c->comm_stream : st;
    int n_events = 0;
    auto join = [&]() -> int {                 // later work on the caller's stream sees the updated parameters
        if (cs == st) return 0;
        HIP_TRY(hipEventRecord(c->comm_done, cs));
        HIP_TRY(hipStreamWaitEvent(st, c->comm_done, 0));
        return 0;
    };
    if (rows == 0) {
        // empty shard of a ragged last global batch: contribute zeros, apply the same update
        for (int n : nets) {
            if (n < 0) continue;
            const NetLayout& N = c->L.net[n];
            HIP_TRY(hipMemsetAsync(c->grads + N.off, 0, (size_t)N.count * sizeof(float), st));
            for (const Bucket& b : exchange_buckets(c, n))
                if ((rc = exchange_bucket(c, n, b, sp, st, cs, n_events))) return rc;
        }
        return join();
    }
    // gather prefetch as in pvae_train_step_prefetch: this rank's next shard rides in the last launch
    const bool direct = enter_direct(c, phase, first_window, rows, sp, false);
    const bool can = !direct && c->W.L == 1 && c->pair_launch && loss_out != nullptr && c->states != nullptr;
    if (direct) {
        // (first layers gather their rows themselves: no staging launch; the last launch pre-touches the next shard's rows)
        if (loss_out) plan_touch(c, next_first, next_rows);
    } else if (can && c->pf.valid && c->pf.first == first_window && c->pf.rows == rows && c->pf.states == c->states) {
        flip_stage_panels(c);
        c->staged_rows = rows;
        c->staged_rows_f = rows;
    } else if ((rc = pvae_gather(c, first_window, rows, stream))) {
        return rc;
    }
    c->pf.valid = false;
    c->next_stage.rows_pad = 0;
    c->next_carried = false;
    if (can && next_rows > 0 && next_rows <= c->L.cfg.max_batch && next_first >= 0 &&
        next_first + next_rows <= c->n_windows)
        c->next_stage = stage_args(c, next_first, nullptr, nullptr, next_rows, true, 1, true);
    if ((rc = check_step(c, phase, rows, sp, true, false))) { if (direct) leave_direct(c); return rc; }
    StepShape S;
    if ((rc = step_shape(c, phase, rows, sp, loss_out, true, S))) return rc;
    if ((rc = run_forward(c, phase, rows, sp, eps, true, S, st))) return rc;
    Plan plan;
    plan_backward(c, phase, rows, sp, true, false, S, st, plan);
    // A stack's slices become final last layer first.  Each time the finished region reaches down
    // to the start of the next exchange bucket, that bucket goes to the exchange stream (reduce over
    // the ranks, then Adam on it) while this stream keeps launching the rest of the backward pass;
    // the parameters a bucket's Adam rewrites are not read again in this step (the fused path
    // rewrites them in the same launches).  The caller's stream rejoins at the end.
    std::vector<Bucket> bk[PVAE_NUM_NETS];
    size_t next_bk[PVAE_NUM_NETS] = {};
    int64_t low[PVAE_NUM_NETS];
    for (int n : nets)
        if (n >= 0) { bk[n] = exchange_buckets(c, n); low[n] = c->L.net[n].off + c->L.net[n].count; }
    for (Stage& s : plan) {
        if ((rc = s.run())) break;
        if (s.ready_cnt <= 0 || s.net < 0) continue;
        const int n = s.net;
        if (s.ready_off + s.ready_cnt != low[n]) {
            rc = fail(-2, "backward plan finished [%lld, +%lld) of stack %d out of order", (long long)s.ready_off,
                      (long long)s.ready_cnt, n);
            break;
        }
        low[n] = s.ready_off;
        while (!rc && next_bk[n] < bk[n].size() && bk[n][next_bk[n]].off >= low[n])
            rc = exchange_bucket(c, n, bk[n][next_bk[n]++], sp, st, cs, n_events);
        if (rc) break;
    }
    if (!rc)
        for (int n : nets)
            if (n >= 0 && next_bk[n] != bk[n].size()) rc = fail(-2, "stack %d left the backward pass unfinished", n);
    const int jrc = join();
    if (!rc) rc = jrc;
    if (!rc && c->next_carried) {
        c->pf.valid = true; c->pf.first = next_first; c->pf.rows = next_rows; c->pf.states = c->states;
    }
    c->next_stage.rows_pad = 0;
    c->next_carried = false;
    if (direct) leave_direct(c);
    return rc;
}

int pvae_train_step(pvae_ctx* c, int phase, int64_t first_window, int32_t rows, const pvae_step_params* sp,
                    const float* eps, float* loss_out, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (!c->states) return fail(-2, "dataset not bound");
    if (check_ready(c, true) == 0 && enter_direct(c, phase, first_window, rows, sp, true)) {
        const int rc = pvae_forward_backward(c, phase, rows, sp, eps, loss_out, PVAE_FLAG_FUSED_ADAM, stream);
        leave_direct(c);
        return rc;
    }
    int rc = pvae_gather(c, first_window, rows, stream);
    if (rc) return rc;
    return pvae_forward_backward(c, phase, rows, sp, eps, loss_out, PVAE_FLAG_FUSED_ADAM, stream);
}

// swap the roles of the two sets of staging panels
static void flip_stage_panels(pvae_ctx* c) {
    for (int n = 0; n < PVAE_NUM_NETS; ++n) std::swap(c->W.net[n].in, c->W.alt_in[n]);
    std::swap(c->W.s2, c->W.alt_s2);
    std::swap(c->W.act_t, c->W.alt_act_t);
}

int pvae_train_step_prefetch(pvae_ctx* c, int phase, int64_t first_window, int32_t rows, const pvae_step_params* sp,
                             const float* eps, float* loss_out, int64_t next_first, int32_t next_rows, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (!c->states) return fail(-2, "dataset not bound");
    int rc;
    if (check_ready(c, true) == 0 && enter_direct(c, phase, first_window, rows, sp, true)) {
        // (first layers gather their rows themselves: no staging launch; the last launch pre-touches the next minibatch's rows)
        if (loss_out) plan_touch(c, next_first, next_rows);
        rc = pvae_forward_backward(c, phase, rows, sp, eps, loss_out, PVAE_FLAG_FUSED_ADAM, stream);
        leave_direct(c);
        return rc;
    }
    const bool can = c->W.L == 1 && c->pair_launch && loss_out != nullptr;   // the carrier is the folding launch
    if (can && c->pf.valid && c->pf.first == first_window && c->pf.rows == rows && c->pf.states == c->states) {
        flip_stage_panels(c);                   // this minibatch is already staged
        c->staged_rows = rows;
        c->staged_rows_f = rows;
    } else if ((rc = pvae_gather(c, first_window, rows, stream))) {
        return rc;
    }
    c->pf.valid = false;
    c->next_stage.rows_pad = 0;
    c->next_carried = false;
    if (can && next_rows > 0 && next_rows <= c->L.cfg.max_batch && next_first >= 0 &&
        next_first + next_rows <= c->n_windows)
        c->next_stage = stage_args(c, next_first, nullptr, nullptr, next_rows, true, 1, true);
    rc = pvae_forward_backward(c, phase, rows, sp, eps, loss_out, PVAE_FLAG_FUSED_ADAM, stream);
    if (!rc && c->next_carried) {
        c->pf.valid = true; c->pf.first = next_first; c->pf.rows = next_rows; c->pf.states = c->states;
    }
    c->next_stage.rows_pad = 0;
    c->next_carried = false;
    return rc;
}

int pvae_read_tensor(pvae_ctx* c, int what, float* dst, int32_t rows, void* stream) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!dst || rows < 1 || rows > c->W.Bp) return fail(-1, "bad dst/rows");
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    const int t = what >> 3;                   // time step (lookahead > 1), 0 otherwise
    what &= 7;
    if (t < 0 || t >= c->W.L) return fail(-1, "time step %d outside [0, %d)", t, c->W.L);
    const int rows_pad = pad32(c->staged_rows > 0 ? c->staged_rows : rows);
    int64_t blk = (int64_t)t * rows_pad;
    const float* src; int ld, col0, nc;
    const NetWork& wte = c->W.net[PVAE_NET_TE];
    switch (what) {
        case 0: src = c->ws + wte.act.back(); ld = c->L.net[PVAE_NET_TE].layers.back().n_out_pad; col0 = 0; nc = Z; break;
        case 1:
            if (c->L.cfg.prior_kind >= PVAE_PRIOR_HYPERSPHERE) return fail(-1, "this encoder has no logvar");
            src = c->ws + wte.act.back(); ld = c->L.net[PVAE_NET_TE].layers.back().n_out_pad; col0 = Z; nc = Z; break;
        case 2: src = c->ws + z_panel(c); ld = c->L.net[PVAE_NET_MD].layers[0].ld; col0 = Db; nc = Z; break;
        case 3: src = c->ws + c->W.net[PVAE_NET_MD].act.back(); ld = c->L.net[PVAE_NET_MD].layers.back().n_out_pad; col0 = 0; nc = Da; break;
        case 4: src = c->ws + c->W.net[PVAE_NET_WM].act.back(); ld = c->L.net[PVAE_NET_WM].layers.back().n_out_pad; col0 = 0; nc = Db;
                if (c->W.L > 1) blk += (int64_t)c->W.L * rows_pad;      // the predicted-action invocation
                break;
        case 5: src = c->ws + c->W.eps; ld = Z; col0 = 0; nc = Z; break;
        case 6:
            if (c->L.net[PVAE_NET_PR].layers.empty()) return fail(-1, "no learned prior in this configuration");
            src = c->ws + c->W.net[PVAE_NET_PR].act.back(); ld = c->L.net[PVAE_NET_PR].layers.back().n_out_pad; col0 = 0; nc = Z;
            break;
        default: return fail(-1, "unknown tensor id %d", what);
    }
    src += blk * ld;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(32), dim3(256), 0, (hipStream_t)stream, src, ld, col0, dst, nc, 0, rows, nc);
    HIP_TRY(hipGetLastError());
    return 0;
}

// pvae_infer / pvae_infer_logits: the action lands in a_hat[r * ld_a + 0 .. Da) and, when `log_std` is given, the
// decoder's log-std vector behind it (AppendLogStd rmt:160-206: logits = [a_hat | log_std]).
// option "rollout_fused" = 0: rollout calls of <= 4 rows go through the staged path (A/B)
static bool rollout_fused() { return g_rollout_fused; }
static int infer_impl(pvae_ctx* c, const float* obs, int32_t rows, const float* eps, int noise, uint64_t rng_seed,
                      uint64_t rng_offset, float* a_hat, int ld_a, const float* log_std, float* s2_hat, float* z_out,
                      void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!obs || !a_hat) return fail(-1, "obs / a_hat is null");
    if (ld_a < c->L.cfg.dim_action * (log_std ? 2 : 1)) return fail(-1, "row stride %d of the action buffer is too small", ld_a);
    hipStream_t st = (hipStream_t)stream;
    if (rows < 1 || rows > c->L.cfg.max_batch) return fail(-1, "rows %d outside [1, %d]", rows, c->L.cfg.max_batch);
    const bool helper = !c->L.net[PVAE_NET_MH].layers.empty();          // (its term joins between decoder and world model: staged path)
    const bool fused_rollout = rollout_fused() && !helper;
    if (rows <= 4 && fused_rollout) {
        // latency path of the control loop (rmt:742-771 at B = 1): no staging / sampler / copy launches, the
        // input panels of a staged training minibatch are not touched
        const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
        float* w = c->ws;
        // (staged_rows / staged_rows_f stay as they are: a staged training minibatch remains valid, and
        //  forward_net picks its kernels by staged_rows_f)
        auto run_net = [&](int n, RolloutIn first, float* out2, int ld2, int n2, const float* ls) -> int {
            const NetLayout& N = c->L.net[n];
            RolloutIn in = first;
            for (const Layer& l : N.layers) {
                float* out = w + c->W.net[n].act[l.index];
                const dim3 grid(l.n_out_pad / 4), block(256);
                const size_t shm = (size_t)(rows <= 1 ? 1 : rows == 2 ? 2 : 4) * l.ld * sizeof(float);
                float* o2 = l.last ? out2 : nullptr;
                const int ps = g_prof.begin(0, 2.0 * rows * l.n_in * l.n_out, st);
#define PVAE_ROLL(R)                                                                                                  \
    hipLaunchKernelGGL((gemv_rollout_kernel<R>), grid, block, shm, st, in, (int)rows, c->params + l.w_off, l.ld,      \
                       c->params + l.b_off, out, l.n_out_pad, l.ld, l.act, o2, ld2, n2, l.n_out, l.last ? ls : nullptr)
                if (rows == 1) PVAE_ROLL(1);
                else if (rows == 2) PVAE_ROLL(2);
                else PVAE_ROLL(4);
#undef PVAE_ROLL
                g_prof.end(ps, st);
                HIP_TRY(hipGetLastError());
                memset(&in, 0, sizeof(in));
                in.kind = 0; in.a = out; in.lda = l.n_out_pad; in.Ka = l.n_out_pad;
            }
            return 0;
        };
        RolloutIn te;
        memset(&te, 0, sizeof(te));
        te.kind = 1; te.a = obs; te.lda = 2 * Db; te.Ka = 2 * Db;
        te.keep = w + c->W.obs_keep;       // what a deferred read of this forward (mu / logvar / prediction / value) re-uses
        if ((rc = run_net(PVAE_NET_TE, te, nullptr, 0, 0, nullptr))) return rc;
        RolloutIn md;
        memset(&md, 0, sizeof(md));
        md.kind = c->L.cfg.prior_kind == PVAE_PRIOR_NONE ? 4 : c->L.cfg.prior_kind == PVAE_PRIOR_HYPERSPHERE ? 5 : 2;
        md.a = obs; md.lda = 2 * Db; md.Ka = Db;
        md.b = w + c->W.net[PVAE_NET_TE].act.back(); md.ldb = c->L.net[PVAE_NET_TE].layers.back().n_out_pad; md.Kb = Z;
        md.eps = eps; md.noise = noise ? 1 : 0; md.seed = rng_seed; md.offset = rng_offset;
        md.z_out = z_out; md.eps_used = w + c->W.eps;
        if ((rc = run_net(PVAE_NET_MD, md, a_hat, ld_a, Da, log_std))) return rc;
        if (s2_hat) {
            RolloutIn wm;
            memset(&wm, 0, sizeof(wm));
            wm.kind = 3; wm.a = obs; wm.lda = 2 * Db; wm.Ka = Db;
            wm.b = w + c->W.net[PVAE_NET_MD].act.back(); wm.ldb = c->L.net[PVAE_NET_MD].layers.back().n_out_pad; wm.Kb = Da;
            if ((rc = run_net(PVAE_NET_WM, wm, s2_hat, Db, Db, nullptr))) return rc;
        }
        return 0;
    }
    if ((rc = stage(c, 0, obs, nullptr, rows, false, st, 1))) return rc;
    c->staged_rows = 0;      // not a training batch
    const int rows_pad = pad32(rows);
    const int Db = c->L.cfg.dim_body, Da = c->L.cfg.dim_action, Z = c->L.cfg.latent;
    float* w = c->ws;
    const NetLayout& TE = c->L.net[PVAE_NET_TE];
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    const NetLayout& WM = c->L.net[PVAE_NET_WM];
    if ((rc = forward_net(c, PVAE_NET_TE, rows_pad, st))) return rc;
    // (the learned prior mean plays no part in the action: rmt:801-809 only records it)
    if ((rc = launch_sampler(c, w + c->W.net[PVAE_NET_TE].act.back(), TE.layers.back().n_out_pad, eps, w + c->W.eps,
                             w + c->W.net[PVAE_NET_MD].in, MD.layers[0].ld, rows, rows_pad, noise ? 1 : 0,
                             (unsigned long long)rng_seed, (unsigned long long)rng_offset, (float*)nullptr, z_out,
                             (const float*)nullptr, 0, st)))                 // z also lands in the caller's buffer
        return rc;
    (void)Z;
    // The decoder's output layer can write a second copy of a_hat: into the world model's input
    // panel when the prediction is wanted, else straight into the caller's buffer (row counts the
    // GEMV kernel covers exactly -- the control loop's B = 1 -- so no padded row is written).
    const bool direct = !helper && !s2_hat && (rows == 1 || rows == 2 || rows == 4);
    FwdTail md_tail;
    if (direct) {
        md_tail.out2 = a_hat; md_tail.ld2 = ld_a; md_tail.off2 = 0; md_tail.n2 = Da;
    } else {
        md_tail.out2 = w + c->W.net[PVAE_NET_WM].in; md_tail.ld2 = WM.layers[0].ld; md_tail.off2 = Db; md_tail.n2 = Da;
    }
    if ((rc = forward_net(c, PVAE_NET_MD, rows_pad, st, md_tail))) return rc;
    if (helper) {                              // rmt:833-835
        const NetLayout& MH = c->L.net[PVAE_NET_MH];
        if ((rc = forward_net(c, PVAE_NET_MH, rows_pad, st))) return rc;
        const int grid = (rows * Da + 255) / 256 < 256 ? (rows * Da + 255) / 256 : 256;
        hipLaunchKernelGGL(helper_add_kernel, dim3(grid), dim3(256), 0, st, w + c->W.net[PVAE_NET_MD].act.back(),
                           MD.layers.back().n_out_pad, w + c->W.net[PVAE_NET_MH].act.back(), MH.layers.back().n_out_pad,
                           w + c->W.net[PVAE_NET_WM].in, WM.layers[0].ld, Db, rows, Da, c->L.cfg.mh_range);
        HIP_TRY(hipGetLastError());
    }
    if (!direct) {
        hipLaunchKernelGGL(copy_cols_kernel, dim3(32), dim3(256), 0, st, w + c->W.net[PVAE_NET_MD].act.back(),
                           MD.layers.back().n_out_pad, 0, a_hat, ld_a, 0, rows, Da);
        HIP_TRY(hipGetLastError());
    }
    if (log_std) {
        hipLaunchKernelGGL(append_logstd_kernel, dim3(8), dim3(256), 0, st, a_hat, ld_a, Da, rows, log_std);
        HIP_TRY(hipGetLastError());
    }
    if (s2_hat) {
        if ((rc = forward_net(c, PVAE_NET_WM, rows_pad, st))) return rc;
        hipLaunchKernelGGL(copy_cols_kernel, dim3(32), dim3(256), 0, st, w + c->W.net[PVAE_NET_WM].act.back(),
                           WM.layers.back().n_out_pad, 0, s2_hat, Db, 0, rows, Db);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

}   // extern "C"

extern "C" {
int pvae_rollout_is_fused(void) { return rollout_fused() ? 1 : 0; }

int pvae_infer(pvae_ctx* c, const float* obs, int32_t rows, const float* eps, int noise, uint64_t rng_seed,
               uint64_t rng_offset, float* a_hat, float* s2_hat, float* z_out, void* stream) {
    return infer_impl(c, obs, rows, eps, noise, rng_seed, rng_offset, a_hat, c ? c->L.cfg.dim_action : 0, nullptr, s2_hat,
                      z_out, stream);
}

int pvae_infer_logits(pvae_ctx* c, const float* obs, int32_t rows, const float* eps, int noise, uint64_t rng_seed,
                      uint64_t rng_offset, float* logits, int32_t ld_logits, const float* log_std, float* s2_hat,
                      float* z_out, void* stream) {
    return infer_impl(c, obs, rows, eps, noise, rng_seed, rng_offset, logits, ld_logits, log_std, s2_hat, z_out, stream);
}

int pvae_mlp_forward(const float* x, int32_t rows, int32_t ldx, int32_t n_layers, const float* const* W,
                     const float* const* bias, const int32_t* n_in, const int32_t* n_out, const int32_t* ldw,
                     int32_t act_kind, const int32_t* layer_acts, float* scratch, float* out, int32_t ld_out,
                     void* stream) {
    if (!x || !W || !n_in || !n_out || !ldw || !out) return fail(-1, "null argument");
    if (rows < 1 || n_layers < 1 || n_layers > 16) return fail(-1, "rows %d / layers %d out of range", rows, n_layers);
    const int out_code = (act_kind >> 8) & 0xff;                 // 1 + PVAE_ACT_* of the OUTPUT layer (0: linear)
    act_kind &= 0xff;
    if (act_kind < 0 || act_kind > PVAE_ACT_ELU) return fail(-1, "unknown act_kind %d", act_kind);
    if (out_code > PVAE_ACT_ELU + 1) return fail(-1, "unknown output activation %d", out_code - 1);
    for (int i = 0; layer_acts && i + 1 < n_layers; ++i)
        if (layer_acts[i] < 0 || layer_acts[i] > PVAE_ACT_LINEAR) return fail(-1, "unknown activation %d of layer %d", layer_acts[i], i);
    int wmax = 0;
    for (int i = 0; i + 1 < n_layers; ++i) wmax = n_out[i] > wmax ? n_out[i] : wmax;
    if (n_layers > 1 && !scratch) return fail(-1, "scratch (2 * rows * widest hidden layer floats) is null");
    hipStream_t st = (hipStream_t)stream;
    const float* in = x;
    int ldi = ldx;
    for (int i = 0; i < n_layers; ++i) {
        if (n_in[i] < 1 || n_out[i] < 1 || ldw[i] < n_in[i] || !W[i]) return fail(-1, "bad layer %d", i);
        if (i > 0 && n_in[i] != n_out[i - 1]) return fail(-1, "layer %d reads %d features, layer %d emits %d", i, n_in[i], i - 1, n_out[i - 1]);
        const bool last = i == n_layers - 1;
        float* o = last ? out : scratch + (size_t)(i & 1) * rows * wmax;
        const int ldo = last ? ld_out : wmax;
        const dim3 grid((n_out[i] + 3) / 4, (rows + 3) / 4);
        hipLaunchKernelGGL((gemv_dense_kernel<4>), grid, dim3(256), 0, st, in, ldi, (int)rows, W[i], (int)ldw[i],
                           bias ? bias[i] : (const float*)nullptr, (int)n_in[i], (int)n_out[i],
                           last ? out_code : (layer_acts ? (layer_acts[i] == PVAE_ACT_LINEAR ? 0 : layer_acts[i] + 1) : act_kind + 1), o, ldo);
        HIP_TRY(hipGetLastError());
        in = o;
        ldi = ldo;
    }
    return 0;
}

int pvae_net_forward(pvae_ctx* c, int net, const float* in, int32_t rows, float* out, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (net < 0 || net >= PVAE_NUM_NETS) return fail(-1, "bad net id %d", net);
    if (!in || !out) return fail(-1, "in / out is null");
    if (rows < 1 || rows > c->L.cfg.max_batch) return fail(-1, "rows %d outside [1, %d]", rows, c->L.cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const NetLayout& N = c->L.net[net];
    const int rows_pad = pad32(rows), ld = N.layers[0].ld;
    int grid = (rows_pad * ld + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(pad_copy_kernel, dim3(grid), dim3(256), 0, st, in, N.n_in, rows, c->ws + c->W.net[net].in, ld,
                       rows_pad, N.layers[0].col0, N.layers[0].n_in);
    HIP_TRY(hipGetLastError());
    c->staged_rows = 0;     // the training panels are no longer a coherent batch
    c->staged_rows_f = rows;
    if ((rc = forward_net(c, net, rows_pad, st))) return rc;
    hipLaunchKernelGGL(copy_cols_kernel, dim3(32), dim3(256), 0, st, c->ws + c->W.net[net].act.back(),
                       N.layers.back().n_out_pad, 0, out, N.n_out, 0, rows, N.n_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int pvae_reparam(pvae_ctx* c, const float* mu_logvar, int32_t rows, const float* eps, int noise, uint64_t rng_seed,
                 uint64_t rng_offset, float* z_out, void* stream) {
    int rc = check_ready(c, false);
    if (rc) return rc;
    if (!mu_logvar || !z_out) return fail(-1, "mu_logvar / z_out is null");
    if (rows < 1 || rows > c->L.cfg.max_batch) return fail(-1, "rows %d outside [1, %d]", rows, c->L.cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    const int Z = c->L.cfg.latent;
    const int ld_md = c->L.net[PVAE_NET_MD].layers[0].ld;
    c->staged_rows = 0;
    const int ldte = c->L.cfg.prior_kind >= PVAE_PRIOR_HYPERSPHERE ? Z : 2 * Z;       // dense [rows][n_out of the encoder]
    // pad rows are not touched: `rows` doubles as rows_pad (the sphere kernel rounds its grid up itself)
    return launch_sampler(c, mu_logvar, ldte, eps, c->ws + c->W.eps, c->ws + c->W.net[PVAE_NET_MD].in, ld_md, rows, rows,
                          noise ? 1 : 0, (unsigned long long)rng_seed, (unsigned long long)rng_offset, (float*)nullptr,
                          z_out, (const float*)nullptr, 0, st);
}

}  // extern "C"
