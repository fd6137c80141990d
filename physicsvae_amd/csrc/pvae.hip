// pvae.hip -- libpvae_gfx950.so: C-ABI (include/pvae.h) + glue kernels around the MFMA
// tile kernel of pvae_gemm.h.  gfx950 (MI355X) only; built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared pvae.hip -o libpvae_gfx950.so
//
// Reference lines restated by each kernel are cited at the kernel (tpv / tm / rmt as in
// include/pvae.h).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "pvae_gemm.h"
#include "pvae_layout.h"

using namespace pvae;

// ---------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIP_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t e_ = (expr);                                                          \
        if (e_ != hipSuccess) return fail(-10, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

// ---------------------------------------------------------------------------------------
// optional per-launch timing (HIP events on the launch stream)
// ---------------------------------------------------------------------------------------
struct Profiler {
    bool on = false;
    static constexpr int kMax = 8192;
    hipEvent_t ev[kMax][2];
    int cat[kMax];
    double flops[kMax];
    int n = 0, created = 0;
    // begin() arms the slot's event pair; the launch wrapper (PVAE_LAUNCH, pvae_gemm.h) hands it to
    // hipExtLaunchKernelGGL, so the pair brackets the kernel itself and not the launch seam.  Every
    // profiled range holds exactly one launch; a range that launched nothing is dropped.
    int begin(int category, double fl, hipStream_t) {
        if (!on || n >= kMax) return -1;
        if (n >= created) {
            if (hipEventCreate(&ev[n][0]) != hipSuccess || hipEventCreate(&ev[n][1]) != hipSuccess) return -1;
            created = n + 1;
        }
        cat[n] = category;
        flops[n] = fl;
        g_kernel_ev[0] = ev[n][0];
        g_kernel_ev[1] = ev[n][1];
        return n;
    }
    void end(int slot, hipStream_t) {
        if (slot < 0) return;
        if (!g_kernel_ev[0]) n = slot + 1;          // consumed by a launch
        g_kernel_ev[0] = g_kernel_ev[1] = nullptr;
    }
    // a range that is not one of our launches (the RCCL collective): events recorded on the stream
    // around the call; `fl` carries the payload bytes instead of flops
    int begin_range(int category, double fl, hipStream_t st) {
        if (!on || n >= kMax) return -1;
        if (n >= created) {
            if (hipEventCreate(&ev[n][0]) != hipSuccess || hipEventCreate(&ev[n][1]) != hipSuccess) return -1;
            created = n + 1;
        }
        cat[n] = category;
        flops[n] = fl;
        if (hipEventRecord(ev[n][0], st) != hipSuccess) return -1;
        return n;
    }
    void end_range(int slot, hipStream_t st) {
        if (slot < 0) return;
        if (hipEventRecord(ev[slot][1], st) == hipSuccess) n = slot + 1;
    }
};
static Profiler g_prof;

// ---------------------------------------------------------------------------------------
// RCCL, resolved at run time.  PyTorch-ROCm ships its own librccl.so.1 and has it loaded; the
// library binds to THAT instance (RTLD_NOLOAD first) instead of linking a second copy, and
// falls back to the system one (/opt/rocm/lib) when used without torch.  Only the five entry
// points of the data-parallel exchange are needed; prototypes as in rccl/rccl.h (2.2x).
// ---------------------------------------------------------------------------------------
struct RcclId { char internal[128]; };                 // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(RcclId*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId, int) = nullptr;          // id is passed BY VALUE
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;   // optional
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;            // optional
    int (*CommDestroy)(void*) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;
    int (*CommUserRank)(void*, int*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok() const { return h && GetUniqueId && CommInitRank && AllReduce && CommDestroy && GetErrorString; }
};
static Rccl g_rccl;
enum { kNcclSum = 0, kNcclFloat32 = 7 };

static int rccl_load() {
    if (g_rccl.ok()) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;            // the instance torch already mapped
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) return fail(-20, "RCCL not found (librccl.so.1): %s", dlerror());
    g_rccl.h = h;
    g_rccl.GetUniqueId = (int (*)(RcclId*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, RcclId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclReduceScatter");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.CommCount = (int (*)(void*, int*))dlsym(h, "ncclCommCount");           // optional (pvae_comm_info)
    g_rccl.CommUserRank = (int (*)(void*, int*))dlsym(h, "ncclCommUserRank");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.ok()) {
        g_rccl = Rccl();
        return fail(-20, "RCCL library lacks an expected symbol");
    }
    return 0;
}
#define RCCL_TRY(expr)                                                                        \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != 0) return fail(-21, "%s: %s", #expr, g_rccl.GetErrorString(r_));            \
    } while (0)

struct pvae_ctx {
    void* comm = nullptr;        // ncclComm_t of the data-parallel group (pvae_comm_init)
    int comm_rank = 0, comm_world = 1;
    // overlapped gradient exchange (pvae_dp_train_step): the buckets of a stack are reduced and
    // applied on comm_stream while the compute stream keeps producing the next ones
    hipStream_t comm_stream = nullptr;
    static constexpr int kMaxBuckets = 64;
    hipEvent_t bucket_ready[kMaxBuckets] = {};
    hipEvent_t comm_done = nullptr;
    int exchange_mode = 0;             // PVAE_EXCHANGE_*: all-reduce + replicated Adam, or sharded (ZeRO-1 shaped)
    int64_t bucket_bytes = -1;         // > 0: bucketed + overlapped; 0: one bucket per stack, in line on the compute
                                       // stream; -1 (default): chosen per step by auto_bucket_bytes()
    int64_t bucket_bytes_now = 0;      // what the step in flight uses (exchange_buckets / dp_train_step)
    int comm_test_delay_us = 0;        // tests: a spin kernel in front of every reduction
    Layout L;
    Workspace W;
    float* params = nullptr;
    float* grads = nullptr;
    float* m = nullptr;
    float* v = nullptr;
    float* ws = nullptr;
    const float* states = nullptr;
    const float* next_states = nullptr;      // pvae_bind_dataset_next (null: next row of `states`)
    const float* actions = nullptr;
    const int32_t* window_row = nullptr;
    int64_t n_rows = 0, n_windows = 0;
    int staged_rows = 0;
    double staged_rows_f = 0;    // rows of the batch being processed (for the profiler's flop count)
    // First layers on the demonstration set where it lies (XSrc in pvae_gemm.h): the training-step entry
    // points (pvae_train_step, _prefetch, pvae_dp_train_step) stage nothing when `direct_ok` holds -- the first layer of
    // every stack gathers its rows of `states` / `actions` itself, the two targets are read from there by the loss
    // epilogues.  pvae_gather / pvae_set_batch + pvae_forward_backward keep the panel path (inspection, explicit batches,
    // lookahead > 1, evaluation, the other priors).  OPT-IN (pvae_set_direct(ctx, 1)): bit-identical to the staged step,
    // but at 256 rows the staged step is the faster one -- its gather rides in the previous step's last launch for free,
    // while a gathered first layer waits for its operand descriptor (kernel arguments that cannot be preloaded) before its
    // first tile fetch: joint 252.3 vs 241.5 us, world 92.1 vs 87.3 (docs/experiments.md, round 5).
    bool direct = false;
    bool data_slack = false;     // both dataset arrays are readable 16 bytes past their last row (checked at bind time)
    struct { bool on = false; RowMap rm{}; } dx;                     // the step in flight: batch row -> row of the set
    TouchRuns next_touch{};                                          // rows of the NEXT minibatch for the last launch to pre-touch
    std::vector<int32_t> window_row_host;                            // copied at bind time: the host finds the episode jumps
    bool pair_launch = true;     // PVAE_PAIR=0 launches every contraction on its own (A/B)
    // gather prefetch (pvae_train_step_prefetch): what the alternate staging panels hold, and the
    // staging job the current step's last launch should carry
    struct { bool valid = false; int64_t first = 0; int rows = 0; const float* states = nullptr; } pf;
    StageArgs next_stage;        // rows_pad > 0: pending for the last launch of this step
    bool next_carried = false;   // set by the launch that took it
    bool seed_pads_clean = false;  // pad columns of the seed panels zeroed (see plan_backward)
    // deferred Adam (AdamSeg, pvae_gemm.h): the layer whose gradient the last launch stored; the next
    // weight-gradient launch of the step updates it with extra workgroups (PVAE_DEFER_ADAM=0: off)
    AdamSeg pending_adam;          // the most recent one
    AdamSeg held_adam;             // a big one that a narrow launch passed on to the next wide launch (take_pending)
    bool defer_adam = true;
    bool same_layer_pairs = true;  // PVAE_SAME_LAYER=0: wgrad_i rides with dgrad_{i-1} as before (A/B)
    bool p2p_selftest_flags_only = false;   // option: the attach-time self-test skips the cached-arena part
    int server_mailbox = 0;                 // option: where the rollout server's request block lives (0 auto, 1 host, 2 device)
    bool fold_sampler = true;      // the sampler runs as the prologue of the decoder's first-layer launch (PVAE_FOLD_SAMPLER=0: its own launch)
                                   // (ProSampler).  Off by default: one launch less, but the step is not shorter -- the kernel
                                   // trace shows 6.4-7.0 us for the merged launch against 4.4 + 4.6, and the un-profiled
                                   // step 254.8 vs 254.5 us (profiles/r03_ab_fold_sampler.txt, docs/experiments.md)
    // peer-mapped exchange (PVAE_EXCHANGE_P2P): every rank's gradient arena, parameter arena and flag block,
    // mapped into this process with hipIpcOpenMemHandle (index = rank; [rank] = the local pointers)
    struct P2p {
        bool open = false;
        int rank = 0, world = 0;
        unsigned* flags = nullptr;                       // own flag block (uncached device memory)
        float* grads[PVAE_P2P_MAX_RANKS] = {};
        float* params[PVAE_P2P_MAX_RANKS] = {};
        unsigned* peer_flags[PVAE_P2P_MAX_RANKS] = {};
        float* staging = nullptr;                        // own staging buffer of the push form (hipMalloc, arena-sized)
        float* peer_staging[PVAE_P2P_MAX_RANKS] = {};
        void* mapped[PVAE_P2P_MAX_RANKS][4] = {};        // what hipIpcOpenMemHandle returned (to close)
        unsigned epoch = 0;                              // exchanges issued so far (identical on every rank)
        float* self_buf = nullptr;                       // self-test scratch: saved regions + checksums (hipMalloc)
        unsigned selftests = 0;                          // self-tests run since the flags were zeroed (identical on every rank)
        long long timeout_ticks = 20ll * 100000000ll;    // 100 MHz wall clock
    } p2p;
    struct RolloutServer* server = nullptr;              // call-persistent rollout kernel (pvae_rollout_server_*)
    // every call that changes parameters through this library counts here and leaves its stream: the rollout server re-reads
    // its resident copy when the count moved (after that stream has drained)
    unsigned long long param_version = 0;
    hipStream_t param_stream = nullptr;                  // (NULL is a stream too: the default one)
    bool param_pending = false;                          // work that writes the parameters may still be queued on it
};
static inline void params_touched(pvae_ctx* c, hipStream_t st, bool queued = true) {
    ++c->param_version; c->param_stream = st; c->param_pending = queued;
}
static inline hipError_t params_settle(pvae_ctx* c) {
    if (!c->param_pending) return hipSuccess;
    c->param_pending = false;
    return hipStreamSynchronize(c->param_stream);
}
static void server_free(pvae_ctx* c);

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
        d0[(size_t)r * ld0 + c] = v;
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

// Philox4x32-10 (Salmon et al., SC'11) -> one standard normal via Box-Muller.
__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// One Philox call = four standard normals: the draws of columns 4g .. 4g + 3 of row `row` (counter = {offset, row,
// g}; two Box-Muller pairs from the four 32-bit outputs).  Hardware transcendentals (v_log_f32, v_sqrt_f32,
// v_sin_f32 / v_cos_f32, which take their argument in revolutions: cos(2 pi u) is ONE instruction): ~1 ulp, which a
// random draw does not notice, at a tenth of the instructions of logf / cosf -- the draws are formed inside a
// contraction launch by every workgroup that needs them (ProSampler below), so their cost is multiplied.
__device__ inline v4f philox_normal4(uint64_t seed, uint64_t offset, uint32_t row, uint32_t group) {
    uint32_t c[4] = {(uint32_t)offset, (uint32_t)(offset >> 32), row, group};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    v4f n;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;       // (0, 1)
        const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), log2 form
        n[2 * h] = r * __builtin_amdgcn_cosf(u2);
        n[2 * h + 1] = r * __builtin_amdgcn_sinf(u2);
    }
    return n;
}
__device__ inline float philox_normal(uint64_t seed, uint64_t offset, uint32_t row, uint32_t col) {
    return philox_normal4(seed, offset, row, col >> 2)[col & 3];
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

// dst[rows_pad][ld] = zero-padded copy of dense src[rows][n]
__global__ void __launch_bounds__(256)
pad_copy_kernel(const float* __restrict__ src, int n, int rows, float* __restrict__ dst, int ld, int rows_pad) {
    const int total = rows_pad * ld;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
        const int r = idx / ld, c = idx - r * ld;
        dst[idx] = (r < rows && c < n) ? src[(size_t)r * n + c] : 0.f;
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

// ---------------------------------------------------------------------------------------
// Direct all-pairs gradient exchange over peer-mapped arenas (PVAE_EXCHANGE_P2P; SURVEY.md section 8e: "direct
// reduce-scatter + all-gather across all 7 links").  ONE launch per bucket and rank:
//   1. workgroup 0 tells every peer "my gradient of this bucket is final" (the launches that produced it precede
//      this one in the stream): epoch -> peer's ready[me];
//   2. every workgroup waits until all peers have told it the same (ready[q] >= epoch, local uncached memory);
//   3. the rank owns slice `me` of the bucket: for each float4 of it, the N gradients are read straight from the
//      N arenas (system-scope loads, all N in flight together), summed IN RANK ORDER, Adam is applied with the
//      local moments, and the new parameters are written to the local arena AND pushed into every peer's;
//   4. the last workgroup to finish (ticket) fences, tells every peer "done" and waits for every peer's "done":
//      when the launch ends this rank's parameter arena is complete and its gradient arena may be overwritten.
// Epochs only grow and every rank issues the same sequence of exchanges, so one word per (kind, source rank) is
// enough and a peer that is one exchange ahead cannot be mistaken (>= comparisons).  Every wait is bounded: a
// peer that never signals raises the error word instead of hanging the GPU.
// Flag block (unsigned words): [0, 8) ready[src], [8, 16) done[src], 16 ticket, 17 waits that gave up.
// ---------------------------------------------------------------------------------------
//                              18 second ticket, [24, 32) pushed[src] (push form), [32, 40) self-test tokens,
//                              [64, 96) self-test payload (4 words per source rank).
constexpr int kP2pReady = 0, kP2pDone = 8, kP2pTicket = 16, kP2pErr = 17, kP2pTicket2 = 18, kP2pPushed = 24, kP2pSelf = 32,
              kP2pPayload = 64, kP2pFlagBytes = 4096;
struct P2pArgs {
    float* g[PVAE_P2P_MAX_RANKS];           // gradient arenas, bucket offset applied (g[me]: local)
    float* p[PVAE_P2P_MAX_RANKS];           // parameter arenas, bucket offset applied
    unsigned* f[PVAE_P2P_MAX_RANKS];        // flag blocks
    float* stage[PVAE_P2P_MAX_RANKS];       // staging buffers (push form): [N][slice] floats at each owner
    float* m; float* v;                     // local moments, bucket offset applied
    long long n4;                           // float4 elements in the bucket
    int me;
    unsigned epoch;
    long long timeout_ticks;
    AdamScalars s;
};
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ inline unsigned p2p_ld(const unsigned* q) { return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline void p2p_st(unsigned* q, unsigned x) { __hip_atomic_store(q, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ inline bool p2p_wait(const unsigned* flag, unsigned epoch, long long timeout, unsigned* err) {
    const long long t0 = wall_clock64();
    while ((int)(p2p_ld(flag) - epoch) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > timeout) { atomicAdd(err, 1u); return false; }
    }
    return true;
}
template <int N>
__global__ void __launch_bounds__(256) p2p_exchange_kernel(P2pArgs a) {
    unsigned* mine = a.f[a.me];
    const int tid = threadIdx.x, me = a.me;
    if (blockIdx.x == 0 && tid < N && tid != me) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");             // (system scope; the producing launches ended before this one began)
        p2p_st(a.f[tid] + kP2pReady + me, a.epoch);
    }
    // A wait that gives up ABORTS the exchange on this rank: no peer gradient that may be unfinished is summed, no
    // moment moves, nothing is pushed -- parameters and moments stay what they were before the launch, the error word
    // says so (pvae_p2p_status), and the "done" hand-shake below still runs so that the peers are not left waiting.
    __shared__ int abort_;
    if (tid == 0) abort_ = 0;
    __syncthreads();
    if (tid < N && tid != me) {
        if (!p2p_wait(mine + kP2pReady + tid, a.epoch, a.timeout_ticks, mine + kP2pErr)) abort_ = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    const long long S = (a.n4 + N - 1) / N, lo = me * S, hi = lo + S < a.n4 ? lo + S : a.n4;
    if (lo < hi && !abort_) {
        // buffer descriptors over this rank's slice of every arena: loads / stores with sc0 sc1 (system scope,
        // past this device's caches) that the compiler schedules and counts like any other memory operation
        __amdgpu_buffer_rsrc_t rg[N], rp[N];
        const unsigned bytes = (unsigned)((hi - lo) * 16);
#pragma unroll
        for (int q = 0; q < N; ++q) {
            rg[q] = __builtin_amdgcn_make_buffer_rsrc(a.g[q] + 4 * lo, 0, bytes, 0x00020000);
            rp[q] = __builtin_amdgcn_make_buffer_rsrc(a.p[q] + 4 * lo, 0, bytes, 0x00020000);
        }
        for (long long i = blockIdx.x * 256ll + tid; i < hi - lo; i += gridDim.x * 256ll) {
            const unsigned off = (unsigned)(i * 16);
            v4f g[N];
#pragma unroll
            for (int q = 0; q < N; ++q) g[q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rg[q], off, 0, 17));
            v4f pp = reinterpret_cast<const v4f*>(a.p[me])[lo + i];
            v4f mm = reinterpret_cast<const v4f*>(a.m)[lo + i];
            v4f vv = reinterpret_cast<const v4f*>(a.v)[lo + i];
            v4f sum = g[0];
#pragma unroll
            for (int q = 1; q < N; ++q) sum += g[q];                // rank order, whoever owns the slice
            adam_update4(sum, pp, mm, vv, a.s);
            store_stream(a.m + 4 * (lo + i), mm);
            store_stream(a.v + 4 * (lo + i), vv);
#pragma unroll
            for (int q = 0; q < N; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pp), rp[q], off, 0, 17);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned last;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        last = atomicAdd(mine + kP2pTicket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (tid == 0) mine[kP2pTicket] = 0;
    if (tid < N && tid != me) {
        p2p_st(a.f[tid] + kP2pDone + me, a.epoch);
        p2p_wait(mine + kP2pDone + tid, a.epoch, a.timeout_ticks, mine + kP2pErr);
    }
}

// The PUSH form of the same exchange (PVAE_EXCHANGE_P2P_PUSH): remote WRITES only.  Posted writes pipeline over a link
// where reads are round trips, so this is the form a fabric with write-favouring links wants; which of the two wins
// on xGMI is for the first multi-GPU run to say (bench.py's exchange_sweep times both).
//   1. every rank writes, for each peer q, ITS contribution to slice q into slot `me` of q's staging buffer;
//      the last workgroup to finish (ticket) fences and tells every peer "pushed";
//   2. every workgroup waits for all peers' "pushed", then the owner sums its slice in rank order -- its own gradient
//      from the arena, the others from its LOCAL staging (system-scope loads: remote agents wrote it) --, applies Adam
//      and pushes the new parameters into every peer's parameter arena;
//   3. last workgroup: "done" to every peer, wait for every peer's "done" (the staging may then be overwritten).
template <int N>
__global__ void __launch_bounds__(256) p2p_push_exchange_kernel(P2pArgs a) {
    unsigned* mine = a.f[a.me];
    const int tid = threadIdx.x, me = a.me;
    const long long S = (a.n4 + N - 1) / N, stride = gridDim.x * 256ll;
    __shared__ unsigned last;
    {   // 1. scatter-push
        __amdgpu_buffer_rsrc_t rs[N];
#pragma unroll
        for (int q = 0; q < N; ++q)
            rs[q] = __builtin_amdgcn_make_buffer_rsrc(a.stage[q] + (size_t)me * S * 4, 0, (unsigned)(S * 16), 0x00020000);
        for (long long i = blockIdx.x * 256ll + tid; i < S; i += stride) {
#pragma unroll
            for (int q = 0; q < N; ++q) {
                if (q == me || q * S + i >= a.n4) continue;
                const v4f g = reinterpret_cast<const v4f*>(a.g[me])[q * S + i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, g), rs[q], (unsigned)(i * 16), 0, 17);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            last = atomicAdd(mine + kP2pTicket2, 1u) == gridDim.x - 1;
        }
        __syncthreads();
        if (last) {
            if (tid == 0) mine[kP2pTicket2] = 0;
            if (tid < N && tid != me) p2p_st(a.f[tid] + kP2pPushed + me, a.epoch);
        }
    }
    __shared__ int abort_;              // (see p2p_exchange_kernel: a wait that gives up aborts this rank's update)
    if (tid == 0) abort_ = 0;
    __syncthreads();
    if (tid < N && tid != me) {       // 2. everything for my slice has arrived
        if (!p2p_wait(mine + kP2pPushed + tid, a.epoch, a.timeout_ticks, mine + kP2pErr)) abort_ = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __syncthreads();
    const long long lo = me * S, hi = lo + S < a.n4 ? lo + S : a.n4;
    if (lo < hi && !abort_) {
        const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(a.stage[me], 0, (unsigned)(N * S * 16), 0x00020000);
        __amdgpu_buffer_rsrc_t rp[N];
#pragma unroll
        for (int q = 0; q < N; ++q) rp[q] = __builtin_amdgcn_make_buffer_rsrc(a.p[q] + 4 * lo, 0, (unsigned)((hi - lo) * 16), 0x00020000);
        for (long long i = blockIdx.x * 256ll + tid; i < hi - lo; i += stride) {
            v4f g[N];
#pragma unroll
            for (int q = 0; q < N; ++q)
                g[q] = q == me ? reinterpret_cast<const v4f*>(a.g[me])[lo + i]
                               : __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rst, (unsigned)((q * S + i) * 16), 0, 17));
            v4f pp = reinterpret_cast<const v4f*>(a.p[me])[lo + i];
            v4f mm = reinterpret_cast<const v4f*>(a.m)[lo + i];
            v4f vv = reinterpret_cast<const v4f*>(a.v)[lo + i];
            v4f sum = g[0];
#pragma unroll
            for (int q = 1; q < N; ++q) sum += g[q];                // rank order
            adam_update4(sum, pp, mm, vv, a.s);
            store_stream(a.m + 4 * (lo + i), mm);
            store_stream(a.v + 4 * (lo + i), vv);
#pragma unroll
            for (int q = 0; q < N; ++q) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, pp), rp[q], (unsigned)(i * 16), 0, 17);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        last = atomicAdd(mine + kP2pTicket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    if (tid == 0) mine[kP2pTicket] = 0;
    if (tid < N && tid != me) {
        p2p_st(a.f[tid] + kP2pDone + me, a.epoch);
        p2p_wait(mine + kP2pDone + tid, a.epoch, a.timeout_ticks, mine + kP2pErr);
    }
}

// Self-test of the mappings, run once when the peers are opened: every rank writes a 4-word record into its slot of
// every peer's flag block (remote write), signals, waits for the peers' signals, checks the records that arrived in
// its own block (written by remote agents) and reads back, from every peer's block, the record it wrote there
// (remote read).  Anything wrong -- a mapping that does not reach the peer, a flag that never arrives -- raises the
// error word within `timeout_ticks` instead of surfacing as a hang in the first training step.
__global__ void p2p_selftest_kernel(P2pArgs a, int n, unsigned token) {
    unsigned* mine = a.f[a.me];
    const int q = threadIdx.x, me = a.me;
    if (q >= n || q == me) return;
    unsigned* theirs = a.f[q];
    for (int wd = 0; wd < 4; ++wd) p2p_st(theirs + kP2pPayload + me * 4 + wd, wd == 0 ? token : wd == 1 ? (unsigned)me : wd == 2 ? (unsigned)q : 0xC0FFEEu);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    p2p_st(theirs + kP2pSelf + me, token);
    p2p_wait(mine + kP2pSelf + q, token, a.timeout_ticks, mine + kP2pErr);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const bool got = p2p_ld(mine + kP2pPayload + q * 4) == token && p2p_ld(mine + kP2pPayload + q * 4 + 1) == (unsigned)q &&
                     p2p_ld(mine + kP2pPayload + q * 4 + 2) == (unsigned)me && p2p_ld(mine + kP2pPayload + q * 4 + 3) == 0xC0FFEEu;
    const bool back = p2p_ld(theirs + kP2pPayload + me * 4) == token && p2p_ld(theirs + kP2pPayload + me * 4 + 3) == 0xC0FFEEu;
    if (!got || !back) atomicAdd(mine + kP2pErr, 1u);
}


// ---- self-test of the CACHED arenas ---------------------------------------------------------
// The flag block above is uncached memory; the arenas the exchange really moves are plain hipMalloc (coarse-grained)
// memory that this device's L2s cache.  Peers overwrite this rank's parameters over the links while the lines may
// still sit in the local L2s from the last forward pass, and the next forward launch starts behind an agent-scope
// acquire only.  If a remote write left a stale line behind, every rank would train on old weights of the slices it
// does not own -- and the replicas would still be bit-identical.  So, once per set-up, the very access paths of the
// exchange are exercised on a TEST REGION of each buffer and every read-back is compared with what was written:
//   parameters  first kSelfFloats floats of the arena (saved first, restored at the end), one 128-byte line per source
//               rank: primed into the local L2s of all XCDs (LDS-DMA loads, the forward kernels' path, and plain
//               loads), overwritten by the peers with the exchange's own `buffer_store ... sc0 sc1`, re-read by a FRESH
//               dependent launch on every XCD through the same two load paths;
//   staging     the 256-float tail of the staging buffer: primed, overwritten by the peers, read in the SAME launch
//               behind the flag wait with the push form's system-scope loads, and again by the fresh launch;
//   gradients   first kSelfFloats floats of the arena (saved / restored): the owner writes pattern A with plain stores, the
//               peers read their line with the pull form's `buffer_load ... sc0 sc1`; the owner overwrites it with
//               pattern B and the peers read again -- a reader-side stale line would return A.
// Any mismatch or missing flag raises the error word; pvae_p2p_selftest then fails and the caller drops the form.
constexpr int kP2pPrimed = 96, kP2pWritten = 104, kP2pGradB = 112, kP2pFin = 120;       // flag words, [src rank]
constexpr int kSelfLine = 32, kSelfFloats = PVAE_P2P_MAX_RANKS * kSelfLine;             // 8 lines of 128 bytes
constexpr int kSelfGrid = 64;                                                           // 8 workgroups on every XCD
struct SelfArgs {
    P2pArgs a;              // g / p / f / stage: the test regions' base pointers (stage: the tail), me, timeout
    float* save;            // [2 * kSelfFloats]: what the parameter and gradient regions held
    unsigned* sink;         // [kSelfGrid] checksums (keeps the priming loads alive)
    int n;
    unsigned token;
};
__device__ inline float self_pat(unsigned token, int src, int dst, int j, int round) {
    return (float)(((token & 0xFFFFu) * 131u + (unsigned)src * 1021u + (unsigned)dst * 67u + (unsigned)round * 4099u) % 65521u) +
           (float)j * 0.0078125f;                                     // exactly representable, distinct per (src, dst, j, round)
}
// the two paths a forward launch reads parameters through: LDS-DMA (default cache policy) and a plain 16-byte load
__device__ inline v4f self_read_dma(const float* region, float* lds, int lane) {
    lds_dma16(region + 4 * lane, lds);                                // 64 lanes x 16 bytes = the 1 KB region
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    return *reinterpret_cast<const v4f*>(lds + 4 * lane);
}
__global__ void __launch_bounds__(64) p2p_self_prime_kernel(SelfArgs s) {
    __shared__ __attribute__((aligned(16))) float lds[kSelfFloats];
    const int lane = threadIdx.x, me = s.a.me;
    const v4f pd = self_read_dma(s.a.p[me], lds, lane);
    const v4f pl = *reinterpret_cast<const v4f*>(s.a.p[me] + 4 * lane);
    const v4f sl = *reinterpret_cast<const v4f*>(s.a.stage[me] + 4 * lane);
    const v4f gl = *reinterpret_cast<const v4f*>(s.a.g[me] + 4 * lane);
    if (blockIdx.x == 0) {
        *reinterpret_cast<v4f*>(s.save + 4 * lane) = pl;
        *reinterpret_cast<v4f*>(s.save + kSelfFloats + 4 * lane) = gl;
        v4f a;                                                        // gradient pattern A: line q is what peer q will read
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] = self_pat(s.token, me, (4 * lane + e) / kSelfLine, (4 * lane + e) % kSelfLine, 0);
        *reinterpret_cast<v4f*>(s.a.g[me] + 4 * lane) = a;            // plain store, as an ordinary producer would
    }
    const float c = pd[0] + pd[3] + pl[1] + sl[2] + gl[0];
    if (lane == 0) s.sink[blockIdx.x] = __float_as_uint(c);
}
// one wave: signal "primed", wait for the peers', write my lines into every peer's parameter and staging regions with the
// exchange's stores, read my line of every peer's gradient region (pattern A) with the exchange's loads, signal
// "written", wait for the peers', and check my staging region in this same launch (the push form's situation)
__global__ void __launch_bounds__(64) p2p_self_write_kernel(SelfArgs s) {
    unsigned* mine = s.a.f[s.a.me];
    const int lane = threadIdx.x, me = s.a.me, n = s.n;
    unsigned bad = 0;
    if (lane < n && lane != me) {
        p2p_st(s.a.f[lane] + kP2pPrimed + me, s.token);
        p2p_wait(mine + kP2pPrimed + lane, s.token, s.a.timeout_ticks, mine + kP2pErr);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __builtin_amdgcn_s_barrier();
    for (int q = 0; q < n; ++q) {
        if (q == me) continue;
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(s.a.p[q], 0, kSelfFloats * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(s.a.stage[q], 0, kSelfFloats * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(s.a.g[q], 0, kSelfFloats * 4, 0x00020000);
        if (lane < kSelfLine / 4) {
            v4f w;
#pragma unroll
            for (int e = 0; e < 4; ++e) w[e] = self_pat(s.token, me, q, 4 * lane + e, 0);
            const unsigned off = (unsigned)((me * kSelfLine + 4 * lane) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, w), rp, off, 0, 17);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, w), rs, off, 0, 17);
            const v4f g = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 17));
#pragma unroll
            for (int e = 0; e < 4; ++e) bad += g[e] != self_pat(s.token, q, me, 4 * lane + e, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __builtin_amdgcn_s_barrier();
    if (lane < n && lane != me) {
        p2p_st(s.a.f[lane] + kP2pWritten + me, s.token);
        p2p_wait(mine + kP2pWritten + lane, s.token, s.a.timeout_ticks, mine + kP2pErr);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __builtin_amdgcn_s_barrier();
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(s.a.stage[me], 0, kSelfFloats * 4, 0x00020000);
    const int q = (4 * lane) / kSelfLine;
    if (q < n && q != me) {
        const v4f v = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rst, (unsigned)(lane * 16), 0, 17));
#pragma unroll
        for (int e = 0; e < 4; ++e) bad += v[e] != self_pat(s.token, q, me, (4 * lane + e) % kSelfLine, 0);
    }
    if (bad) atomicAdd(mine + kP2pErr, bad);
}
// the fresh dependent launch: every XCD re-reads the parameter region through both forward-pass load paths and the
// staging region through plain and system-scope loads; the lines of the peers must hold what the peers wrote
__global__ void __launch_bounds__(64) p2p_self_verify_kernel(SelfArgs s) {
    __shared__ __attribute__((aligned(16))) float lds[kSelfFloats];
    const int lane = threadIdx.x, me = s.a.me;
    const v4f pd = self_read_dma(s.a.p[me], lds, lane);
    const v4f pl = *reinterpret_cast<const v4f*>(s.a.p[me] + 4 * lane);
    const v4f sl = *reinterpret_cast<const v4f*>(s.a.stage[me] + 4 * lane);
    const __amdgpu_buffer_rsrc_t rst = __builtin_amdgcn_make_buffer_rsrc(s.a.stage[me], 0, kSelfFloats * 4, 0x00020000);
    const v4f ss = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rst, (unsigned)(lane * 16), 0, 17));
    const int q = (4 * lane) / kSelfLine;
    unsigned bad = 0;
    if (q < s.n && q != me) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float want = self_pat(s.token, q, me, (4 * lane + e) % kSelfLine, 0);
            bad += (pd[e] != want) + (pl[e] != want) + (sl[e] != want) + (ss[e] != want);
        }
    }
    if (bad) atomicAdd(s.a.f[me] + kP2pErr, bad);
}
// pattern B over the gradient region (plain stores); the next launch tells the peers and reads theirs
__global__ void __launch_bounds__(64) p2p_self_gradb_kernel(SelfArgs s) {
    const int lane = threadIdx.x, me = s.a.me;
    v4f b;
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = self_pat(s.token, me, (4 * lane + e) / kSelfLine, (4 * lane + e) % kSelfLine, 1);
    *reinterpret_cast<v4f*>(s.a.g[me] + 4 * lane) = b;
}
__global__ void __launch_bounds__(64) p2p_self_reread_kernel(SelfArgs s) {
    unsigned* mine = s.a.f[s.a.me];
    const int lane = threadIdx.x, me = s.a.me, n = s.n;
    unsigned bad = 0;
    if (lane < n && lane != me) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        p2p_st(s.a.f[lane] + kP2pGradB + me, s.token);
        p2p_wait(mine + kP2pGradB + lane, s.token, s.a.timeout_ticks, mine + kP2pErr);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    __builtin_amdgcn_s_barrier();
    for (int q = 0; q < n; ++q) {
        if (q == me || lane >= kSelfLine / 4) continue;
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(s.a.g[q], 0, kSelfFloats * 4, 0x00020000);
        const v4f g = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rg, (unsigned)((me * kSelfLine + 4 * lane) * 4), 0, 17));
#pragma unroll
        for (int e = 0; e < 4; ++e) bad += g[e] != self_pat(s.token, q, me, 4 * lane + e, 1);
    }
    if (bad) atomicAdd(mine + kP2pErr, bad);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (lane < n && lane != me) {                 // nobody restores its regions while a peer may still be reading them
        p2p_st(s.a.f[lane] + kP2pFin + me, s.token);
        p2p_wait(mine + kP2pFin + lane, s.token, s.a.timeout_ticks, mine + kP2pErr);
    }
}
__global__ void __launch_bounds__(64) p2p_self_restore_kernel(SelfArgs s) {
    const int lane = threadIdx.x, me = s.a.me;
    *reinterpret_cast<v4f*>(s.a.p[me] + 4 * lane) = *reinterpret_cast<const v4f*>(s.save + 4 * lane);
    *reinterpret_cast<v4f*>(s.a.g[me] + 4 * lane) = *reinterpret_cast<const v4f*>(s.save + kSelfFloats + 4 * lane);
}

// ---------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------
static AdamScalars adam_scalars(const pvae_step_params* sp, int net) {
    // torch computes the bias corrections in Python floats (double): tm:119-122 -> torch/optim/adam.py
    const int t = sp->adam_t[net] > 0 ? sp->adam_t[net] : 1;
    const double bc1 = 1.0 - std::pow(sp->beta1, t);
    const double bc2 = 1.0 - std::pow(sp->beta2, t);
    AdamScalars s;
    s.step_size = (float)(sp->lr / bc1);
    s.inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
    s.beta1 = (float)sp->beta1;
    s.beta2 = (float)sp->beta2;
    s.eps = (float)sp->adam_eps;
    s.one_minus_beta1 = (float)(1.0 - sp->beta1);
    s.one_minus_beta2 = (float)(1.0 - sp->beta2);
    s.weight_decay = sp->weight_decay;
    return s;
}

static int check_ready(const pvae_ctx* c, bool need_arenas) {
    if (!c) return fail(-1, "null ctx");
    if (!c->ws) return fail(-2, "workspace not bound");
    if (need_arenas && !c->params) return fail(-2, "parameter arena not bound");
    return 0;
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
            out->net = l.net; out->index = l.index; out->n_in = l.n_in; out->n_out = l.n_out;
            out->ld = l.ld; out->n_out_pad = l.n_out_pad; out->w_offset = l.w_off; out->b_offset = l.b_off;
            out->act = l.act == 0 ? PVAE_ACT_LINEAR : l.act - 1; out->reserved = 0;
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
static int launch_sampler(pvae_ctx* c, const float* te_out, int ldte, const float* eps, float* eps_used, float* md_in,
                          int ld_md, int rows, int rows_pad, int noise, unsigned long long seed,
                          unsigned long long offset, float* partial, float* z_dense, const float* mu_p, int ldmp,
                          hipStream_t st) {
    const int Db = c->L.cfg.dim_body, Z = c->L.cfg.latent;
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
    if (fused && !(c->defer_adam && c->grads)) return false;          // (the same-layer schedule of plan_backward_net)
    if (Da > ProCols::kMaxN || Z > ProCols::kMaxN) return false;
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
    S.seed_sampler = backward && phase == PVAE_PHASE_JOINT && c->W.L == 1 && c->pair_launch &&
                     c->L.cfg.prior_kind < PVAE_PRIOR_HYPERSPHERE;
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
                           w + wmd.in + bt * ld_md, ld_md, Db, Z, rows, u.rows_pad, 1,
                           (unsigned long long)sp->rng_seed, (unsigned long long)(sp->rng_offset + t),
                           part + 2 * kLossParts + t * S.gridz, (float*)nullptr);
        HIP_TRY(hipGetLastError());
        FwdTail md_tail;                       // a_hat -> action columns of the predicted-action WM input
        md_tail.out2 = w + wwm.in + bp * ld_wm; md_tail.ld2 = ld_wm; md_tail.off2 = Db; md_tail.n2 = Da;
        if ((rc = forward_net(c, PVAE_NET_MD, u.rows_pad, st, md_tail, bt))) return rc;
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
                               w + wwm.act.back() + bp * ldo_wm, ldo_wm, rows, Db, w + wte.in + nt * ld_te, ld_te,
                               w + wmd.in + nt * ld_md, ld_md, w + wwm.in + np * ld_wm, ld_wm,
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
    std::vector<int> train_nets;
    if (joint) { train_nets.push_back(PVAE_NET_MD); train_nets.push_back(PVAE_NET_TE); }
    else train_nets.push_back(PVAE_NET_WM);
    int krows_of[PVAE_NUM_NETS] = {0, 0, 0, 0};
    if (backward) {
        for (int n : train_nets) {
            const NetLayout* N = &NL[n];
            const NetWork* nw = &NW[n];
            std::vector<char> act;
            if (n == PVAE_NET_WM) {
                for (int t = 0; t < T; ++t) act.push_back(u.use_g);
                for (int t = 0; t < T; ++t) act.push_back(p_act[t]);
            } else {
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
    bool paired_done[PVAE_NUM_NETS] = {false, false, false, false};
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
            if (pair_wm) paired_chain(PVAE_NET_WM, T + t, 1, true);
            else dgrad_chain(PVAE_NET_WM, T + t, upstream);
        }
        if (!upstream) continue;
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
        const bool pair_here = look_pair && t == 0 && joint;
        if (pair_here && krows_of[PVAE_NET_MD] > 0) paired_chain(PVAE_NET_MD, t, 0, false);
        else dgrad_chain(PVAE_NET_MD, t, true);
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

// ---- data-parallel exchange inside the library ---------------------------------------------
int pvae_comm_unique_id(void* id128) {
    if (!id128) return fail(-1, "null id buffer");
    int rc = rccl_load();
    if (rc) return rc;
    RcclId id;
    RCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id128, id.internal, sizeof(id.internal));
    return 0;
}

static int ensure_comm_stream(pvae_ctx* c);
int pvae_comm_init(pvae_ctx* c, int rank, int world, const void* id128) {
    if (!c || !id128) return fail(-1, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(-1, "rank %d outside [0, %d)", rank, world);
    if (c->comm) return fail(-2, "communicator already initialised");
    int rc = rccl_load();
    if (rc) return rc;
    RcclId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    void* comm = nullptr;
    RCCL_TRY(g_rccl.CommInitRank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    if ((rc = ensure_comm_stream(c))) return rc;
    return 0;
}

// the exchange stream and its events (bucketed + overlapped exchange), shared by the RCCL and the peer-mapped transport
static int ensure_comm_stream(pvae_ctx* c) {
    if (c->comm_stream) return 0;
    int lo = 0, hi = 0;                                   // hi = numerically lowest = most urgent
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, hi));
    for (hipEvent_t& e : c->bucket_ready) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->comm_done, hipEventDisableTiming));
    return 0;
}

// ---- peer-mapped exchange: set-up ------------------------------------------------------------
struct P2pBlob {                                          // PVAE_P2P_BLOB_BYTES on the wire
    uint32_t magic, abi;
    int64_t arena_floats;
    hipIpcMemHandle_t h[4];                               // allocations holding grads, params, flags, staging
    int64_t off[4];                                       // byte offset of the buffer inside its allocation
};
static_assert(sizeof(P2pBlob) <= PVAE_P2P_BLOB_BYTES, "blob layout");
constexpr uint32_t kP2pMagic = 0x50325056u;               // "PV2P"

int pvae_p2p_export(pvae_ctx* c, void* blob) {
    if (!c || !blob) return fail(-1, "null argument");
    if (!c->params || !c->grads) return fail(-2, "parameter / gradient arenas not bound");
    if (c->p2p.open) return fail(-2, "peer-mapped exchange is open: pvae_p2p_close before exporting again");
    if (!c->p2p.flags) HIP_TRY(hipExtMallocWithFlags((void**)&c->p2p.flags, kP2pFlagBytes, hipDeviceMallocUncached));
    // every set-up starts from a zeroed flag block (epochs restart at 0 in pvae_p2p_open): a block that an earlier,
    // closed set-up left its epochs in would satisfy the first waits of the new one.  The exchange of the blobs that
    // follows is the barrier between this and any peer's first write.
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(c->p2p.flags, 0, kP2pFlagBytes));
    HIP_TRY(hipDeviceSynchronize());
    c->p2p.selftests = 0;
    P2pBlob b;
    memset(&b, 0, sizeof(b));
    b.magic = kP2pMagic; b.abi = PVAE_ABI_VERSION; b.arena_floats = c->L.arena_floats;
    if (!c->p2p.staging) HIP_TRY(hipMalloc((void**)&c->p2p.staging, ((size_t)c->L.arena_floats + 256) * sizeof(float)));
    void* ptrs[4] = {c->grads, c->params, c->p2p.flags, c->p2p.staging};
    const char* what[4] = {"gradient arena", "parameter arena", "flag block", "staging buffer"};
    for (int k = 0; k < 4; ++k) {
        void* base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, ptrs[k]) != hipSuccess || !base)
            return fail(-10, "%s: not inside a hipMalloc allocation", what[k]);
        hipError_t e = hipIpcGetMemHandle(&b.h[k], base);
        if (e != hipSuccess)
            return fail(-10, "hipIpcGetMemHandle(%s): %s (the arenas must come from hipMalloc -- PyTorch's default "
                             "caching allocator, not expandable segments -- and HSA_ENABLE_IPC_MODE_LEGACY=0 must be set "
                             "where the driver only supports dmabuf IPC)", what[k], hipGetErrorString(e));
        b.off[k] = (char*)ptrs[k] - (char*)base;
    }
    memset(blob, 0, PVAE_P2P_BLOB_BYTES);
    memcpy(blob, &b, sizeof(b));
    return 0;
}

int pvae_p2p_close(pvae_ctx* c) {
    if (!c) return fail(-1, "null ctx");
    pvae_ctx::P2p& P = c->p2p;
    for (int q = 0; q < PVAE_P2P_MAX_RANKS; ++q)
        for (int k = 0; k < 4; ++k) {
            if (!P.mapped[q][k]) continue;
            bool dup = false;                             // one mapping may serve two buffers of a peer
            for (int j = 0; j < k; ++j) dup = dup || P.mapped[q][j] == P.mapped[q][k];
            if (!dup) (void)hipIpcCloseMemHandle(P.mapped[q][k]);
        }
    memset(P.mapped, 0, sizeof(P.mapped));
    memset(P.grads, 0, sizeof(P.grads)); memset(P.params, 0, sizeof(P.params)); memset(P.peer_flags, 0, sizeof(P.peer_flags));
    memset(P.peer_staging, 0, sizeof(P.peer_staging));
    P.open = false; P.world = 0; P.rank = 0;
    if (c->exchange_mode == PVAE_EXCHANGE_P2P || c->exchange_mode == PVAE_EXCHANGE_P2P_PUSH) c->exchange_mode = PVAE_EXCHANGE_ALLREDUCE;
    if (!c->comm) { c->comm_world = 1; c->comm_rank = 0; }
    return 0;
}

int pvae_p2p_open(pvae_ctx* c, int rank, int world, const void* blobs) {
    if (!c || !blobs) return fail(-1, "null argument");
    if (world < 1 || world > PVAE_P2P_MAX_RANKS || rank < 0 || rank >= world)
        return fail(-1, "rank %d / world %d outside [0, %d]", rank, world, PVAE_P2P_MAX_RANKS);
    pvae_ctx::P2p& P = c->p2p;
    if (P.open) return fail(-2, "peer-mapped exchange already open");
    if (!P.flags || !c->params || !c->grads) return fail(-2, "pvae_p2p_export first");
    if (c->comm && (c->comm_world != world || c->comm_rank != rank))
        return fail(-1, "rank %d / world %d differ from the RCCL communicator's %d / %d", rank, world, c->comm_rank, c->comm_world);
    const char* all = (const char*)blobs;
    for (int q = 0; q < world; ++q) {
        P2pBlob b;
        memcpy(&b, all + (size_t)q * PVAE_P2P_BLOB_BYTES, sizeof(b));
        if (b.magic != kP2pMagic || b.abi != PVAE_ABI_VERSION || b.arena_floats != c->L.arena_floats) {
            pvae_p2p_close(c);
            return fail(-1, "blob of rank %d does not describe a matching ctx", q);
        }
        if (q == rank) {
            P.grads[q] = c->grads; P.params[q] = c->params; P.peer_flags[q] = P.flags; P.peer_staging[q] = P.staging;
            continue;
        }
        void* base[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < 4; ++k) {
            for (int j = 0; j < k; ++j)                   // two buffers inside one allocation: open it once
                if (memcmp(&b.h[j], &b.h[k], sizeof(b.h[k])) == 0) base[k] = base[j];
            if (!base[k]) {
                hipError_t e = hipIpcOpenMemHandle(&base[k], b.h[k], hipIpcMemLazyEnablePeerAccess);
                if (e != hipSuccess) {
                    pvae_p2p_close(c);
                    return fail(-10, "hipIpcOpenMemHandle(rank %d, buffer %d): %s", q, k, hipGetErrorString(e));
                }
            }
            P.mapped[q][k] = base[k];
        }
        P.grads[q] = (float*)((char*)base[0] + b.off[0]);
        P.params[q] = (float*)((char*)base[1] + b.off[1]);
        P.peer_flags[q] = (unsigned*)((char*)base[2] + b.off[2]);
        P.peer_staging[q] = (float*)((char*)base[3] + b.off[3]);
    }
    P.rank = rank; P.world = world; P.epoch = 0; P.open = true;
    c->comm_rank = rank; c->comm_world = world;
    int rc = ensure_comm_stream(c);
    if (rc) return rc;
    return 0;
}

int pvae_p2p_status(pvae_ctx* c, int* rank, int* world, uint32_t* timeouts, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (rank) *rank = c->p2p.open ? c->p2p.rank : 0;
    if (world) *world = c->p2p.open ? c->p2p.world : 0;
    if (timeouts) {
        *timeouts = 0;
        if (c->p2p.flags) {
            HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
            if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
            HIP_TRY(hipMemcpy(timeouts, c->p2p.flags + kP2pErr, sizeof(uint32_t), hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

int pvae_p2p_selftest(pvae_ctx* c, void* stream) {
    if (!c) return fail(-1, "null ctx");
    pvae_ctx::P2p& P = c->p2p;
    if (!P.open) return fail(-2, "peer-mapped exchange not open (pvae_p2p_open)");
    hipStream_t st = (hipStream_t)stream;
    P2pArgs a;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < P.world; ++q) a.f[q] = P.peer_flags[q];
    a.me = P.rank;
    a.timeout_ticks = P.timeout_ticks < 100000000ll ? P.timeout_ticks : 100000000ll;      // at most 1 s
    // (one token per call, the same on every rank: the waits compare with >=, so a second self-test on the same flag
    //  block must not be satisfied by the first one's tokens)
    const unsigned token = 0x5E1F0000u + (++P.selftests) * 16u + (unsigned)P.world;
    uint32_t before = 0, after = 0;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&before, P.flags + kP2pErr, sizeof(before), hipMemcpyDeviceToHost));
    // (1) the uncached flag block: remote write, remote read, flag delivery
    hipLaunchKernelGGL(p2p_selftest_kernel, dim3(1), dim3(64), 0, st, a, P.world, token);
    HIP_TRY(hipGetLastError());
    // (2) the cached arenas, through the exchange's own access paths (see p2p_self_prime_kernel)
    const bool arenas = P.world > 1 && c->L.arena_floats >= kSelfFloats && !c->p2p_selftest_flags_only;
    if (arenas) {
        if (!P.self_buf) HIP_TRY(hipMalloc((void**)&P.self_buf, (2 * kSelfFloats + kSelfGrid) * sizeof(float)));
        SelfArgs s;
        memset(&s, 0, sizeof(s));
        s.a = a;
        for (int q = 0; q < P.world; ++q) {
            s.a.g[q] = P.grads[q]; s.a.p[q] = P.params[q];
            s.a.stage[q] = P.peer_staging[q] + c->L.arena_floats;        // the 256-float tail behind the arena-sized part
        }
        s.save = P.self_buf; s.sink = (unsigned*)(P.self_buf + 2 * kSelfFloats); s.n = P.world; s.token = token;
        hipLaunchKernelGGL(p2p_self_prime_kernel, dim3(kSelfGrid), dim3(64), 0, st, s);
        hipLaunchKernelGGL(p2p_self_write_kernel, dim3(1), dim3(64), 0, st, s);
        hipLaunchKernelGGL(p2p_self_verify_kernel, dim3(kSelfGrid), dim3(64), 0, st, s);
        hipLaunchKernelGGL(p2p_self_gradb_kernel, dim3(1), dim3(64), 0, st, s);
        hipLaunchKernelGGL(p2p_self_reread_kernel, dim3(1), dim3(64), 0, st, s);
        hipLaunchKernelGGL(p2p_self_restore_kernel, dim3(1), dim3(64), 0, st, s);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpy(&after, P.flags + kP2pErr, sizeof(after), hipMemcpyDeviceToHost));
    if (after != before) {
        HIP_TRY(hipMemcpy(P.flags + kP2pErr, &before, sizeof(before), hipMemcpyHostToDevice));
        return fail(-22, "peer-mapped exchange self-test failed on rank %d: %u record(s) / flag(s) / arena word(s) from peers wrong, "
                         "stale or missing", P.rank, after - before);
    }
    return 0;
}

/* Zero the "waits that gave up" word (after the caller has dealt with them: a rejected calibration candidate,
 * a restored snapshot).  Synchronises `stream`. */
int pvae_p2p_clear_errors(pvae_ctx* c, void* stream) {
    if (!c) return fail(-1, "null ctx");
    if (!c->p2p.flags) return 0;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (c->comm_stream) HIP_TRY(hipStreamSynchronize(c->comm_stream));
    const uint32_t zero = 0;
    HIP_TRY(hipMemcpy(c->p2p.flags + kP2pErr, &zero, sizeof(zero), hipMemcpyHostToDevice));
    return 0;
}

// one bucket through the peer-mapped exchange (see p2p_exchange_kernel)
static int p2p_exchange(pvae_ctx* c, int net, int64_t off, int64_t cnt, const pvae_step_params* sp, hipStream_t cs) {
    pvae_ctx::P2p& P = c->p2p;
    if (!P.open) return fail(-2, "peer-mapped exchange not open (pvae_p2p_open)");
    if (!c->m || !c->v) return fail(-2, "Adam moment arenas not bound");
    if (P.grads[P.rank] != c->grads || P.params[P.rank] != c->params) return fail(-2, "arenas were re-bound after pvae_p2p_export");
    if ((off & 3) || (cnt & 3) || cnt <= 0) return fail(-1, "bucket [%lld, +%lld) not float4-aligned", (long long)off, (long long)cnt);
    // (the kernels address a bucket through 32-bit buffer descriptors: one bucket stays below 4 GiB -- a billion
    //  parameters; larger stacks go through in several buckets, PVAE_DP_BUCKET_MB)
    if (cnt + 4 * (int64_t)P.world >= ((int64_t)1 << 30)) return fail(-1, "bucket of %lld floats: the peer-mapped exchange takes < 2^30 per bucket", (long long)cnt);
    P2pArgs a;
    memset(&a, 0, sizeof(a));
    for (int q = 0; q < P.world; ++q) {
        a.g[q] = P.grads[q] + off; a.p[q] = P.params[q] + off; a.f[q] = P.peer_flags[q]; a.stage[q] = P.peer_staging[q];
    }
    const bool push = c->exchange_mode == PVAE_EXCHANGE_P2P_PUSH;
    a.m = c->m + off; a.v = c->v + off;
    a.n4 = cnt / 4; a.me = P.rank; a.epoch = ++P.epoch; a.timeout_ticks = P.timeout_ticks;
    a.s = adam_scalars(sp, net);
    const long long slice = (a.n4 + P.world - 1) / P.world;
    int grid = (int)((slice + 255) / 256);
    if (grid > 256) grid = 256;
    if (grid < 1) grid = 1;
    const int ps = g_prof.begin_range(4, (double)cnt * sizeof(float), cs);
    switch (P.world) {
#define PVAE_P2P_CASE(N) case N:                                                                         \
        if (push) hipLaunchKernelGGL((p2p_push_exchange_kernel<N>), dim3(grid), dim3(256), 0, cs, a); \
        else hipLaunchKernelGGL((p2p_exchange_kernel<N>), dim3(grid), dim3(256), 0, cs, a);           \
        break;
        PVAE_P2P_CASE(1) PVAE_P2P_CASE(2) PVAE_P2P_CASE(3) PVAE_P2P_CASE(4)
        PVAE_P2P_CASE(5) PVAE_P2P_CASE(6) PVAE_P2P_CASE(7) PVAE_P2P_CASE(8)
#undef PVAE_P2P_CASE
        default: return fail(-1, "world %d", P.world);
    }
    HIP_TRY(hipGetLastError());
    g_prof.end_range(ps, cs);
    return 0;
}

int pvae_p2p_exchange(pvae_ctx* c, int net, int64_t offset, int64_t count, const pvae_step_params* sp, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!sp) return fail(-1, "null step params");
    if (net < 0 || net >= PVAE_NUM_NETS) return fail(-1, "bad net id %d", net);
    const NetLayout& N = c->L.net[net];
    if (offset < N.off || count < 0 || offset + count > N.off + N.count)
        return fail(-1, "segment [%lld, +%lld) not inside net %d", (long long)offset, (long long)count, net);
    if (count == 0) return 0;
    params_touched(c, (hipStream_t)stream);
    return p2p_exchange(c, net, offset, count, sp, (hipStream_t)stream);
}

int pvae_comm_mode(pvae_ctx* c, int mode) {
    if (!c) return fail(-1, "null ctx");
    if (mode == PVAE_EXCHANGE_P2P || mode == PVAE_EXCHANGE_P2P_PUSH) {
        if (!c->p2p.open) return fail(-2, "peer-mapped exchange not open (pvae_p2p_export / pvae_p2p_open)");
        c->exchange_mode = mode;
        return 0;
    }
    if (mode == PVAE_EXCHANGE_LOCAL) {
        if (!c->comm && !c->p2p.open) return fail(-2, "no communicator and no peer-mapped exchange");
        c->exchange_mode = mode;
        return 0;
    }
    if (mode != PVAE_EXCHANGE_ALLREDUCE && mode != PVAE_EXCHANGE_SHARDED) return fail(-1, "unknown exchange mode %d", mode);
    if (mode == PVAE_EXCHANGE_SHARDED) {
        int rc = rccl_load();
        if (rc) return rc;
        if (!g_rccl.ReduceScatter || !g_rccl.AllGather) return fail(-20, "RCCL lacks ncclReduceScatter / ncclAllGather");
    }
    c->exchange_mode = mode;
    return 0;
}

int pvae_comm_info(pvae_ctx* c, int* rank, int* nranks) {
    if (!c || !rank || !nranks) return fail(-1, "null argument");
    *rank = 0; *nranks = 0;
    if (!c->comm) return 0;                    // no communicator: 0 ranks
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) return fail(-20, "RCCL lacks ncclCommCount / ncclCommUserRank");
    RCCL_TRY(g_rccl.CommCount(c->comm, nranks));
    RCCL_TRY(g_rccl.CommUserRank(c->comm, rank));
    return 0;
}

int pvae_comm_config(pvae_ctx* c, int64_t bucket_bytes, int32_t test_delay_us) {
    if (!c) return fail(-1, "null ctx");
    if (bucket_bytes < 0 || test_delay_us < 0 || test_delay_us > 100000) return fail(-1, "bad exchange settings");
    c->bucket_bytes = bucket_bytes;
    c->comm_test_delay_us = test_delay_us;
    return 0;
}

int pvae_comm_destroy(pvae_ctx* c) {
    if (!c) return fail(-1, "null ctx");
    if (c->comm) {
        RCCL_TRY(g_rccl.CommDestroy(c->comm));
        c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    }
    if (c->p2p.open) { c->comm_world = c->p2p.world; c->comm_rank = c->p2p.rank; }
    if (c->comm_stream && !c->p2p.open) {
        HIP_TRY(hipStreamSynchronize(c->comm_stream));
        HIP_TRY(hipStreamDestroy(c->comm_stream));
        c->comm_stream = nullptr;
        for (hipEvent_t& e : c->bucket_ready) { if (e) (void)hipEventDestroy(e); e = nullptr; }
        if (c->comm_done) (void)hipEventDestroy(c->comm_done);
        c->comm_done = nullptr;
    }
    return 0;
}

// Exchange buckets of one stack: whole layers, last layer first (the order the backward pass
// finishes them), closed as soon as they hold bucket_bytes.  A function of the layout and the
// bucket size only, so every rank -- also one whose shard of a ragged last batch is empty --
// issues the same sequence of reductions.
struct Bucket { int64_t off, cnt; };
// Default exchange schedule.  With one rank there is nothing to hide: in line.  With several ranks the
// all-reduce of a stack (14 MB) takes about as long over xGMI as the backward pass of a stack (~100 us), so
// in the JOINT phase the decoder's reduction is worth hiding behind the encoder's backward pass even at the
// ~27 us the two stream hand-offs cost (section 5 of DESIGN.md): 6 MiB buckets on the exchange stream.  The
// world phase has one stack and ~36 us of backward left after its first bucket closes: in line.
// A function of (communicator size, phase) only, so every rank chooses the same.
static int64_t auto_bucket_bytes(const pvae_ctx* c, int phase) {
    if (c->bucket_bytes >= 0) return c->bucket_bytes;
    return (c->comm_world > 1 && phase == PVAE_PHASE_JOINT && c->comm_stream) ? (int64_t)6 << 20 : 0;
}
static std::vector<Bucket> exchange_buckets(const pvae_ctx* c, int net) {
    const NetLayout& N = c->L.net[net];
    std::vector<Bucket> out;
    if (c->bucket_bytes_now <= 0) { out.push_back({N.off, N.count}); return out; }
    int64_t end = N.off + N.count;
    for (int i = (int)N.layers.size() - 1; i >= 0; --i) {
        const int64_t lo = i == 0 ? N.off : N.layers[i].w_off;
        if ((end - lo) * (int64_t)sizeof(float) >= c->bucket_bytes_now || i == 0) {
            out.push_back({lo, end - lo});
            end = lo;
        }
    }
    return out;
}

int pvae_owned_slices(pvae_ctx* c, int phase, int net, int64_t* offsets, int64_t* counts, int32_t* replicated,
                      int32_t max, int32_t* n) {
    if (!c || !offsets || !counts || !replicated || !n) return fail(-1, "null argument");
    if (net < 0 || net >= PVAE_NUM_NETS) return fail(-1, "bad net id %d", net);
    if (phase != PVAE_PHASE_WORLD && phase != PVAE_PHASE_JOINT) return fail(-1, "unknown phase %d", phase);
    *n = 0;
    if (c->L.net[net].layers.empty()) return 0;
    const int64_t N = c->comm_world > 0 ? c->comm_world : 1, r = c->comm_rank;
    const int64_t keep = c->bucket_bytes_now;
    c->bucket_bytes_now = auto_bucket_bytes(c, phase);
    const std::vector<Bucket> bk = exchange_buckets(c, net);
    c->bucket_bytes_now = keep;
    const bool p2p = c->exchange_mode == PVAE_EXCHANGE_P2P || c->exchange_mode == PVAE_EXCHANGE_P2P_PUSH;
    for (const Bucket& b : bk) {
        int64_t off = b.off, cnt = b.cnt;
        int rep = 1;
        if (N > 1 && p2p) {
            const int64_t n4 = b.cnt / 4, S = (n4 + N - 1) / N, lo = r * S, hi = lo + S < n4 ? lo + S : n4;
            off = b.off + 4 * lo; cnt = lo < hi ? 4 * (hi - lo) : 0; rep = 0;
        } else if (N > 1 && c->exchange_mode == PVAE_EXCHANGE_SHARDED && g_rccl.ReduceScatter && g_rccl.AllGather &&
                   b.cnt % (N * 4) == 0 && b.cnt > 0) {
            cnt = b.cnt / N; off = b.off + r * cnt; rep = 0;
        }
        if (*n >= max) return fail(-1, "more than %d buckets", (int)max);
        offsets[*n] = off; counts[*n] = cnt; replicated[*n] = rep;
        ++*n;
    }
    return 0;
}

__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// Reduce one bucket over the ranks and apply Adam to it.  `cs` == `st`: in line.  Otherwise the
// bucket is handed to the exchange stream behind an event, and the compute stream carries on.
static int exchange_bucket(pvae_ctx* c, int net, const Bucket& b, const pvae_step_params* sp, hipStream_t st,
                           hipStream_t cs, int& n_events) {
    int rc;
    if (cs != st) {
        if (n_events >= pvae_ctx::kMaxBuckets) return fail(-2, "more than %d exchange buckets in a step", pvae_ctx::kMaxBuckets);
        hipEvent_t e = c->bucket_ready[n_events++];
        HIP_TRY(hipEventRecord(e, st));
        HIP_TRY(hipStreamWaitEvent(cs, e, 0));
    }
    if (c->comm_test_delay_us > 0) {
        hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, cs, (long long)c->comm_test_delay_us * 100);   // 100 MHz
        HIP_TRY(hipGetLastError());
    }
    // Sharded exchange (PVAE_EXCHANGE_SHARDED, ZeRO-1 shaped): every rank reduces only ITS 1/N slice of the
    // bucket (reduce-scatter, in place), applies Adam to that slice (1/N of the p, g, m, v traffic) and the
    // updated parameter slices are all-gathered in place.  Same bytes on the links as a ring all-reduce;
    // the moments of the other ranks' slices are never touched here (they stay at whatever they were).
    if (c->exchange_mode == PVAE_EXCHANGE_P2P || c->exchange_mode == PVAE_EXCHANGE_P2P_PUSH)
        return p2p_exchange(c, net, b.off, b.cnt, sp, cs);
    if (c->exchange_mode == PVAE_EXCHANGE_LOCAL) return pvae_adam_segment(c, net, b.off, b.cnt, sp, cs);
    const int64_t N = c->comm_world;
    if (c->exchange_mode == PVAE_EXCHANGE_SHARDED && g_rccl.ReduceScatter && g_rccl.AllGather &&
        b.cnt % (N * 4) == 0 && b.cnt > 0) {
        const int64_t slice = b.cnt / N, mine = b.off + c->comm_rank * slice;
        int ps = g_prof.begin_range(4, (double)b.cnt * sizeof(float), cs);
        RCCL_TRY(g_rccl.ReduceScatter(c->grads + b.off, c->grads + mine, (size_t)slice, kNcclFloat32, kNcclSum, c->comm, cs));
        g_prof.end_range(ps, cs);
        if ((rc = pvae_adam_segment(c, net, mine, slice, sp, cs))) return rc;
        ps = g_prof.begin_range(4, (double)b.cnt * sizeof(float), cs);
        RCCL_TRY(g_rccl.AllGather(c->params + mine, c->params + b.off, (size_t)slice, kNcclFloat32, c->comm, cs));
        g_prof.end_range(ps, cs);
        return 0;
    }
    if ((rc = pvae_allreduce_grads(c, b.off, b.cnt, cs))) return rc;
    return pvae_adam_segment(c, net, b.off, b.cnt, sp, cs);
}

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
    c->dx.on = true;
    c->dx.rm = row_map(c, first_window, rows);
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

int pvae_allreduce_grads(pvae_ctx* c, int64_t offset, int64_t count, void* stream) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!c->comm) return fail(-2, "no communicator (pvae_comm_init)");
    if (!c->grads) return fail(-2, "gradient arena not bound");
    if (offset < 0 || count < 0 || offset + count > c->L.arena_floats)
        return fail(-1, "slice [%lld, +%lld) outside the arena", (long long)offset, (long long)count);
    if (count == 0) return 0;
    const int ps = g_prof.begin_range(4, (double)count * sizeof(float), (hipStream_t)stream);
    RCCL_TRY(g_rccl.AllReduce(c->grads + offset, c->grads + offset, (size_t)count, kNcclFloat32, kNcclSum, c->comm,
                              (hipStream_t)stream));
    g_prof.end_range(ps, (hipStream_t)stream);
    return 0;
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
    const int nets[3] = {phase == PVAE_PHASE_WORLD ? PVAE_NET_WM : PVAE_NET_MD,
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
        case 2: src = c->ws + c->W.net[PVAE_NET_MD].in; ld = c->L.net[PVAE_NET_MD].layers[0].ld; col0 = Db; nc = Z; break;
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
    const bool fused_rollout = rollout_fused();
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
    const bool direct = !s2_hat && (rows == 1 || rows == 2 || rows == 4);
    FwdTail md_tail;
    if (direct) {
        md_tail.out2 = a_hat; md_tail.ld2 = ld_a; md_tail.off2 = 0; md_tail.n2 = Da;
    } else {
        md_tail.out2 = w + c->W.net[PVAE_NET_WM].in; md_tail.ld2 = WM.layers[0].ld; md_tail.off2 = Db; md_tail.n2 = Da;
    }
    if ((rc = forward_net(c, PVAE_NET_MD, rows_pad, st, md_tail))) return rc;
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

// ---------------------------------------------------------------------------------------
// Call-persistent rollout server (rmt:742-771 at B = 1; callers envs/rllib_env_imitation.py:215-266).
//
// The per-layer launches above cost the control loop 7 dependent launches whose weights are a cold fetch each
// (35 us device -> device).  Here ONE kernel stays resident across calls on the 32 CUs of ONE XCD.  It copies the
// encoder's and the decoder's weights into LDS once (every workgroup holds the rows of 1/32 of every layer's output
// features: 3.4 MB over 32 x 160 KB for the default stacks), then serves requests from a mailbox in pinned host
// memory: workgroup 0 polls the request word over PCIe, fetches the observation, and releases the other 31
// through a word in the XCD's L2; every layer is a GEMV from LDS-resident weights followed by the single-XCD L2
// barrier of tools/xcd_barrier.hip (arrive = atomic add that executes in the L2, poll = sc1 load, payload = plain
// stores drained before arriving and read back with sc1 loads: 1.2 us per round); the sampler of rmt:734-740 is
// formed in place by every workgroup; after the last barrier workgroup 0 writes [a_hat | mu | logvar | z] to the
// mailbox and then the completion word.  No launch, no stream operation and no cold weight fetch per call.
// The arithmetic of a feature is gemv_rollout_kernel's, operation for operation (same lane -> k mapping, same fma
// chain, same butterfly), so the action equals pvae_infer's bit for bit.
// Bounded by construction: workgroup 0 gives up after `idle_ticks` without a request (the host relaunches on the
// next call), every spin re-checks an absolute lifetime, and a stop command ends it at once.  While it is resident,
// a DEVICE-wide synchronisation (hipDeviceSynchronize, hipFree) waits for it -- at most the idle time-out.
// ---------------------------------------------------------------------------------------
constexpr int kSrvMaxLayers = 12, kSrvMaxObs = 2048, kSrvMaxOut = 1024;
struct SrvRequest {                       // host -> device.  Lives in DEVICE memory when the host can write it directly
    // (large BAR: the host PUSHES the observation and the kernel polls local memory), else in pinned host memory (the
    // kernel PULLS over PCIe).  ONE 32-byte line of control words, then the observation.
    volatile uint32_t req_seq;            // written LAST by the host: request number
    uint32_t cmd;                         // 0 infer, 1 stop, 2 reload the weights from the arena, then infer,
                                          // 3 decoder only ("pass_through", rllib_env_imitation.py:233-258): obs = [s1 (Db) | z (Z)]
    uint32_t noise, pad0;
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    uint32_t pad1[8];
    float obs[kSrvMaxObs];
};
struct SrvReply {                         // device -> host, pinned host memory (the host spins on its own RAM)
    volatile uint32_t done_seq;           // written LAST by the device: the request this result belongs to
    volatile uint32_t state;              // 0 not started, 1 serving, 2 exited (idle / stop / lifetime), 3 refused (placement)
    uint32_t served, pad2[13];
    float out[kSrvMaxOut];                // [a_hat (Da) | mu (Z) | logvar (Z) | z (Z)]
};
struct SrvLayer { long long w_off, b_off; int ld, n_out_pad, n_out, act, F, lds_off; };   // F: features per group (the last
                                                                                          // active group may own fewer)
constexpr int kSrvActStride = 2048;
struct SrvArgs {
    SrvLayer layer[kSrvMaxLayers];
    int n_layers, n_te;                   // layers [0, n_te) are the encoder's, the rest the decoder's
    int groups, one_xcd;                  // 32 workgroups on ONE XCD, or 256 over the whole chip (stacks too big for one XCD's LDS)
    int xcd;                              // which XCD (one_xcd): servers of one process take different ones
    int Db, Da, Z, prior_kind;
    const float* params;
    unsigned long long* acts;             // [n_layers + 1][kSrvActStride] TAGGED values: slot 0 = the observation, slot l + 1 = layer l's output
    unsigned* sync;                       // device words: 0 start-up barrier, 1 go_seq, 2..8 the request's control words, 16 xcc of group 0, 17 error, 18 group 0 has left
    SrvRequest* req;                      // device view of the request block
    SrvReply* mb;                         // device view of the reply block
    int obs_direct;                       // the request block is device memory: every group reads the observation from it
    long long idle_ticks, life_ticks;     // 100 MHz wall clock
    int xs_off;                           // float offset of the input vector inside the dynamic LDS
    unsigned seq0;                        // requests served by earlier instances (this one answers seq0 + 1, ...)
    unsigned long long* dbg;              // [64] wall-clock stamps of group 0 for the LAST request (pvae_rollout_server_timeline)
};
__device__ inline unsigned srv_ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }   // sc1: past the L1
// A value travels between workgroups as ONE 8-byte word {tag, float bits}: the consumer polls the word itself (sc1 loads,
// served by the XCD's L2) until it carries the tag of this request and layer -- no barrier between a layer and the next,
// one L2 round trip after the producer's store has landed.  Tags only grow (request * 16 + layer), so a word left over from
// an earlier request can never be mistaken.
__device__ inline void srv_put(unsigned long long* slot, float v, unsigned tag) {
    __hip_atomic_store(slot, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline float srv_get(const unsigned long long* slot, unsigned tag, long long t_start, long long life, int& failed,
                                const unsigned* gone) {
    unsigned long long u;
    unsigned spins = 0;
    while ((unsigned)((u = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 32) != tag) {
        // (a producer that never comes: group 0 left on its idle time-out just as this request arrived, or the lifetime is over)
        if ((++spins & 255u) == 0 && (srv_ldu(gone) != 0u || wall_clock64() - t_start > life)) { failed = 1; break; }
    }
    return __uint_as_float((unsigned)u);
}

__device__ inline void srv_get2(const unsigned long long* p0, const unsigned long long* p1, unsigned tag, float& v0, float& v1,
                                long long t_start, long long life, int& failed, const unsigned* gone) {
    unsigned long long u0, u1;
    unsigned spins = 0;
    for (;;) {
        u0 = __hip_atomic_load(p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u1 = __hip_atomic_load(p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(u0 >> 32) == tag && (unsigned)(u1 >> 32) == tag) break;
        if ((++spins & 255u) == 0 && (srv_ldu(gone) != 0u || wall_clock64() - t_start > life)) { failed = 1; break; }
    }
    v0 = __uint_as_float((unsigned)u0);
    v1 = __uint_as_float((unsigned)u1);
}
// xs[k] = word k of `prev` for k = tid, tid + 256, ... < n (tag `tag`), ALL of a thread's words polled together: their loads are
// in flight at once and a spin costs one round trip whatever the layer's width (one word after the other, a 1024-wide input
// cost four round trips per layer: 34 us for the 4x1024 stacks against 19 now)
__device__ inline void srv_get_row(float* xs, const unsigned long long* prev, int n, int ld, unsigned tag, int tid, long long t_start,
                                   long long life, int& failed, const unsigned* gone) {
    constexpr int kMax = kSrvActStride / 256;
    unsigned long long u[kMax];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            const int k = tid + 256 * i;
            u[i] = k < n ? __hip_atomic_load(prev + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int i = 0; i < kMax; ++i) all = all && (unsigned)(u[i] >> 32) == tag;
        if (all) break;
        if ((++spins & 255u) == 0 && (srv_ldu(gone) != 0u || wall_clock64() - t_start > life)) { failed = 1; break; }
    }
#pragma unroll
    for (int i = 0; i < kMax; ++i) {
        const int k = tid + 256 * i;
        if (k < ld) xs[k] = k < n ? __uint_as_float((unsigned)u[i]) : 0.f;
    }
}

// Lane 0's value of `for (o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64)`: the halving tree r[i] += r[i + h], h = 32 ... 1
// (additions commute, so only the association matters), with the two cross-row steps as gfx950's permlane swaps and the
// four in-row steps as DPP row shifts -- register moves, where __shfl_xor compiles to a ds_bpermute round trip per step.
// Lanes other than 0 hold partial garbage.
__device__ inline float srv_tree_sum(float v) {
    unsigned u = __float_as_uint(v);
    v += __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);       // lanes 0..31 += lanes 32..63
    u = __float_as_uint(v);
    v += __uint_as_float(__builtin_amdgcn_permlane16_swap(u, u, false, false)[1]);       // lanes 0..15 += lanes 16..31
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x108, 0xf, 0xf, true));   // row_shl:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x104, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x102, 0xf, 0xf, true));
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x101, 0xf, 0xf, true));
    return v;
}

__global__ void __launch_bounds__(256) rollout_server_kernel(SrvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float srv_lds[];
    __shared__ unsigned s_word[8];
    __shared__ int s_failed;
    if (a.one_xcd && (int)(blockIdx.x & 7) != a.xcd) return;   // workgroup b runs on XCD b % 8: the 32 of one XCD stay
    const int g = a.one_xcd ? blockIdx.x >> 3 : blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long t_start = wall_clock64();
    float* xs = srv_lds + a.xs_off;
    unsigned* ctr = a.sync;
    if (tid == 0) s_failed = 0;
    // placement check: all 32 groups must sit on the XCD of group 0 (the hand-overs live in ITS L2)
    const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7u;
    if (tid == 0) {
        if (g == 0) {
            __hip_atomic_store(a.sync + 1, a.seq0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // (go word: nothing new yet)
            __hip_atomic_store(a.sync + 16, xcc + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        unsigned x0;
        while ((x0 = srv_ldu(a.sync + 16)) == 0u) {
            if (wall_clock64() - t_start > a.life_ticks) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (a.one_xcd && x0 != xcc + 1u) atomicAdd(a.sync + 17, 1u);
    }
    auto load_weights = [&]() {
        for (int l = 0; l < a.n_layers; ++l) {
            const SrvLayer L = a.layer[l];
            int nf = L.n_out_pad - g * L.F;                // this group's features of the layer (0: none -- narrow layers
            nf = nf < 0 ? 0 : (nf > L.F ? L.F : nf);       //  leave the last groups idle)
            const int n4 = nf * L.ld / 4;                  // its rows are contiguous in the arena
            const v4f* src = reinterpret_cast<const v4f*>(a.params + L.w_off + (long long)g * L.F * L.ld);
            v4f* dst = reinterpret_cast<v4f*>(srv_lds + L.lds_off);
            for (int i = tid; i < n4; i += 256) dst[i] = src[i];
            if (tid < nf) srv_lds[L.lds_off + L.F * L.ld + tid] = a.params[L.b_off + g * L.F + tid];
        }
        __syncthreads();
    };
    load_weights();
    bool alive = true;
    {   // start-up barrier in the XCD's L2 (once): everybody placed, checked and loaded
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned ok = 1;
            while (srv_ldu(ctr) < (unsigned)a.groups) {
                if (wall_clock64() - t_start > a.life_ticks) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            s_word[7] = ok;
        }
        __syncthreads();
        if (s_word[7] == 0) alive = false;
    }
    if (alive && srv_ldu(a.sync + 17) != 0u) {              // not on one XCD: refuse (the host falls back to the launches)
        if (g == 0 && tid == 0) { a.mb->state = 3; __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
        return;
    }
    if (g == 0 && tid == 0 && alive) { a.mb->state = 1; __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); }
    unsigned last = a.seq0;                                // (request numbers keep growing across instances of the kernel:
    while (alive) {                                        //  the tags of the hand-over words derive from them)
        // ---- wait for a request: wave 0 of group 0 polls the mailbox's control line, the other groups the go word in the L2 ----
        if (g == 0) {
            if (wave == 0) {
                const long long t_idle = wall_clock64();
                const unsigned* line = (const unsigned*)&a.req->req_seq;
                unsigned w = 0, seq = last, cmd = 1;
                for (;;) {
                    if (lane < 8) w = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // one 32-byte read
                    seq = __builtin_amdgcn_readlane(w, 0);
                    if (seq != last) { cmd = __builtin_amdgcn_readlane(w, 1); break; }
                    const long long now = wall_clock64();
                    if (now - t_idle > a.idle_ticks || now - t_start > a.life_ticks) { seq = last + 1u; cmd = 1; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                // (no acquire fence: it would invalidate the L2, ~1.7 us; everything read after this point is read with
                //  system- / agent-scope loads that do not hit stale lines, issued behind the load that saw the request word)
                if (lane == 0) { s_word[0] = seq; s_word[1] = cmd; }
                if (lane >= 2 && lane < 8) s_word[lane] = w;                   // noise, pad, seed lo / hi, offset lo / hi
                if (!a.obs_direct) {
                    // release the other groups at once (they start polling the observation's words)
                    if (lane >= 1 && lane < 8)
                        __hip_atomic_store(a.sync + 1 + lane, lane == 1 ? cmd : w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (the control words have landed; a release store would
                    if (lane == 0) __hip_atomic_store(a.sync + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   //  write the L2 back)
                }
            }
            __syncthreads();
            if (s_word[1] != 1u && !a.obs_direct) {        // the observation: pinned host memory -> slot 0, tagged
                const unsigned tag0 = s_word[0] * 16u;
                const int n = s_word[1] == 3u ? a.Db + a.Z : 2 * a.Db;
                for (int i = tid; i < n; i += 256)
                    srv_put(a.acts + i, __hip_atomic_load(a.req->obs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), tag0);
            }
        } else if (a.obs_direct) {
            // the request block is device memory: every group watches its control line itself (no hop through group 0);
            // group 0's own exits (idle time-out, lifetime) still arrive through the go word
            if (wave == 0) {
                const unsigned* line = (const unsigned*)&a.req->req_seq;
                unsigned w = 0, seq = last, cmd = 1, polls = 0;
                for (;;) {
                    if (lane < 8) w = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    seq = __builtin_amdgcn_readlane(w, 0);
                    if (seq != last) { cmd = __builtin_amdgcn_readlane(w, 1); break; }
                    if ((++polls & 15u) == 0) {
                        if (srv_ldu(a.sync + 18) != 0u || wall_clock64() - t_start > a.life_ticks + 100000000ll) { seq = last + 1u; cmd = 1; w = 0; break; }
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                // (no acquire fence: it would invalidate the L2, ~1.7 us; everything read after this point is read with
                //  system- / agent-scope loads that do not hit stale lines, issued behind the load that saw the request word)
                if (lane == 0) { s_word[0] = seq; s_word[1] = cmd; }
                if (lane >= 2 && lane < 8) s_word[lane] = w;
            }
        } else {
            if (tid == 0) {
                unsigned seq;
                while ((seq = srv_ldu(a.sync + 1)) == last) {
                    if (wall_clock64() - t_start > a.life_ticks + 100000000ll) { seq = last + 1u; break; }   // (group 0 is gone)
                    __builtin_amdgcn_s_sleep(1);
                }
                s_word[0] = seq;
                s_word[1] = 1;
                if (seq == srv_ldu(a.sync + 1))
                    for (int i = 1; i < 8; ++i) s_word[i] = srv_ldu(a.sync + 1 + i);
            }
        }
        __syncthreads();
        last = s_word[0];
        const unsigned cmd = s_word[1];
        if (cmd == 1u) break;
        if (cmd == 2u) load_weights();
        const int noise = (int)s_word[2];
        const unsigned long long seed = s_word[4] | ((unsigned long long)s_word[5] << 32);
        const unsigned long long offset = s_word[6] | ((unsigned long long)s_word[7] << 32);
        const unsigned tag0 = last * 16u;
        int failed = 0;
        const bool stamp = g == 0 && tid == 0;
        if (stamp) { a.dbg[0] = wall_clock64(); a.dbg[62] = (unsigned long long)clock64(); }   // request seen by group 0 (+ shader clock)
        // ---- the layers: inputs polled word by word, outputs published word by word ----
        const bool decode_only = cmd == 3u;                                  // the caller supplies z: the encoder is skipped
        for (int l = decode_only ? a.n_te : 0; l < a.n_layers; ++l) {
            const SrvLayer L = a.layer[l];
            const unsigned long long* prev = a.acts + (size_t)l * kSrvActStride;   // slot l: the previous layer's output (0: obs)
            const unsigned tagp = tag0 + (unsigned)l;
            if (l == 0) {                                                    // [s1 | s2 | 0]
                if (a.obs_direct) {                                          // (complete before the request word)
                    for (int k = tid; k < L.ld; k += 256)
                        xs[k] = k < 2 * a.Db ? __hip_atomic_load(a.req->obs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
                } else {
                    srv_get_row(xs, a.acts, 2 * a.Db, L.ld, tag0, tid, t_start, a.life_ticks, failed, a.sync + 18);
                }
            } else if (l == a.n_te && decode_only) {                         // [s1 | z | 0] as the caller sent it
                if (a.obs_direct) {
                    for (int k = tid; k < L.ld; k += 256)
                        xs[k] = k < a.Db + a.Z ? __hip_atomic_load(a.req->obs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
                } else {
                    srv_get_row(xs, a.acts, a.Db + a.Z, L.ld, tag0, tid, t_start, a.life_ticks, failed, a.sync + 18);
                }
            } else if (l == a.n_te) {                                        // [s1 | z | 0], the sampler formed in place
                for (int k = tid; k < L.ld; k += 256) {
                    float v = 0.f;
                    if (k < a.Db) v = a.obs_direct ? __hip_atomic_load(a.req->obs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                   : srv_get(a.acts + k, tag0, t_start, a.life_ticks, failed, a.sync + 18);
                    else if (k < a.Db + a.Z) {
                        const int j = k - a.Db;
                        if (a.prior_kind == PVAE_PRIOR_NONE) v = srv_get(prev + j, tagp, t_start, a.life_ticks, failed, a.sync + 18);
                        else {
                            const float e = noise ? philox_normal(seed, offset, 0, j) : 0.f;       // (before the wait: off its path)
                            float mu, lv;
                            srv_get2(prev + j, prev + a.Z + j, tagp, mu, lv, t_start, a.life_ticks, failed, a.sync + 18);
                            v = mu + e * expf(0.5f * lv);
                        }
                    }
                    xs[k] = v;
                }
            } else {
                srv_get_row(xs, prev, L.ld, L.ld, tagp, tid, t_start, a.life_ticks, failed, a.sync + 18);
            }
            if (failed) s_failed = 1;
            __syncthreads();
            if (stamp) a.dbg[1 + 2 * l] = wall_clock64();                    // layer l: inputs in LDS
            const float* Wl = srv_lds + L.lds_off;
            unsigned long long* outp = a.acts + (size_t)(l + 1) * kSrvActStride;
            // one wave per feature, gemv_rollout_kernel's sum operation for operation -- four features of the wave at a time,
            // so that their reductions overlap, and the butterfly as register moves (srv_tree_sum) instead of six
            // ds_bpermute round trips per feature
            int nf = L.n_out_pad - g * L.F;
            nf = nf < 0 ? 0 : (nf > L.F ? L.F : nf);
            for (int f0 = wave; f0 < nf; f0 += 16) {
                const int cnt = (nf - f0 + 3) >> 2;                          // features f0, f0 + 4, ... of this wave in this pass
                auto rows = [&](auto nrows) {                                // (one unguarded body per count: the LDS reads of a
                    constexpr int N = decltype(nrows)::value;                //  k-step are in flight together)
                    float acc[N];
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[i] = 0.f;
                    for (int k = lane * 4; k < L.ld; k += 256) {
                        const v4f xv = *reinterpret_cast<const v4f*>(xs + k);
                        v4f wv[N];
#pragma unroll
                        for (int i = 0; i < N; ++i) wv[i] = *reinterpret_cast<const v4f*>(Wl + (f0 + 4 * i) * L.ld + k);
#pragma unroll
                        for (int i = 0; i < N; ++i)
                            acc[i] = fmaf(wv[i].x, xv.x, fmaf(wv[i].y, xv.y, fmaf(wv[i].z, xv.z, fmaf(wv[i].w, xv.w, acc[i]))));
                    }
#ifdef PVAE_SRV_FINE
                    if (stamp && l == 4) a.dbg[40] = wall_clock64();
#endif
#pragma unroll
                    for (int i = 0; i < N; ++i) acc[i] = srv_tree_sum(acc[i]);
#ifdef PVAE_SRV_FINE
                    if (stamp && l == 4) a.dbg[41] = wall_clock64();
#endif
                    // lane i finishes feature i (bias, activation, hand-over word): the N epilogues run side by side instead of
                    // one after the other on lane 0 (0.6 us of a 1.3 us layer when they did)
                    float mine = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i) {
                        const float si = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(acc[i]), 0));
                        mine = lane == i ? si : mine;
                    }
                    if (lane < N) {
                        const int f = f0 + 4 * lane, n = g * L.F + f;
                        float v = mine + Wl[L.F * L.ld + f];
                        v = (L.act > 1 && n >= L.n_out) ? 0.f : act_apply(v, L.act);
                        srv_put(outp + n, v, tagp + 1u);
                    }
                };
#ifdef PVAE_SRV_FINE
                if (stamp && l == 4) a.dbg[39] = wall_clock64();
#endif
                if (cnt >= 4) rows(std::integral_constant<int, 4>());
                else if (cnt == 3) rows(std::integral_constant<int, 3>());
                else if (cnt == 2) rows(std::integral_constant<int, 2>());
                else rows(std::integral_constant<int, 1>());
            }
            // (a bare barrier: only LDS is shared here.  __syncthreads() would also wait for the hand-over stores above to be
            //  acknowledged by the memory system -- half a microsecond per layer that now overlaps the next layer's polling)
#ifdef PVAE_SRV_FINE
            if (stamp && l == 4) a.dbg[42] = wall_clock64();
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (stamp) a.dbg[2 + 2 * l] = wall_clock64();                    // layer l: this group's outputs published
            if (s_failed) break;
        }
        if (s_failed) { alive = false; break; }
        // ---- result: group 0 -> mailbox, payload first, completion word last ----
        if (g == 0) {
            const unsigned long long* md_out = a.acts + (size_t)a.n_layers * kSrvActStride;
            const unsigned long long* te_out = a.acts + (size_t)a.n_te * kSrvActStride;
            const unsigned tag_md = tag0 + (unsigned)a.n_layers, tag_te = tag0 + (unsigned)a.n_te;
            const int n_out = decode_only ? a.Da : a.Da + 3 * a.Z;           // (decoder only: just the action)
            for (int i = tid; i < n_out; i += 256) {
                float v;
                if (i < a.Da) v = srv_get(md_out + i, tag_md, t_start, a.life_ticks, failed, a.sync + 18);
                else if (i < a.Da + 2 * a.Z) v = a.prior_kind == PVAE_PRIOR_NONE && i >= a.Da + a.Z ? 0.f
                                                 : srv_get(te_out + (i - a.Da), tag_te, t_start, a.life_ticks, failed, a.sync + 18);
                else {                                                       // z as the decoder saw it (same expression as above)
                    const int j = i - a.Da - 2 * a.Z;
                    if (a.prior_kind == PVAE_PRIOR_NONE) v = srv_get(te_out + j, tag_te, t_start, a.life_ticks, failed, a.sync + 18);
                    else {
                        const float e = noise ? philox_normal(seed, offset, 0, j) : 0.f;
                        float mu, lv;
                        srv_get2(te_out + j, te_out + a.Z + j, tag_te, mu, lv, t_start, a.life_ticks, failed, a.sync + 18);
                        v = mu + e * expf(0.5f * lv);
                    }
                }
                __hip_atomic_store(a.mb->out + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // (no read of host memory on this path, and no release fence -- it would write the whole L2 back, twice: the payload
            //  went out as system-scope stores that are not cached, the wait above saw them acknowledged, and posted writes of
            //  one agent arrive in order)
            if (tid == 0) {
                __hip_atomic_store(&a.mb->done_seq, last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                a.dbg[1 + 2 * a.n_layers] = wall_clock64();                  // completion word issued
                a.dbg[2 + 2 * a.n_layers] = (unsigned long long)a.n_layers;
                a.dbg[63] = (unsigned long long)clock64();
            }
        }
    }
    if (g == 0 && tid == 0) {
        __hip_atomic_store(a.sync + 18, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);       // "group 0 has left" (see srv_get)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __hip_atomic_store(&a.mb->state, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct RolloutServer {
    SrvReply* mb = nullptr;               // hipHostMalloc (mapped)
    SrvReply* mb_dev = nullptr;
    SrvRequest* req = nullptr;            // host view of the request block (device mode: the device pointer itself, written through the BAR)
    SrvRequest* req_dev = nullptr;
    bool req_on_device = false;
    unsigned* sync = nullptr;             // device
    unsigned long long* dbg = nullptr;    // device: group 0's stamps of the last request
    unsigned long long* acts = nullptr;   // device: tagged hand-over words
    hipStream_t stream = nullptr;
    SrvArgs args{};
    size_t lds_bytes = 0;
    uint32_t seq = 0, served = 0;
    unsigned long long loaded_version = 0;   // pvae_ctx::param_version of the resident weights
    int scope = 0, xcd = -1;
    bool launched = false;
    double idle_ms = 100.0, life_s = 600.0;
};

// LDS bytes per workgroup when every layer's output features are dealt out over `groups` workgroups (0: a layer or the
// observation is wider than the server takes); fills S.args.layer / counts
static size_t server_layout(pvae_ctx* c, RolloutServer& S, int groups) {
    const NetLayout& TE = c->L.net[PVAE_NET_TE];
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    SrvArgs& a = S.args;
    memset(&a, 0, sizeof(a));
    int off = 0, max_ld = 0, i = 0;
    for (const NetLayout* N : {&TE, &MD})
        for (const Layer& l : N->layers) {
            SrvLayer& L = a.layer[i++];
            L.w_off = l.w_off; L.b_off = l.b_off; L.ld = l.ld; L.n_out_pad = l.n_out_pad; L.n_out = l.n_out; L.act = l.act;
            L.F = (l.n_out_pad + groups - 1) / groups;
            if (l.n_out_pad > kSrvActStride || l.ld > kSrvActStride) return 0;
            L.lds_off = off;
            off += L.F * l.ld + ((L.F + 3) & ~3);                         // rows + biases (16-byte granules)
            if (l.ld > max_ld) max_ld = l.ld;
        }
    a.n_layers = i; a.n_te = (int)TE.layers.size();
    a.groups = groups; a.one_xcd = groups == 32 ? 1 : 0;
    a.Db = c->L.cfg.dim_body; a.Da = c->L.cfg.dim_action; a.Z = c->L.cfg.latent; a.prior_kind = c->L.cfg.prior_kind;
    a.xs_off = off;
    return (size_t)(off + max_ld) * sizeof(float);
}

// scope: 0 = one XCD if the stacks fit its CUs' LDS, else the whole chip; 1 = one XCD; 2 = the whole chip
static int server_plan(pvae_ctx* c, RolloutServer& S, int scope) {
    const int n = (int)(c->L.net[PVAE_NET_TE].layers.size() + c->L.net[PVAE_NET_MD].layers.size());
    if (n > kSrvMaxLayers) return fail(-24, "rollout server: %d layers (at most %d)", n, kSrvMaxLayers);
    if (c->L.cfg.prior_kind == PVAE_PRIOR_HYPERSPHERE)
        return fail(-24, "rollout server: this latent prior is served by the per-layer launches only");
    if (2 * c->L.cfg.dim_body > kSrvMaxObs || c->L.cfg.dim_action + 3 * c->L.cfg.latent > kSrvMaxOut)
        return fail(-24, "rollout server: observation / action too wide");
    constexpr size_t kFit = 156 * 1024;
    size_t need = 0;
    for (int groups : {32, 256}) {
        if ((groups == 32 && scope == 2) || (groups == 256 && scope == 1)) continue;
        need = server_layout(c, S, groups);
        if (need == 0) return fail(-24, "rollout server: a layer wider than %d", kSrvActStride);
        if (need <= kFit) { S.lds_bytes = need; return 0; }
    }
    return fail(-24, "rollout server: the encoder's and decoder's weights need %zu KB of LDS per workgroup even when dealt out over "
                     "%s, more than a CU has: these stacks are served by the per-layer launches", need / 1024,
                scope == 1 ? "the 32 CUs of one XCD" : "all 256 CUs");
}

static int server_launch(pvae_ctx* c, RolloutServer& S) {
    HIP_TRY(hipMemsetAsync(S.sync, 0, 64 * sizeof(unsigned), S.stream));
    S.mb->state = 0; S.mb->done_seq = S.seq;
    S.req->cmd = 0; S.req->req_seq = S.seq;
    __builtin_ia32_sfence();                                    // (device-resident request block: write-combined stores)
    HIP_TRY(params_settle(c));                                   // (the launch reads the parameters as they are NOW)
    S.loaded_version = c->param_version;
    S.args.seq0 = S.seq;
    S.args.params = c->params;
    S.args.idle_ticks = (long long)(S.idle_ms * 1e5);
    S.args.life_ticks = (long long)(S.life_s * 1e8);
    HIP_TRY(hipFuncSetAttribute((const void*)rollout_server_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes));
    hipLaunchKernelGGL(rollout_server_kernel, dim3(256), dim3(256), S.lds_bytes, S.stream, S.args);
    HIP_TRY(hipGetLastError());
    S.launched = true;
    // until the kernel reports "serving" (or refuses): bounded
    const auto t0 = std::chrono::steady_clock::now();
    while (S.mb->state == 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0)
            return fail(-25, "rollout server: the kernel did not come up within 5 s");
        std::this_thread::yield();
    }
    if (S.mb->state == 3) {
        HIP_TRY(hipStreamSynchronize(S.stream));
        S.launched = false;
        return fail(-24, "rollout server: its 32 workgroups were not placed on one XCD; use scope 2 (whole chip) or the per-layer launches");
    }
    return 0;
}

extern "C" {
/* see include/pvae.h */
int pvae_rollout_server_start(pvae_ctx* c, double idle_timeout_ms, double lifetime_s, int scope) {
    int rc = check_ready(c, true);
    if (rc) return rc;
    if (!c->server) c->server = new RolloutServer();
    RolloutServer& S = *c->server;
    if (S.launched && S.mb && S.mb->state == 1) return 0;                 // already serving
    if (scope < 0) scope = S.scope;                                       // (a relaunch keeps what the caller chose)
    if (scope < 0 || scope > 2) return fail(-1, "scope %d: 0 auto, 1 one XCD, 2 the whole chip", scope);
    S.scope = scope;
    if ((rc = server_plan(c, S, scope))) return rc;
    if (!S.mb) {
        HIP_TRY(hipHostMalloc((void**)&S.mb, sizeof(SrvReply), hipHostMallocMapped));
        memset((void*)S.mb, 0, sizeof(SrvReply));
        HIP_TRY(hipHostGetDevicePointer((void**)&S.mb_dev, (void*)S.mb, 0));
        // The request block: with a large BAR the host reaches device memory through the pointer itself (tools/bar_probe.py),
        // so the block lives in UNCACHED device memory -- the host pushes observation + request word, the kernel polls and
        // reads local memory.  Otherwise (or option "server_mailbox" = 1) pinned host memory that the kernel pulls from.
        int dev = 0, large_bar = 0;
        HIP_TRY(hipGetDevice(&dev));
        (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev);
        S.req_on_device = large_bar != 0 && c->server_mailbox != 1;
        if (c->server_mailbox == 2) S.req_on_device = true;
        if (S.req_on_device) {
            HIP_TRY(hipExtMallocWithFlags((void**)&S.req_dev, sizeof(SrvRequest), hipDeviceMallocUncached));
            HIP_TRY(hipMemset(S.req_dev, 0, sizeof(SrvRequest)));
            HIP_TRY(hipDeviceSynchronize());
            S.req = S.req_dev;
        } else {
            HIP_TRY(hipHostMalloc((void**)&S.req, sizeof(SrvRequest), hipHostMallocMapped));
            memset((void*)S.req, 0, sizeof(SrvRequest));
            HIP_TRY(hipHostGetDevicePointer((void**)&S.req_dev, (void*)S.req, 0));
        }
        HIP_TRY(hipMalloc((void**)&S.sync, 64 * sizeof(unsigned)));
        HIP_TRY(hipMalloc((void**)&S.dbg, 64 * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(S.dbg, 0, 64 * sizeof(unsigned long long)));
        HIP_TRY(hipMalloc((void**)&S.acts, (size_t)(kSrvMaxLayers + 1) * kSrvActStride * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(S.acts, 0, (size_t)(kSrvMaxLayers + 1) * kSrvActStride * sizeof(unsigned long long)));
        int lo = 0, hi = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));               // lo: least urgent.  A priority of its own = a hardware
        HIP_TRY(hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, lo));   // queue no compute stream is mapped onto
    }
    if (S.launched) { HIP_TRY(hipStreamSynchronize(S.stream)); S.launched = false; }   // an instance that gave up (idle): reap it
    if (idle_timeout_ms > 0) S.idle_ms = idle_timeout_ms;
    if (lifetime_s > 0) S.life_s = lifetime_s;
    S.args.mb = S.mb_dev; S.args.req = S.req_dev; S.args.obs_direct = S.req_on_device ? 1 : 0;
    S.args.sync = S.sync; S.args.acts = S.acts; S.args.dbg = S.dbg;
    // (every server of this process on an XCD of its own: two engines can serve side by side)
    static int next_xcd = 0;
    if (S.xcd < 0) S.xcd = next_xcd++ & 7;
    S.args.xcd = S.xcd;
    return server_launch(c, S);
}

static int server_request(pvae_ctx* c, uint32_t cmd, const float* obs, int noise, uint64_t seed, uint64_t offset, double timeout_ms) {
    RolloutServer& S = *c->server;
    SrvReply* mb = S.mb;
    SrvRequest* rq = S.req;
    if (obs) memcpy((void*)rq->obs, obs, (size_t)(cmd == 3u ? S.args.Db + S.args.Z : 2 * S.args.Db) * sizeof(float));
    rq->cmd = cmd; rq->noise = noise ? 1u : 0u;
    rq->seed_lo = (uint32_t)seed; rq->seed_hi = (uint32_t)(seed >> 32); rq->off_lo = (uint32_t)offset; rq->off_hi = (uint32_t)(offset >> 32);
    const uint32_t seq = ++S.seq;
    // the request word goes LAST: behind a store fence when the block is device memory (write-combined stores through the
    // BAR may leave the core out of order; posted PCIe writes then arrive in the order they left)
    if (S.req_on_device) __builtin_ia32_sfence();
    __atomic_store_n(&rq->req_seq, seq, __ATOMIC_RELEASE);
    if (S.req_on_device) __builtin_ia32_sfence();
    if (cmd == 1) return 0;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&mb->done_seq, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 1023u) == 0) {
            if (__atomic_load_n(&mb->state, __ATOMIC_ACQUIRE) != 1u) return 1;           // the kernel left (idle time-out raced the request)
            if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
                return fail(-25, "rollout server: no answer within %.1f ms", timeout_ms);
        }
    }
    return 0;
}

int pvae_rollout_server_infer(pvae_ctx* c, const float* obs, int noise, uint64_t rng_seed, uint64_t rng_offset, int reload,
                              float* a_hat, float* mu_logvar, float* z, double timeout_ms) {
    if (!c || !c->server || !c->server->mb) return fail(-2, "rollout server not started (pvae_rollout_server_start)");
    if (!obs || !a_hat) return fail(-1, "obs / a_hat is null");
    RolloutServer& S = *c->server;
    if (timeout_ms <= 0) timeout_ms = 1000.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (!S.launched || S.mb->state != 1u) {                  // it left after its idle time: bring it back (weights re-read)
            int rc = pvae_rollout_server_start(c, 0, 0, -1);
            if (rc) return rc;
            reload = 0;
        }
        if (S.loaded_version != c->param_version) {              // optimizer steps went through this library since: re-read
            HIP_TRY(params_settle(c));
            S.loaded_version = c->param_version;
            reload = 1;
        }
        const int r = server_request(c, reload ? 2u : 0u, obs, noise, rng_seed, rng_offset, timeout_ms);
        if (r < 0) return r;
        if (r == 0) {
            ++S.served;
            const int Da = S.args.Da, Z = S.args.Z;
            memcpy(a_hat, (const void*)S.mb->out, (size_t)Da * sizeof(float));
            if (mu_logvar) memcpy(mu_logvar, (const void*)(S.mb->out + Da), (size_t)2 * Z * sizeof(float));
            if (z) memcpy(z, (const void*)(S.mb->out + Da + 2 * Z), (size_t)Z * sizeof(float));
            return 0;
        }
    }
    return fail(-25, "rollout server: the kernel left twice while a request was pending");
}

/* forward_decoder at B = 1 ("pass_through" rollouts, rllib_env_imitation.py:233-258: z drawn by the caller): s1_z = [s1 (Db) | z (Z)]
 * -> a_hat[Da], the same bits as pvae_net_forward(PVAE_NET_MD) on that row.  The encoder's layers are skipped. */
int pvae_rollout_server_decode(pvae_ctx* c, const float* s1_z, float* a_hat, double timeout_ms) {
    if (!c || !c->server || !c->server->mb) return fail(-2, "rollout server not started (pvae_rollout_server_start)");
    if (!s1_z || !a_hat) return fail(-1, "s1_z / a_hat is null");
    RolloutServer& S = *c->server;
    if (timeout_ms <= 0) timeout_ms = 1000.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (S.launched && S.mb->state == 1u && S.loaded_version != c->param_version) {
            int rc = pvae_rollout_server_stop(c);                // (no reload form of this request: a relaunch re-reads)
            if (rc) return rc;
        }
        if (!S.launched || S.mb->state != 1u) {
            int rc = pvae_rollout_server_start(c, 0, 0, -1);
            if (rc) return rc;
        }
        const int r = server_request(c, 3u, s1_z, 0, 0, 0, timeout_ms);
        if (r < 0) return r;
        if (r == 0) {
            ++S.served;
            memcpy(a_hat, (const void*)S.mb->out, (size_t)S.args.Da * sizeof(float));
            return 0;
        }
    }
    return fail(-25, "rollout server: the kernel left twice while a request was pending");
}

/* n requests back to back with the SAME observation, each timed on the host clock inside this call (what a compiled host
 * sees; a Python caller adds its own call overhead): us[i] = host observation -> host action of request i. */
int pvae_rollout_server_selfbench(pvae_ctx* c, const float* obs, int noise, int32_t n, double* us) {
    if (!c || !c->server || !c->server->mb) return fail(-2, "rollout server not started (pvae_rollout_server_start)");
    if (!obs || !us || n < 1) return fail(-1, "bad arguments");
    std::vector<float> a(c->server->args.Da);
    for (int i = 0; i < n; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        const int rc = pvae_rollout_server_infer(c, obs, noise, 1, (uint64_t)i, 0, a.data(), nullptr, nullptr, 1000.0);
        us[i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (rc) return rc;
    }
    return 0;
}

/* Where the last request's time went on the device: us[0] = 0 (request seen by workgroup 0), us[1 + 2 l] = layer l's inputs
 * in LDS, us[2 + 2 l] = layer l's outputs published, us[1 + 2 n_layers] = completion word issued; *n = entries written. */
int pvae_rollout_server_timeline(pvae_ctx* c, double* us, int32_t max, int32_t* n) {
    if (!c || !c->server || !c->server->dbg) return fail(-2, "rollout server not started (pvae_rollout_server_start)");
    unsigned long long t[64];
    HIP_TRY(hipMemcpy(t, c->server->dbg, sizeof(t), hipMemcpyDeviceToHost));
    const int cnt = 2 + 2 * c->server->args.n_layers;
    int m = 0;
    for (; m < cnt && m < max; ++m) us[m] = (double)(long long)(t[m] - t[0]) / 100.0;
    // last entry: the shader clock during the request, MHz (s_memtime ticks per microsecond of the 100 MHz wall clock)
    if (m < max && cnt >= 2 && t[cnt - 1] > t[0]) us[m++] = (double)(long long)(t[63] - t[62]) / ((double)(long long)(t[cnt - 1] - t[0]) / 100.0);
#ifdef PVAE_SRV_FINE
    for (int k = 39; k <= 42 && m < max; ++k) us[m++] = (double)(long long)(t[k] - t[0]) / 100.0;
#endif
    if (n) *n = m;
    return 0;
}

int pvae_rollout_server_stop(pvae_ctx* c) {
    if (!c) return fail(-1, "null ctx");
    if (!c->server || !c->server->mb) return 0;
    RolloutServer& S = *c->server;
    if (S.launched) {
        if (S.mb->state == 1u) server_request(c, 1u, nullptr, 0, 0, 0, 0);
        HIP_TRY(hipStreamSynchronize(S.stream));                 // bounded: stop command, else idle time-out, else lifetime
        S.launched = false;
    }
    return 0;
}

int pvae_params_changed(pvae_ctx* c, void* stream) {
    if (!c) return fail(-1, "null ctx");
    params_touched(c, (hipStream_t)stream);
    return 0;
}

int pvae_rollout_server_status(pvae_ctx* c, int32_t* serving, uint32_t* served, int32_t* lds_bytes) {
    if (!c) return fail(-1, "null ctx");
    const RolloutServer* S = c->server;
    if (serving) *serving = (S && S->mb && S->launched && S->mb->state == 1u) ? (S->req_on_device ? 2 : 1) : 0;   // 2: request block in device memory
    if (served) *served = S ? S->served : 0u;
    if (lds_bytes) *lds_bytes = S ? (int32_t)S->lds_bytes * (S->args.one_xcd ? 1 : -1) : 0;   // (negative: dealt out over the whole chip)
    return 0;
}
}   // extern "C"

static void server_free(pvae_ctx* c) {
    if (!c->server) return;
    (void)pvae_rollout_server_stop(c);
    RolloutServer& S = *c->server;
    if (S.stream) (void)hipStreamDestroy(S.stream);
    if (S.sync) (void)hipFree(S.sync);
    if (S.dbg) (void)hipFree(S.dbg);
    if (S.acts) (void)hipFree(S.acts);
    if (S.mb) (void)hipHostFree((void*)S.mb);
    if (S.req) { if (S.req_on_device) (void)hipFree((void*)S.req); else (void)hipHostFree((void*)S.req); }
    delete c->server;
    c->server = nullptr;
}

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
                       rows_pad);
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

// Shader clock sustained while every SIMD issues fp32 MFMAs back to back on the caller's operands (DVFS:
// the chip clocks to its power budget, and MFMA power depends on how much the operands toggle -- zeros
// run at the 2.4 GHz spec clock, real weights ~10 % lower).  One workgroup reports shader cycles
// (s_memtime) against the 100 MHz wall clock.
__global__ void __launch_bounds__(256)
mfma_clock_kernel(const float* __restrict__ src, int n_src, float* __restrict__ sink, int n, unsigned long long* out) {
    // eight different operand pairs per lane, cycled: consecutive MFMAs see different values, as in a real
    // contraction (with ONE constant pair the multiplier array hardly switches and the probe reads high)
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        a[u] = src[(threadIdx.x + 256 * blockIdx.x + 4099 * u) % n_src];
        b[u] = src[(7919 + threadIdx.x + 17 * blockIdx.x + 6151 * u) % n_src];
    }
    v4f acc[4] = {v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}, v4f{0.f, 0.f, 0.f, 0.f}};
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; i += 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[u & 3], 0, 0, 0);
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    const v4f s4 = (acc[0] + acc[1]) + (acc[2] + acc[3]);
    if (s4[0] == 123.456f) sink[threadIdx.x] = s4[1];              // keeps the MFMAs alive; never true in practice
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

int pvae_mfma_clock_probe(const float* operands, int64_t n_operands, float* scratch, double* ghz, double* tflops_peak,
                          void* stream) {
    if (!operands || n_operands < 8192 || !scratch || !ghz || !tflops_peak) return fail(-1, "bad probe arguments");
    hipStream_t st = (hipStream_t)stream;
    int cus = 0, dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(scratch);      // 16 bytes, then the sink
    const int n_src = (int)(n_operands > (1 << 30) ? (1 << 30) : n_operands);
    // the power-management loop reacts over milliseconds: ~10 ms of this load before the launch that is read
    // (two launches still report the 2.38 GHz the chip starts at; after 2 ms it has settled near 2.17)
    for (int rep = 0; rep < 30; ++rep)
        hipLaunchKernelGGL(mfma_clock_kernel, dim3(4 * cus), dim3(256), 0, st, operands, n_src, scratch + 64, 2048, d_out);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    unsigned long long h[2] = {0, 0};
    HIP_TRY(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    if (!h[1]) return fail(-10, "clock probe measured nothing");
    *ghz = (double)h[0] / ((double)h[1] * 10.0);                    // cycles / ns
    *tflops_peak = (double)cus * 256.0 * *ghz * 1e9 / 1e12;         // 256 FLOP / clk / CU (MI355X_MICROARCH.md)
    return 0;
}

int pvae_profile_enable(int on) {
    g_prof.on = on != 0;
    if (on) g_prof.n = 0;
    return 0;
}

int pvae_profile_read(int category, double* total_ms, int64_t* launches, double* total_flops) {
    if (!total_ms || !launches || !total_flops) return fail(-1, "null output");
    double ms = 0, fl = 0;
    int64_t cnt = 0;
    for (int i = 0; i < g_prof.n; ++i) {
        if (g_prof.cat[i] != category) continue;
        HIP_TRY(hipEventSynchronize(g_prof.ev[i][1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, g_prof.ev[i][0], g_prof.ev[i][1]));
        ms += t; fl += g_prof.flops[i]; ++cnt;
    }
    *total_ms = ms; *launches = cnt; *total_flops = fl;
    return 0;
}

int pvae_gemm_probe(int kind, const float* a, int lda, const float* b, int ldb, float* cc, int ldc,
                    const float* bias_or_mask, int ld_mask, int m, int n, int k, int relu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!a || !b || !cc) return fail(-1, "null operand");
    if (kind == 0) {
        if (m % 32 || n % 32 || k % 64) return fail(-1, "forward probe needs M%%32==0, N%%32==0, K%%64==0");
        HIP_TRY(gemm_forward(a, lda, b, ldb, bias_or_mask, cc, ldc, m, n, k, relu, st));
    } else if (kind == 1) {
        if (m % 32 || k % 32 || n % 64) return fail(-1, "dgrad probe needs M%%32==0, K%%32==0, N%%64==0");
        HIP_TRY(gemm_dgrad(a, lda, b, ldb, bias_or_mask, ld_mask, cc, ldc, m, k, n, st));
    } else if (kind == 2) {
        if (m % 32 || n % 64 || k % 64) return fail(-1, "wgrad probe needs M%%32==0, N%%64==0, K%%64==0");
        EpiGradStore e{cc, ldc};
        HIP_TRY(gemm_wgrad(a, lda, b, ldb, n, k, m, e, st));
    } else {
        return fail(-1, "unknown probe kind %d", kind);
    }
    return 0;
}

}  // extern "C"
