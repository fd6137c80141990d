// pvae_internal.h -- what the translation units of libpvae_gfx950.so share: error handling, the optional per-launch
// profiler, the run-time RCCL binding, the context, Philox, and the few functions one unit calls in another.
//   pvae.hip                 the training step (glue kernels, forward / backward plans, the C ABI of the step and of rollout launches)
//   pvae_exchange.hip        data-parallel exchange: RCCL calls, the peer-mapped exchange kernels, their set-up and self-test
//   pvae_rollout_server.hip  the call-persistent rollout server
//   pvae_probe.hip           measurement entry points (clock probe, profiler read-out, contraction probe)
#pragma once
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
inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
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
inline Profiler g_prof;

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
inline Rccl g_rccl;
enum { kNcclSum = 0, kNcclFloat32 = 7 };

inline int rccl_load() {
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
    // while the gathered layer-0 weight gradients assemble every chunk of X from two unaligned loads and a select in the
    // launch that also carries the Adam epilogue: joint 250.0 vs 240.3 us, world 92.2 vs 86.6 (docs/experiments.md, round 5).
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
    hipStream_t param_stream = nullptr;                  // (NULL is a stream too: the default one.  The caller's stream must
                                                         //  outlive the writes it queued, as for any other call here.)
    bool param_pending = false;                          // work that writes the parameters may still be queued on it
};
static inline void params_touched(pvae_ctx* c, hipStream_t st, bool queued = true) {
    // one stream is remembered: writes still queued on ANOTHER one are drained here (a caller that switches streams with
    // parameter writes in flight -- rare, and it costs that caller one wait -- instead of a server that reloads too early)
    if (c->param_pending && c->param_stream != st) (void)hipStreamSynchronize(c->param_stream);
    ++c->param_version; c->param_stream = st; c->param_pending = queued;
}
static inline hipError_t params_settle(pvae_ctx* c) {
    if (!c->param_pending) return hipSuccess;
    c->param_pending = false;
    return hipStreamSynchronize(c->param_stream);
}
void server_free(pvae_ctx* c);                 // pvae_rollout_server.hip


// ---------------------------------------------------------------------------------------
// Philox (the sampler of the training step, the rollout launches and the rollout server draw the same stream)
// ---------------------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------
inline AdamScalars adam_scalars(const pvae_step_params* sp, int net) {
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

inline int check_ready(const pvae_ctx* c, bool need_arenas) {
    if (!c) return fail(-1, "null ctx");
    if (!c->ws) return fail(-2, "workspace not bound");
    if (need_arenas && !c->params) return fail(-2, "parameter arena not bound");
    return 0;
}


// ---- pvae_exchange.hip, called by the data-parallel step in pvae.hip ----
struct Bucket { int64_t off, cnt; };
int64_t auto_bucket_bytes(const pvae_ctx* c, int phase);
std::vector<Bucket> exchange_buckets(const pvae_ctx* c, int net);
int exchange_bucket(pvae_ctx* c, int net, const Bucket& b, const pvae_step_params* sp, hipStream_t st, hipStream_t cs, int& n_events);
