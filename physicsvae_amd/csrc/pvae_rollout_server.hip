// pvae_rollout_server.hip -- the call-persistent rollout server (rmt:742-771 at B = 1).  gfx950 only.
#include "pvae_internal.h"
#include <atomic>

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
constexpr int kSrvMaxLayers = 16, kSrvMaxObs = 4096, kSrvMaxOut = 2048;
constexpr int kSrvActStride = 2048;
// words of one hand-over slot of an instance that takes `rmax` rows per request: rmax rows of kSrvActStride words -- plus, for
// the multi-row instance, an odd number of 128-byte lines, so that consecutive slots do not start 64 KB apart (slots 64 KB
// apart cost the SINGLE-row kernel 0.8 us per request when it used this layout: profiles/r05_ab_server_b1.txt)
__host__ __device__ constexpr size_t srv_slot_words(int rmax) { return (size_t)rmax * kSrvActStride + (rmax > 1 ? 1040 : 0); }
constexpr int kSrvMaxRows = 4;              // rows per request (rmt:742-771 serves any batch; the control loop's is 1)
struct SrvRequest {                       // host -> device.  Lives in DEVICE memory when the host can write it directly
    // (large BAR: the host PUSHES the observation and the kernel polls local memory), else in pinned host memory (the
    // kernel PULLS over PCIe).  ONE 32-byte line of control words, then the observation.
    volatile uint32_t req_seq;            // written LAST by the host: request number
    uint32_t cmd;                         // low byte: 0 infer, 1 stop, 2 reload the weights from the arena, then infer,
                                          // 3 decoder only ("pass_through", rllib_env_imitation.py:233-258): obs = [s1 (Db) | z (Z)]
                                          // bits 8-9: rows - 1 of this request (obs = rows x [...], densely packed)
    uint32_t noise, check;                // check: srv_check() of the other seven words -- the kernel takes a line only when it
                                          // matches, so a read of the line that saw req_seq but an older cmd / seed is re-polled
    uint32_t seed_lo, seed_hi, off_lo, off_hi;
    uint32_t pad1[8];
    float obs[kSrvMaxObs];
};
__host__ __device__ inline uint32_t srv_check(uint32_t seq, uint32_t cmd, uint32_t noise, uint32_t s0, uint32_t s1, uint32_t o0, uint32_t o1) {
    return 0x5eedc0deu ^ seq ^ (cmd * 0x9e3779b1u) ^ (noise << 7) ^ s0 ^ (s1 * 3u) ^ (o0 * 5u) ^ (o1 * 7u);
}
struct SrvReply {                         // device -> host, pinned host memory (the host spins on its own RAM)
    volatile uint32_t done_seq;           // written LAST by the device: the request this result belongs to
    volatile uint32_t state;              // 0 not started, 1 serving, 2 exited (idle / stop / lifetime), 3 refused (placement)
    uint32_t served, pad2[13];
    float out[kSrvMaxOut];                // [a_hat (Da) | mu (Z) | logvar (Z) | z (Z)]
};
struct SrvLayer { long long w_off, b_off; int ld, n_out_pad, n_out, act, F, lds_off; };   // F: features per group (the last
                                                                                          // active group may own fewer)
struct SrvArgs {
    SrvLayer layer[kSrvMaxLayers];
    int n_layers, n_te;                   // layers [0, n_te) are the encoder's, [n_te, n_md) the decoder's,
    int n_md;                             // [n_md, n_layers) the motor decoder's helper's (rmt:670-680; none: n_md == n_layers)
    float mh_range;                       // a_hat = decoder + mh_range * helper (rmt:833-835)
    int groups, one_xcd;                  // 32 workgroups on ONE XCD, or 256 over the whole chip (stacks too big for one XCD's LDS)
    int xcd;                              // which XCD (one_xcd): servers of one process take different ones
    int Db, Da, Z, prior_kind;
    const float* params;
    unsigned long long* acts;             // [n_layers + 1][RMAX of the instance][kSrvActStride] TAGGED values: slot 0 = the observation, slot l + 1 = layer l's output
    unsigned* sync;                       // device words: 0 start-up barrier, 1 go_seq, 2..8 the request's control words, 16 xcc of group 0, 17 error, 18 group 0 has left
    SrvRequest* req;                      // device view of the request block
    SrvReply* mb;                         // device view of the reply block
    int obs_direct;                       // the request block is device memory: every group reads the observation from it
    long long idle_ticks, life_ticks;     // 100 MHz wall clock
    int xs_off, xs_ld;                    // float offset of the input vectors inside the dynamic LDS, floats per row (kSrvMaxRows of them)
    unsigned seq0;                        // requests served by earlier instances (this one answers seq0 + 1, ...)
    unsigned long long* dbg;              // [64] wall-clock stamps of group 0 for the LAST request (pvae_rollout_server_timeline)
};
// the control line as eight lanes read it (lane i: word i): whole iff word 3 is srv_check() of the other seven
__device__ inline bool srv_line_whole(unsigned w) {
    return srv_check(__builtin_amdgcn_readlane(w, 0), __builtin_amdgcn_readlane(w, 1), __builtin_amdgcn_readlane(w, 2),
                     __builtin_amdgcn_readlane(w, 4), __builtin_amdgcn_readlane(w, 5), __builtin_amdgcn_readlane(w, 6),
                     __builtin_amdgcn_readlane(w, 7)) == (unsigned)__builtin_amdgcn_readlane(w, 3);
}
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

// The same for the rows of a multi-row request at once (row r of the slot: prev + r * kSrvActStride -> xs + r * xs_ld): every
// row's words are in flight together, so a layer's inputs cost ONE round trip after the producers' stores instead of one per row.
template <int R>
__device__ inline void srv_get_rows(float* xs, int xs_ld, const unsigned long long* prev, int rows, int n, int ld, unsigned tag, int tid,
                                    long long t_start, long long life, int& failed, const unsigned* gone) {
    constexpr int kMax = kSrvActStride / 256;
    unsigned long long u[R][kMax];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < kMax; ++i) {
                const int k = tid + 256 * i;
                u[r][i] = (r < rows && k < n) ? __hip_atomic_load(prev + (size_t)r * kSrvActStride + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                              : ((unsigned long long)tag << 32);
            }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < kMax; ++i) all = all && (unsigned)(u[r][i] >> 32) == tag;
        if (all) break;
        if ((++spins & 255u) == 0 && (srv_ldu(gone) != 0u || wall_clock64() - t_start > life)) { failed = 1; break; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < kMax; ++i) {
            const int k = tid + 256 * i;
            if (r < rows && k < ld) xs[r * xs_ld + k] = k < n ? __uint_as_float((unsigned)u[r][i]) : 0.f;
        }
}
// (mu, logvar) of latent j for every row of a multi-row request, polled together
template <int R>
__device__ inline void srv_get2_rows(const unsigned long long* p0, const unsigned long long* p1, int rows, unsigned tag, float (&v0)[R],
                                     float (&v1)[R], long long t_start, long long life, int& failed, const unsigned* gone) {
    unsigned long long u0[R], u1[R];
    unsigned spins = 0;
    for (;;) {
        bool all = true;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            u0[r] = r < rows ? __hip_atomic_load(p0 + (size_t)r * kSrvActStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
            u1[r] = r < rows ? __hip_atomic_load(p1 + (size_t)r * kSrvActStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)tag << 32);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) all = all && (unsigned)(u0[r] >> 32) == tag && (unsigned)(u1[r] >> 32) == tag;
        if (all) break;
        if ((++spins & 255u) == 0 && (srv_ldu(gone) != 0u || wall_clock64() - t_start > life)) { failed = 1; break; }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) { v0[r] = __uint_as_float((unsigned)u0[r]); v1[r] = __uint_as_float((unsigned)u1[r]); }
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

// RMAX: the most rows a request to this instance may carry.  The instance a server starts with is RMAX = 1 -- the control
// loop's kernel, nothing of the multi-row code in it (15.2 us at the default stacks; with the 4-row bodies compiled into the
// same function the register allocator spilled twice as many SGPRs in the layer loop: 17.5 us) --; the first request with
// more than one row replaces it by the RMAX = 4 instance (pvae_rollout_server_infer_rows), which serves 1-4 rows from then on.
// HELPER: the stacks include the motor decoder's helper (layers [n_md, n_layers)); without one that code is not in the instance.
template <int RMAX, bool HELPER>
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
                    if (lane < 8) w = __hip_atomic_load(line + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // eight lanes, one 32-byte line
                    seq = __builtin_amdgcn_readlane(w, 0);
                    if (seq != last && srv_line_whole(w)) { cmd = __builtin_amdgcn_readlane(w, 1); break; }
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
            if ((s_word[1] & 0xffu) != 1u && !a.obs_direct) {   // the observation(s): pinned host memory -> slot 0, tagged
                const unsigned tag0 = s_word[0] * 16u;
                const int n = (s_word[1] & 0xffu) == 3u ? a.Db + a.Z : 2 * a.Db, nr = RMAX == 1 ? 1 : (int)((s_word[1] >> 8) & 3u) + 1;
                for (int r = 0; r < nr; ++r)
                    for (int i = tid; i < n; i += 256)
                        srv_put(a.acts + (size_t)r * kSrvActStride + i,
                                __hip_atomic_load(a.req->obs + r * n + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), tag0);
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
                    if (seq != last && srv_line_whole(w)) { cmd = __builtin_amdgcn_readlane(w, 1); break; }
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
        const unsigned cmd = s_word[1] & 0xffu;
        const int rows = RMAX == 1 ? 1 : (int)((s_word[1] >> 8) & 3u) + 1;  // rows of this request
        if (cmd == 1u) break;
        if (cmd == 2u) {
            // the arena was rewritten (Adam's stores from other XCDs, an SDMA copy) while this kernel was resident: no kernel
            // boundary has invalidated this XCD's L2 / this CU's L1 since, so do it here (off the served path's common case)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            load_weights();
        }
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
            // the helper's first layer reads what the decoder's first layer read: [s1 | z | 0], z from the ENCODER's output slot
            const bool dec_in = l == a.n_te || (HELPER && l == a.n_md);
            const int lp = dec_in ? a.n_te : l;
            const unsigned long long* prev0 = a.acts + (size_t)lp * srv_slot_words(RMAX);   // slot l: the previous layer's output (0: obs)
            const unsigned tagp = tag0 + (unsigned)lp;
            const unsigned tago = tag0 + (unsigned)l + 1u;                         // tag of THIS layer's outputs (slot l + 1)
            const int n_obs = decode_only ? a.Db + a.Z : 2 * a.Db;                // floats per row of the request block
            // more than one row: the rows' hand-over words are polled TOGETHER wherever a layer waits for its producers
            bool formed = false;
            if constexpr (RMAX > 1) {
                if (rows > 1 && l == 0 && a.obs_direct) {
                    // the observations sit in the request block (device memory): every row's loads in flight together
                    for (int idx = tid; idx < rows * L.ld; idx += 256) {
                        const int r = idx / L.ld, k = idx - r * L.ld;
                        xs[r * a.xs_ld + k] = k < 2 * a.Db ? __hip_atomic_load(a.req->obs + r * n_obs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
                    }
                    formed = true;
                } else if (rows > 1 && !dec_in) {
                    // (layer 0: slot 0 = the observations group 0 copied; else the previous layer's outputs)
                    const unsigned long long* src = l == 0 ? a.acts : prev0;
                    const int n_in = l == 0 ? 2 * a.Db : L.ld;
                    const unsigned tg = l == 0 ? tag0 : tagp;
                    if (rows == 2) srv_get_rows<2>(xs, a.xs_ld, src, rows, n_in, L.ld, tg, tid, t_start, a.life_ticks, failed, a.sync + 18);
                    else srv_get_rows<RMAX>(xs, a.xs_ld, src, rows, n_in, L.ld, tg, tid, t_start, a.life_ticks, failed, a.sync + 18);
                    formed = true;
                } else if (rows > 1 && dec_in && !decode_only && a.prior_kind != PVAE_PRIOR_NONE) {
                    for (int k = tid; k < L.ld; k += 256) {
                        if (k >= a.Db && k < a.Db + a.Z) {
                            const int j = k - a.Db;
                            float mu[RMAX], lv[RMAX];
                            srv_get2_rows<RMAX>(prev0 + j, prev0 + a.Z + j, rows, tagp, mu, lv, t_start, a.life_ticks, failed, a.sync + 18);
                            for (int r = 0; r < rows; ++r) {
                                const float e = noise ? philox_normal(seed, offset, r, j) : 0.f;
                                xs[r * a.xs_ld + k] = mu[r] + e * expf(0.5f * lv[r]);
                            }
                        } else {
                            for (int r = 0; r < rows; ++r) {
                                float v = 0.f;
                                if (k < a.Db) v = a.obs_direct ? __hip_atomic_load(a.req->obs + r * n_obs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                               : srv_get(a.acts + (size_t)r * kSrvActStride + k, tag0, t_start, a.life_ticks, failed, a.sync + 18);
                                xs[r * a.xs_ld + k] = v;
                            }
                        }
                    }
                    formed = true;
                }
            }
            for (int r = 0; r < (formed ? 0 : rows); ++r) {
                float* xr = xs + r * a.xs_ld;
                const unsigned long long* prev = prev0 + (size_t)r * kSrvActStride;
                const unsigned long long* obs_w = a.acts + (size_t)r * kSrvActStride;      // slot 0, row r
                const float* obs_r = a.req->obs + r * n_obs;
                if (l == 0) {                                                    // [s1 | s2 | 0]
                    if (a.obs_direct) {                                          // (complete before the request word)
                        for (int k = tid; k < L.ld; k += 256)
                            xr[k] = k < 2 * a.Db ? __hip_atomic_load(obs_r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
                    } else {
                        srv_get_row(xr, obs_w, 2 * a.Db, L.ld, tag0, tid, t_start, a.life_ticks, failed, a.sync + 18);
                    }
                } else if (dec_in && decode_only) {                              // [s1 | z | 0] as the caller sent it
                    if (a.obs_direct) {
                        for (int k = tid; k < L.ld; k += 256)
                            xr[k] = k < a.Db + a.Z ? __hip_atomic_load(obs_r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.f;
                    } else {
                        srv_get_row(xr, obs_w, a.Db + a.Z, L.ld, tag0, tid, t_start, a.life_ticks, failed, a.sync + 18);
                    }
                } else if (dec_in) {                                             // [s1 | z | 0], the sampler formed in place
                    for (int k = tid; k < L.ld; k += 256) {
                        float v = 0.f;
                        if (k < a.Db) v = a.obs_direct ? __hip_atomic_load(obs_r + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                                       : srv_get(obs_w + k, tag0, t_start, a.life_ticks, failed, a.sync + 18);
                        else if (k < a.Db + a.Z) {
                            const int j = k - a.Db;
                            if (a.prior_kind == PVAE_PRIOR_NONE) v = srv_get(prev + j, tagp, t_start, a.life_ticks, failed, a.sync + 18);
                            else {
                                const float e = noise ? philox_normal(seed, offset, r, j) : 0.f;   // (before the wait: off its path)
                                float mu, lv;
                                srv_get2(prev + j, prev + a.Z + j, tagp, mu, lv, t_start, a.life_ticks, failed, a.sync + 18);
                                v = mu + e * expf(0.5f * lv);
                            }
                        }
                        xr[k] = v;
                    }
                } else {
                    srv_get_row(xr, prev, L.ld, L.ld, tagp, tid, t_start, a.life_ticks, failed, a.sync + 18);
                }
            }
            if (failed) s_failed = 1;
            __syncthreads();
            if (stamp) a.dbg[1 + 2 * l] = wall_clock64();                    // layer l: inputs in LDS
            const float* Wl = srv_lds + L.lds_off;
            unsigned long long* outp = a.acts + (size_t)(l + 1) * srv_slot_words(RMAX);     // (+ r * kSrvActStride: row r)
            // one wave per feature, gemv_rollout_kernel's sum operation for operation -- four features of the wave at a time,
            // so that their reductions overlap, and the butterfly as register moves (srv_tree_sum) instead of six
            // ds_bpermute round trips per feature.  More than one row: every weight fragment read from LDS feeds all rows
            // (R = 2 or 4 accumulators per feature; a 3-row request runs the 4-row body and drops the last row).
            int nf = L.n_out_pad - g * L.F;
            nf = nf < 0 ? 0 : (nf > L.F ? L.F : nf);
            for (int f0 = wave; f0 < nf; f0 += 16) {
                const int cnt = (nf - f0 + 3) >> 2;                          // features f0, f0 + 4, ... of this wave in this pass
                auto body = [&](auto nfeat, auto nrows) {                    // (one unguarded body per count: the LDS reads of a
                    constexpr int N = decltype(nfeat)::value, R = decltype(nrows)::value;   // k-step are in flight together)
                    float acc[N][R];
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[i][r] = 0.f;
                    for (int k = lane * 4; k < L.ld; k += 256) {
                        v4f xv[R];
#pragma unroll
                        for (int r = 0; r < R; ++r) xv[r] = *reinterpret_cast<const v4f*>(xs + (R == 1 ? 0 : r * a.xs_ld) + k);
                        v4f wv[N];
#pragma unroll
                        for (int i = 0; i < N; ++i) wv[i] = *reinterpret_cast<const v4f*>(Wl + (f0 + 4 * i) * L.ld + k);
#pragma unroll
                        for (int i = 0; i < N; ++i)
#pragma unroll
                            for (int r = 0; r < R; ++r)
                                acc[i][r] = fmaf(wv[i].x, xv[r].x, fmaf(wv[i].y, xv[r].y, fmaf(wv[i].z, xv[r].z, fmaf(wv[i].w, xv[r].w, acc[i][r]))));
                    }
#ifdef PVAE_SRV_FINE
                    if (stamp && l == 4) a.dbg[40] = wall_clock64();
#endif
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[i][r] = srv_tree_sum(acc[i][r]);
#ifdef PVAE_SRV_FINE
                    if (stamp && l == 4) a.dbg[41] = wall_clock64();
#endif
                    // lane i + 4 r finishes feature i of row r (bias, activation, hand-over word): the epilogues run side by side
                    // instead of one after the other on lane 0 (0.6 us of a 1.3 us layer when they did)
                    float mine = 0.f;
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const float si = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(acc[i][r]), 0));
                            mine = lane == i + 4 * r ? si : mine;
                        }
                    if (lane < 4 * R && (lane & 3) < N && (lane >> 2) < rows) {
                        const int f = f0 + 4 * (lane & 3), n = g * L.F + f;
                        float v = mine + Wl[L.F * L.ld + f];
                        v = (L.act > 1 && n >= L.n_out) ? 0.f : act_apply(v, L.act);
                        srv_put(outp + (R == 1 ? 0 : (size_t)(lane >> 2) * kSrvActStride) + n, v, tago);
                    }
                };
#ifdef PVAE_SRV_FINE
                if (stamp && l == 4) a.dbg[39] = wall_clock64();
#endif
                auto feat = [&](auto nrows) {
                    if (cnt >= 4) body(std::integral_constant<int, 4>(), nrows);
                    else if (cnt == 3) body(std::integral_constant<int, 3>(), nrows);
                    else if (cnt == 2) body(std::integral_constant<int, 2>(), nrows);
                    else body(std::integral_constant<int, 1>(), nrows);
                };
                if constexpr (RMAX == 1) feat(std::integral_constant<int, 1>());
                else {
                    if (rows == 1) feat(std::integral_constant<int, 1>());
                    else if (rows == 2) feat(std::integral_constant<int, 2>());
                    else feat(std::integral_constant<int, 4>());
                }
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
            constexpr bool helper = HELPER;
            const unsigned tag_md = tag0 + (unsigned)a.n_md, tag_te = tag0 + (unsigned)a.n_te, tag_mh = tag0 + (unsigned)a.n_layers;
            const int n_out = decode_only ? a.Da : a.Da + 3 * a.Z;           // (decoder only: just the action) -- per row
            for (int ri = tid; ri < rows * n_out; ri += 256) {
                const int r = RMAX == 1 ? 0 : ri / n_out, i = ri - r * n_out;
                const unsigned long long* md_out = a.acts + (size_t)a.n_md * srv_slot_words(RMAX) + (size_t)r * kSrvActStride;
                const unsigned long long* te_out = a.acts + (size_t)a.n_te * srv_slot_words(RMAX) + (size_t)r * kSrvActStride;
                const unsigned long long* mh_out = a.acts + (size_t)a.n_layers * srv_slot_words(RMAX) + (size_t)r * kSrvActStride;
                float v;
                if (i < a.Da) {
                    v = srv_get(md_out + i, tag_md, t_start, a.life_ticks, failed, a.sync + 18);
                    if (helper)                                              // (helper_add_kernel's expression: same bits)
                        v = __fmaf_rn(a.mh_range, srv_get(mh_out + i, tag_mh, t_start, a.life_ticks, failed, a.sync + 18), v);
                } else if (i < a.Da + 2 * a.Z) v = a.prior_kind == PVAE_PRIOR_NONE && i >= a.Da + a.Z ? 0.f
                                                 : srv_get(te_out + (i - a.Da), tag_te, t_start, a.life_ticks, failed, a.sync + 18);
                else {                                                       // z as the decoder saw it (same expression as above)
                    const int j = i - a.Da - 2 * a.Z;
                    if (a.prior_kind == PVAE_PRIOR_NONE) v = srv_get(te_out + j, tag_te, t_start, a.life_ticks, failed, a.sync + 18);
                    else {
                        const float e = noise ? philox_normal(seed, offset, r, j) : 0.f;
                        float mu, lv;
                        srv_get2(te_out + j, te_out + a.Z + j, tag_te, mu, lv, t_start, a.life_ticks, failed, a.sync + 18);
                        v = mu + e * expf(0.5f * lv);
                    }
                }
                __hip_atomic_store(a.mb->out + ri, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    int max_rows = 1;                     // rows per request the resident instance takes (1, or kSrvMaxRows after the first multi-row request)
    bool launched = false;
    double idle_ms = 100.0, life_s = 600.0;
};

// LDS bytes per workgroup when every layer's output features are dealt out over `groups` workgroups (0: a layer or the
// observation is wider than the server takes); fills S.args.layer / counts
static size_t server_layout(pvae_ctx* c, RolloutServer& S, int groups) {
    const NetLayout& TE = c->L.net[PVAE_NET_TE];
    const NetLayout& MD = c->L.net[PVAE_NET_MD];
    const NetLayout& MH = c->L.net[PVAE_NET_MH];                         // (empty without a helper)
    SrvArgs& a = S.args;
    memset(&a, 0, sizeof(a));
    int off = 0, max_ld = 0, i = 0;
    for (const NetLayout* N : {&TE, &MD, &MH})
        for (const Layer& l : N->layers) {
            SrvLayer& L = a.layer[i++];
            L.w_off = l.w_off; L.b_off = l.b_off; L.ld = l.ld; L.n_out_pad = l.n_out_pad; L.n_out = l.n_out; L.act = l.act;
            L.F = (l.n_out_pad + groups - 1) / groups;
            if (l.n_out_pad > kSrvActStride || l.ld > kSrvActStride) return 0;
            L.lds_off = off;
            off += L.F * l.ld + ((L.F + 3) & ~3);                         // rows + biases (16-byte granules)
            if (l.ld > max_ld) max_ld = l.ld;
        }
    a.n_layers = i; a.n_te = (int)TE.layers.size(); a.n_md = a.n_te + (int)MD.layers.size();
    a.mh_range = c->L.cfg.mh_range;
    a.groups = groups; a.one_xcd = groups == 32 ? 1 : 0;
    a.Db = c->L.cfg.dim_body; a.Da = c->L.cfg.dim_action; a.Z = c->L.cfg.latent; a.prior_kind = c->L.cfg.prior_kind;
    a.xs_off = off; a.xs_ld = max_ld;
    return (size_t)(off + S.max_rows * max_ld) * sizeof(float);
}

// scope: 0 = one XCD if the stacks fit its CUs' LDS, else the whole chip; 1 = one XCD; 2 = the whole chip
static int server_plan(pvae_ctx* c, RolloutServer& S, int scope) {
    const int n = (int)(c->L.net[PVAE_NET_TE].layers.size() + c->L.net[PVAE_NET_MD].layers.size() +
                        c->L.net[PVAE_NET_MH].layers.size());
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
    const bool helper = S.args.n_md < S.args.n_layers;
#define PVAE_SRV_GO(R, H)                                                                                                                 \
    do {                                                                                                                                  \
        HIP_TRY(hipFuncSetAttribute((const void*)rollout_server_kernel<R, H>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S.lds_bytes)); \
        hipLaunchKernelGGL((rollout_server_kernel<R, H>), dim3(256), dim3(256), S.lds_bytes, S.stream, S.args);                             \
    } while (0)
    if (S.max_rows == 1) { if (helper) PVAE_SRV_GO(1, true); else PVAE_SRV_GO(1, false); }
    else { if (helper) PVAE_SRV_GO(kSrvMaxRows, true); else PVAE_SRV_GO(kSrvMaxRows, false); }
#undef PVAE_SRV_GO
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
        HIP_TRY(hipMalloc((void**)&S.acts, (size_t)(kSrvMaxLayers + 1) * srv_slot_words(kSrvMaxRows) * sizeof(unsigned long long)));
        HIP_TRY(hipMemset(S.acts, 0, (size_t)(kSrvMaxLayers + 1) * srv_slot_words(kSrvMaxRows) * sizeof(unsigned long long)));
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
    static std::atomic<int> next_xcd{0};
    if (S.xcd < 0) S.xcd = next_xcd.fetch_add(1) & 7;
    S.args.xcd = S.xcd;
    return server_launch(c, S);
}

static int server_request(pvae_ctx* c, uint32_t cmd, const float* obs, int noise, uint64_t seed, uint64_t offset, double timeout_ms,
                          int rows = 1) {
    RolloutServer& S = *c->server;
    SrvReply* mb = S.mb;
    SrvRequest* rq = S.req;
    if (obs) memcpy((void*)rq->obs, obs, (size_t)rows * (cmd == 3u ? S.args.Db + S.args.Z : 2 * S.args.Db) * sizeof(float));
    const uint32_t op = cmd;
    cmd |= (uint32_t)(rows - 1) << 8;                            // (bits 8-9: rows - 1)
    rq->cmd = cmd; rq->noise = noise ? 1u : 0u;
    rq->seed_lo = (uint32_t)seed; rq->seed_hi = (uint32_t)(seed >> 32); rq->off_lo = (uint32_t)offset; rq->off_hi = (uint32_t)(offset >> 32);
    const uint32_t seq = ++S.seq;
    rq->check = srv_check(seq, cmd, rq->noise, rq->seed_lo, rq->seed_hi, rq->off_lo, rq->off_hi);
    // the request word goes LAST: behind a store fence when the block is device memory (write-combined stores through the
    // BAR may leave the core out of order; posted PCIe writes then arrive in the order they left)
    if (S.req_on_device) __builtin_ia32_sfence();
    __atomic_store_n(&rq->req_seq, seq, __ATOMIC_RELEASE);
    if (S.req_on_device) __builtin_ia32_sfence();
    if (op == 1) return 0;
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
    return pvae_rollout_server_infer_rows(c, obs, 1, noise, rng_seed, rng_offset, reload, a_hat, mu_logvar, z, timeout_ms);
}

int pvae_rollout_server_infer_rows(pvae_ctx* c, const float* obs, int32_t rows, int noise, uint64_t rng_seed, uint64_t rng_offset,
                                   int reload, float* a_hat, float* mu_logvar, float* z, double timeout_ms) {
    if (!c || !c->server || !c->server->mb) return fail(-2, "rollout server not started (pvae_rollout_server_start)");
    if (!obs || !a_hat) return fail(-1, "obs / a_hat is null");
    RolloutServer& S = *c->server;
    if (rows < 1 || rows > kSrvMaxRows) return fail(-1, "rollout server: rows %d outside [1, %d]", (int)rows, kSrvMaxRows);
    if (rows * 2 * S.args.Db > kSrvMaxObs || rows * (S.args.Da + 3 * S.args.Z) > kSrvMaxOut)
        return fail(-24, "rollout server: %d rows of this observation / action do not fit the mailbox", (int)rows);
    if (timeout_ms <= 0) timeout_ms = 1000.0;
    if (rows > S.max_rows) {
        // the first request with more than one row: the single-row instance makes way for the multi-row one (once; a few
        // hundred microseconds).  If the wider input buffer no longer fits the LDS plan, the single-row instance stays.
        // Planned BEFORE the resident instance is stopped, so a refusal (-24) leaves it serving single rows.
        const SrvArgs keep_args = S.args;
        const size_t keep_lds = S.lds_bytes;
        S.max_rows = kSrvMaxRows;
        int rc = server_plan(c, S, S.scope);
        if (rc) {
            S.max_rows = 1; S.args = keep_args; S.lds_bytes = keep_lds;
            return rc;
        }
        S.max_rows = 1; S.args = keep_args; S.lds_bytes = keep_lds;      // (stop() talks to the instance that is resident)
        if ((rc = pvae_rollout_server_stop(c))) return rc;
        S.max_rows = kSrvMaxRows;
        if ((rc = pvae_rollout_server_start(c, 0, 0, -1))) {
            S.max_rows = 1;
            return rc;
        }
        reload = 0;
    }
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
        const int r = server_request(c, reload ? 2u : 0u, obs, noise, rng_seed, rng_offset, timeout_ms, rows);
        if (r < 0) return r;
        if (r == 0) {
            ++S.served;
            const int Da = S.args.Da, Z = S.args.Z, n_out = Da + 3 * Z;
            for (int q = 0; q < rows; ++q) {                 // reply row q: [a_hat | mu | logvar | z]
                const float* o = (const float*)S.mb->out + (size_t)q * n_out;
                memcpy(a_hat + (size_t)q * Da, o, (size_t)Da * sizeof(float));
                if (mu_logvar) memcpy(mu_logvar + (size_t)q * 2 * Z, o + Da, (size_t)2 * Z * sizeof(float));
                if (z) memcpy(z + (size_t)q * Z, o + Da + 2 * Z, (size_t)Z * sizeof(float));
            }
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
            const int keep_rows = S.max_rows;
            int rc = pvae_rollout_server_stop(c);                // (no reload form of this request: a relaunch re-reads)
            S.max_rows = keep_rows;
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
    S.max_rows = 1;                                              // (an explicit stop: the next start is the single-row instance again)
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

void server_free(pvae_ctx* c) {
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

