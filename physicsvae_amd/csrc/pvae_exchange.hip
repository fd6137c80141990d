// pvae_exchange.hip -- the data-parallel gradient exchange (tm:131-161 is one process): RCCL calls
// bound at run time, the peer-mapped exchange kernels (pull / push), their set-up over hipIpc, the attach-time self-test.
// gfx950 only.
#include "pvae_internal.h"

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


extern "C" {
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
// Default exchange schedule.  With one rank there is nothing to hide: in line.  With several ranks the
// all-reduce of a stack (14 MB) takes about as long over xGMI as the backward pass of a stack (~100 us), so
// in the JOINT phase the decoder's reduction is worth hiding behind the encoder's backward pass even at the
// ~27 us the two stream hand-offs cost (section 5 of DESIGN.md): 6 MiB buckets on the exchange stream.  The
// world phase has one stack and ~36 us of backward left after its first bucket closes: in line.
// A function of (communicator size, phase) only, so every rank chooses the same.
extern "C++" int64_t auto_bucket_bytes(const pvae_ctx* c, int phase) {
    if (c->bucket_bytes >= 0) return c->bucket_bytes;
    return (c->comm_world > 1 && phase == PVAE_PHASE_JOINT && c->comm_stream) ? (int64_t)6 << 20 : 0;
}
extern "C++" std::vector<Bucket> exchange_buckets(const pvae_ctx* c, int net) {
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
extern "C++" int exchange_bucket(pvae_ctx* c, int net, const Bucket& b, const pvae_step_params* sp, hipStream_t st,
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

}  // extern "C"
