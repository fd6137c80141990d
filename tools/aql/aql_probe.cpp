// aql_probe.cpp -- what does a DEPENDENT kernel boundary cost as a function of the AQL packet's fence scopes?
// HIP dispatches every kernel with barrier = 1, acquire = agent, release = agent (AMD_LOG_LEVEL=4: header 0xb02) and offers
// no way to change that.  This probe goes below HIP: its own HSA queue, hand-written AQL packets, the kernels loaded
// from a code object (tools/aql/aql_kernels.hip, hipcc --genco).  Back-to-back dependent launches of
//   (a) an empty 256-workgroup kernel, (b) a 256-workgroup chain kernel that checks what its predecessor wrote,
// with acquire / release scopes NONE, AGENT, SYSTEM.
// build: hipcc --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -O3 -o aql_kernels.hsaco aql_kernels.hip
//        g++ -O2 -I/opt/rocm/include aql_probe.cpp -o aql_probe -L/opt/rocm/lib -lhsa-runtime64
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m; hsa_status_string(s_, &m); printf("%s: %s\n", #x, m); exit(1); } } while (0)

static hsa_agent_t g_gpu; static bool g_have = false;
static hsa_amd_memory_pool_t g_vram, g_kernarg; static bool g_hv = false, g_hk = false;
static hsa_status_t agent_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have) { g_gpu = a; g_have = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t pool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_hv) { g_vram = p; g_hv = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_agent_t g_cpu; static bool g_hc = false;
static hsa_status_t cpu_cb(hsa_agent_t a, void*) {
    hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && !g_hc) { g_cpu = a; g_hc = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_status_t kpool_cb(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    uint32_t flags; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_hk) { g_kernarg = p; g_hk = true; }
    return HSA_STATUS_SUCCESS;
}

struct Kern { uint64_t object; uint32_t kernarg, group, priv; };
static Kern get_kernel(hsa_executable_t ex, const char* name) {
    hsa_executable_symbol_t sym;
    CK(hsa_executable_get_symbol_by_name(ex, name, &g_gpu, &sym));
    Kern k;
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
    CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
    return k;
}

struct ChainArgs { const unsigned* in; unsigned* out; unsigned expect, val; unsigned* err; int sc1; };

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "aql_kernels.hsaco";
    CK(hsa_init());
    CK(hsa_iterate_agents(agent_cb, nullptr));
    CK(hsa_iterate_agents(cpu_cb, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_gpu, pool_cb, nullptr));
    CK(hsa_amd_agent_iterate_memory_pools(g_cpu, kpool_cb, nullptr));
    if (!g_hv || !g_hk) { printf("pools not found\n"); return 1; }
    hsa_queue_t* q;
    CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    int fd = open(path, O_RDONLY);
    if (fd < 0) { printf("cannot open %s\n", path); return 1; }
    hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_file(fd, &rd));
    hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
    CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
    CK(hsa_executable_freeze(ex, nullptr));
    const Kern ke = get_kernel(ex, "empty_k.kd"), kc = get_kernel(ex, "chain_k.kd");
    printf("empty_k: kernarg %u B; chain_k: kernarg %u B\n", ke.kernarg, kc.kernarg);
    unsigned *bufA, *bufB, *err;
    CK(hsa_amd_memory_pool_allocate(g_vram, 65536 * 4, 0, (void**)&bufA));
    CK(hsa_amd_memory_pool_allocate(g_vram, 65536 * 4, 0, (void**)&bufB));
    CK(hsa_amd_memory_pool_allocate(g_vram, 4096, 0, (void**)&err));
    CK(hsa_amd_memory_fill(bufA, 0, 65536)); CK(hsa_amd_memory_fill(bufB, 0, 65536)); CK(hsa_amd_memory_fill(err, 0, 1024));
    const int N = 2000;
    char* kargs;
    CK(hsa_amd_memory_pool_allocate(g_kernarg, (size_t)N * 64, 0, (void**)&kargs));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));
    unsigned* herr;
    CK(hsa_amd_memory_pool_allocate(g_kernarg, 4096, 0, (void**)&herr));
    CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, herr));
    hsa_signal_t done; CK(hsa_signal_create(1, 0, nullptr, &done));

    const char* scope_name[3] = {"none", "agent", "system"};
    for (int kind = 0; kind < 3; ++kind) {                 // 0: empty kernel, 1: chain with plain accesses, 2: chain with sc0 sc1 accesses
        for (int acq = 0; acq < 3; ++acq) {
            for (int rel = 0; rel < 3; ++rel) {
                if (acq != rel && !(acq == 0 && rel == 1) && !(acq == 1 && rel == 0)) continue;
                double best = 1e30; unsigned mism = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hsa_amd_memory_fill(bufA, 0, 65536)); CK(hsa_amd_memory_fill(bufB, 0, 65536)); CK(hsa_amd_memory_fill(err, 0, 1024));
                    for (int i = 0; i < N; ++i) {
                        ChainArgs a{(i & 1) ? bufA : bufB, (i & 1) ? bufB : bufA, i >= 2 ? (unsigned)(i - 1) : 0u, (unsigned)i + 1, err, kind == 2};
                        if (i == 1) a.expect = 1;            // launch 0 wrote 1 into bufA ... (values: launch i writes i + 1, reads what launch i-1 wrote = i)
                        a.expect = i == 0 ? 0u : (unsigned)i;
                        memcpy(kargs + (size_t)i * 64, &a, sizeof(a));
                    }
                    hsa_signal_store_relaxed(done, 1);
                    const uint64_t base = hsa_queue_load_write_index_relaxed(q);
                    const auto t0 = std::chrono::steady_clock::now();
                    for (int i = 0; i < N; ++i) {
                        while (base + i - hsa_queue_load_read_index_scacquire(q) >= q->size) {}
                        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((base + i) & (q->size - 1));
                        const Kern& k = kind == 0 ? ke : kc;
                        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
                        p->workgroup_size_x = 256; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
                        p->grid_size_x = 256 * 256; p->grid_size_y = 1; p->grid_size_z = 1;
                        p->private_segment_size = k.priv; p->group_segment_size = k.group;
                        p->kernel_object = k.object;
                        p->kernarg_address = kargs + (size_t)i * 64;
                        p->completion_signal.handle = i == N - 1 ? done.handle : 0;
                        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                                (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                        __atomic_store_n((uint16_t*)p, header, __ATOMIC_RELEASE);
                        hsa_queue_store_write_index_screlease(q, base + i + 1);
                        hsa_signal_store_screlease(q->doorbell_signal, base + i);
                    }
                    while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_ACTIVE) != 0) {}
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                    if (us < best) best = us;
                    CK(hsa_memory_copy(herr, err, 4));
                    mism = herr[0];
                }
                printf("%-28s acquire %-6s release %-6s : %.2f us per launch%s\n",
                       kind == 0 ? "empty 256-WG kernel" : kind == 1 ? "chain kernel, plain accesses" : "chain kernel, sc0 sc1",
                       scope_name[acq], scope_name[rel], best, kind ? (mism ? "   MISMATCHES" : "   all values fresh") : "");
                if (kind && mism) printf("    (%u stale reads)\n", mism);
            }
        }
    }
    return 0;
}
