// device code of tools/aql/aql_probe.cpp (built with hipcc --genco into a code object that the probe loads through HSA)
#include <hip/hip_runtime.h>
extern "C" __global__ void empty_k() {}
// chain kernel: reads what the previous launch wrote (through the caches as the mode says), writes its own value
extern "C" __global__ void chain_k(const unsigned* in, unsigned* out, unsigned expect, unsigned val, unsigned* err, int sc1) {
    const unsigned i = __builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
    unsigned v;
    if (sc1) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(in + (i * 2654435761u) % 65536) : "memory");
    else v = in[(i * 2654435761u) % 65536];
    if (v != expect) atomicAdd(err, 1u);
    if (sc1) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(out + i), "v"(val) : "memory");
    else out[i] = val;
}
