"""Can the HOST read and write device memory through its own virtual address (large BAR), and how long does a host write
take to become visible to a polling kernel?  Decides whether the rollout server's request mailbox can live in DEVICE memory
(host pushes the observation, the kernel polls local memory) instead of pinned host memory (the kernel pulls over PCIe).
Each attempt runs in a child process: a fault there is an answer, not a crash of the caller.
    python tools/bar_probe.py"""
import ctypes
import multiprocessing as mp
import os
import sys


def attempt(kind, q):
    import torch
    hip = ctypes.CDLL("libamdhip64.so")
    p = ctypes.c_void_p()
    if kind == "hipMalloc":
        rc = hip.hipMalloc(ctypes.byref(p), 4096)
    else:
        flags = {"finegrained": 0x1, "uncached": 0x3}[kind]
        rc = hip.hipExtMallocWithFlags(ctypes.byref(p), 4096, flags)
    if rc != 0:
        q.put((kind, "alloc failed rc=%d" % rc))
        return
    src = (ctypes.c_uint32 * 4)(0x11111111, 0x22222222, 0x33333333, 0x44444444)
    hip.hipMemcpy(p, src, 16, 1)                       # H2D through the runtime
    hip.hipDeviceSynchronize()
    q.put((kind, "allocated, trying a host read"))
    back = (ctypes.c_uint32 * 4)()
    ctypes.memmove(back, p.value, 16)                  # HOST load through the device pointer
    ok_read = list(back) == list(src)
    q.put((kind, "host read ok=%s values=%s" % (ok_read, [hex(v) for v in back])))
    w = (ctypes.c_uint32 * 4)(0xAAAA0001, 0xAAAA0002, 0xAAAA0003, 0xAAAA0004)
    ctypes.memmove(p.value, w, 16)                     # HOST store through the device pointer
    chk = (ctypes.c_uint32 * 4)()
    hip.hipMemcpy(chk, p, 16, 2)                       # D2H through the runtime
    q.put((kind, "host write visible to the runtime's copy: %s" % (list(chk) == list(w))))


def main():
    mp.set_start_method("spawn")
    for kind in ("uncached", "finegrained", "hipMalloc"):
        q = mp.Queue()
        pr = mp.Process(target=attempt, args=(kind, q))
        pr.start()
        pr.join(60)
        msgs = []
        while not q.empty():
            msgs.append(q.get())
        print(kind, "exit code", pr.exitcode, "|", " ; ".join(m[1] for m in msgs), flush=True)


if __name__ == "__main__":
    main()
