import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from physicsvae_amd import _lib
lib = _lib.load()
dev = "cuda"
scratch = torch.zeros(1024, device=dev)
for name, t in (("uniform[-1,1]", torch.rand(1 << 20, device=dev) * 2 - 1), ("normal*0.03", torch.randn(1 << 20, device=dev) * 0.03),
                ("zeros", torch.zeros(1 << 20, device=dev)), ("normal", torch.randn(1 << 20, device=dev))):
    g, p = C.c_double(), C.c_double()
    for rep in range(2):
        _lib.check(lib.pvae_mfma_clock_probe(t.data_ptr(), t.numel(), scratch.data_ptr(), C.byref(g), C.byref(p), torch.cuda.current_stream().cuda_stream))
    print("%-14s %.3f GHz  %.1f TFLOP/s" % (name, g.value, p.value))
