"""Build libpvae_gfx950.so in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpvae_gfx950.so")
SOURCES = ["pvae.hip", "pvae_gemm.h", "pvae_layout.h"]
HEADER = os.path.join(os.path.dirname(HERE), "include", "pvae.h")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpvae_gfx950.so")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    # -amdgpu-kernarg-preload-count: leading scalar / pointer kernel arguments arrive in SGPRs at wave launch (gfx950)
    # instead of by an s_load inside the kernel (pvae_gemm.h PVAE_GA_PARAMS; tools/kernarg_preload_probe.hip)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-mllvm", "-amdgpu-kernarg-preload-count=16",
           "-Wall", "-Wno-unused-function", os.path.join(CSRC, "pvae.hip"), "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
