"""Build libpvae_gfx950.so in-tree with hipcc (cross-compiles for gfx950 without a GPU): one object per translation
unit of physicsvae_amd/csrc (compiled in parallel), linked into one shared library."""
import concurrent.futures
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpvae_gfx950.so")
UNITS = ["pvae.hip", "pvae_exchange.hip", "pvae_rollout_server.hip", "pvae_probe.hip"]
HEADERS = ["pvae_internal.h", "pvae_gemm.h", "pvae_layout.h"]
SOURCES = UNITS + HEADERS
HEADER = os.path.join(os.path.dirname(HERE), "include", "pvae.h")
# -amdgpu-kernarg-preload-count: leading scalar / pointer kernel arguments arrive in SGPRs at wave launch (gfx950)
# instead of by an s_load inside the kernel (pvae_gemm.h PVAE_GA_PARAMS; tools/kernarg_preload_probe.hip)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-kernarg-preload-count=16", "-Wall",
         "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libpvae_gfx950.so")


STAMP = LIB + ".sha256"         # digest of the sources + flags the in-tree library was built from (git-ignored, travels with it)


def source_digest(defines=()):
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [HEADER]:
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read() + b"\0")
    h.update(" ".join(FLAGS + ["-D" + d for d in defines]).encode())
    return h.hexdigest()


def is_stale():
    """By CONTENT, not by modification time: a copy of the tree (the snapshot a GPU box receives) keeps no usable mtimes, and a
    spurious rebuild there costs every pytest session 30 s -- or, with several sessions starting at once, lets two linkers
    write the same file."""
    if not os.path.exists(LIB):
        return True
    try:
        with open(STAMP) as f:
            return f.read().strip() != source_digest()
    except OSError:
        return True


def build(force=False, verbose=False, defines=(), out=None):
    """`defines`: extra -D flags (diagnostic builds under ab_libs/); `out`: where the library goes (default: in-tree)."""
    out = out or LIB
    if not force and out == LIB and not is_stale():
        return LIB
    cc = _hipcc()
    tmp = tempfile.mkdtemp(prefix="pvae_build_")
    try:
        def compile_unit(u):
            obj = os.path.join(tmp, u.replace(".hip", ".o"))
            cmd = [cc] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, u), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (u, res.stdout, res.stderr))
            return obj
        with concurrent.futures.ThreadPoolExecutor(len(UNITS)) as pool:
            objs = list(pool.map(compile_unit, UNITS))
        link_tmp = os.path.join(tmp, "lib.so")                    # (private to this build: concurrent builds cannot interleave)
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", link_tmp]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
        shutil.copyfile(link_tmp, out + ".tmp.%d" % os.getpid())
        os.chmod(out + ".tmp.%d" % os.getpid(), 0o755)
        os.replace(out + ".tmp.%d" % os.getpid(), out)
        if out == LIB:
            with open(STAMP + ".tmp.%d" % os.getpid(), "w") as f:
                f.write(source_digest(defines) + "\n")
            os.replace(STAMP + ".tmp.%d" % os.getpid(), STAMP)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


if __name__ == "__main__":
    import sys
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    outs = [a for a in sys.argv[1:] if not a.startswith("-D")]
    print(build(force=True, verbose=True, defines=defs, out=os.path.abspath(outs[0]) if outs else None))
