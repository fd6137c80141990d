"""The drop-in boundary used from plain C: tests/c_abi/train_loop.c is a host program with no
Python and no torch in it (dlopen of libpvae_gfx950.so + the HIP runtime for device memory).  The CPU
test compiles it against include/pvae.h; the GPU test runs it: world-model steps, then joint steps,
finite losses that fall; then one observation through the per-layer launches and through the call-persistent
rollout server (host pointers in and out), whose actions must be the same bits."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "train_loop.c")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def _build(out):
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", SRC, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROCM, "include"), "-L", os.path.join(ROCM, "lib"), "-Wl,-rpath," + os.path.join(ROCM, "lib"),
           "-lamdhip64", "-ldl", "-lm", "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_c_host_compiles_against_the_header(tmp_path):
    _build(str(tmp_path / "train_loop"))


@pytest.mark.gpu
def test_c_host_trains_through_the_abi(tmp_path):
    from physicsvae_amd import build
    exe = str(tmp_path / "train_loop")
    _build(exe)
    r = subprocess.run([exe, build.LIB], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    assert r.stdout.strip().splitlines()[-1].startswith("ok "), r.stdout[-500:]
    assert "served action == launched action (bit for bit)" in r.stdout
    print(r.stdout[-400:], file=sys.stderr)
