"""The step under CONTENTION: four processes share the GPU, each repeating the same three optimizer steps from the same
state; every arena must equal the first repetition's bit for bit.  Round 5's one real kernel bug -- a write-after-read race
in the six-slot LDS-DMA ring of the wave-specialised forward kernel (pvae_gemm.h, `splitk_ws_body`) -- was invisible to
every single-process test and showed within tens of repetitions as soon as other processes' workgroups shared the CUs
(the same pressure an RCCL or peer-exchange kernel beside the step exerts on a real node).  This is that hunt
(tools/p2p_race_hunt.py) as a driver-run test, over EVERY LDS-DMA ring the step can run on:

  c3         256 rows, dims 197 / 45     32x32 tiles: six-slot super-step ring, and the `_pro_` kernel's four-slot ring
  c3_direct  the same with `pvae_set_direct`: the `*_gather_kernel` forms of the first layers
  c5         512 rows, dims 400 / 90     64x32 tiles (gemm_splitk_ws64_kernel, PT = 32)
  r1024      1024 rows                   64x64 tiles (PT = 64)

both phases each, R = 10 repetitions of 3 steps per process.  The reference has no counterpart (tm:131-161 is one process
on one device)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "contention_worker.py")
NPROC, R = 4, 10


def test_four_processes_on_one_gpu_every_repetition_bit_identical(tmp_path):
    out = str(tmp_path / "c")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PVAE_DIRECT")}
    procs = [subprocess.Popen([sys.executable, WORKER, ROOT, out, str(r), str(NPROC), str(R)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(NPROC)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    reports = [torch.load("%s.%d" % (out, r)) for r in range(NPROC)]
    keys = sorted(reports[0])
    assert len(keys) == 8, keys
    for r, rep in enumerate(reports):
        for k in keys:
            e = rep[k]
            assert e["repetitions"] == R and e["trained"], (r, k, e)
            assert e["direct"] == k.startswith("c3_direct"), (r, k, e)        # the direct configuration really took the gathered kernels
            assert e["deviating"] == [], "process %d, %s: repetitions that differ from the first (rep, [params, m, v, losses] words): %s" % (r, k, e["deviating"])
