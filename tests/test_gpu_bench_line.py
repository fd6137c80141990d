"""bench.py as the driver launches it: the plain single-GPU line, `--gpus 2` self-launched, and under
`python -m torch.distributed.run` (on a 1-GPU box the ranks share the device over gloo: same code path, flagged)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import arch_from_meta, make_trainer, max_err_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(flags), cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_bench_plain_two_ranks_self_launch():
    """`python bench.py --gpus 2`, launched plainly, starts its two ranks itself and prints ONE line.
    On a 1-GPU box the ranks share the device and exchange over gloo (RCCL refuses two ranks on one
    device): same sharding / reduction / Adam path, flagged in the line."""
    d = _bench("--gpus", "2", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-rocprof")
    assert d["n_gpus"] == 2 and d["steps_requested"] == 20 and d["steps"] == d["timing"]["timed_steps_per_region"] and d["warmup"] == 5
    assert d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp2"
    assert d["timing"]["timed_steps_per_region"] >= 200 and d["timing"]["regions"] == 3
    assert d["value"] > 0 and np.isfinite(d["last_loss"])
    if torch.cuda.device_count() < 2:
        assert d["ranks_share_a_gpu"] is True and d["rccl_ranks"] == 0
    else:
        assert d["ranks_share_a_gpu"] is False and d["rccl_ranks"] == 2
        assert d["allreduce_us_per_step"] > 0
    assert "roofline" in d and 0 < d["roofline"]["frac"] < 1
    # the N-rank step against ONE process with the global batch (three optimizer steps from a common state): what a
    # stale read of peer-written parameters could not pass while the replicas stay bit-identical
    ref = d["single_process_reference"]
    assert d["matches_single_process"] is True and d["replicas_identical"] is True, ref
    assert ref["steps"] == 3 and ref["global_batch"] == 512 and ref["max_rel_loss_diff"] < 1e-5 and ref["update_rel_l2_diff"] < 1e-3
    assert ref["losses_n_ranks"][0] != ref["losses_n_ranks"][1]                 # the steps really trained
    # every exchange form, back to back in the same run (what a multi-GPU lease must yield in one go)
    sweep = d["exchange_sweep"]
    assert set(sweep) == {"inline", "bucketed", "sharded", "p2p", "p2p_push", "local"}
    for form in ("p2p", "p2p_push"):
        assert sweep[form]["p2p_ranks"] == 2 and sweep[form]["timeouts"] == 0 and sweep[form]["value"] > 0
    assert sweep["p2p"]["exchange_launches_per_step"] >= 2          # one launch per bucket, two stacks in the joint phase
    assert sweep["local"]["ms_per_step"] > 0 and "exchange_exposed_us_per_step" in sweep["p2p"]
    if torch.cuda.device_count() < 2:
        assert all("skipped" in sweep[m] for m in ("inline", "bucketed", "sharded"))     # RCCL needs one GPU per rank
    else:
        assert all(sweep[m]["value"] > 0 for m in ("inline", "bucketed", "sharded"))


def test_bench_under_the_launcher_command_of_the_scaling_run():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    --gpus N ...`: the command the multi-GPU scaling run uses, here with N = 2.  On a box with fewer GPUs than ranks
    the ranks share the devices over gloo (`parallel.init_from_env`): same code path up to the transport, flagged in
    the line; the last stdout line that parses is rank 0's ONE JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29633", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "10", "--warmup", "3"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps_requested"] == 10 and d["steps"] >= 200 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 512 and d["value"] > 0 and np.isfinite(d["last_loss"])
    assert d["replicas_identical"] is True and d["matches_single_process"] is True, d.get("single_process_reference")
    assert d["ranks_share_a_gpu"] is (torch.cuda.device_count() < 2)
    assert d["exchange_autotune"]["chosen"] in ("inline", "bucketed", "sharded", "p2p", "p2p_push")


def test_bench_plain_single_gpu_line():
    d = _bench("--steps", "20", "--warmup", "5", "--no-rocprof")
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 0 and d["config"]["phase"] == "joint"
    assert d["metric"].startswith("train samples/sec (world-model+VAE step)")
    for key in ("roofline", "world_roofline", "cpu_baseline", "world_value"):
        assert key in d, key
    for roof in (d["roofline"], d["world_roofline"]):
        assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["peak"] == 157.3
    cb = d["cpu_baseline"]
    assert cb["value"] >= 0.8 * cb["value_1thread"] > 0 and cb["threads_best"] in [int(k) for k in cb["sweep"]]
    # whole epochs after one warm-up epoch at the best thread count; physical cores reported beside the threads
    assert cb["epochs_timed"] >= 1 and cb["warmup_epochs"] == 1 and cb["samples_per_epoch"] == 9990
    assert cb["cores"] == cb["threads_best"] and cb["physical_cores"] and cb["physical_cores"] <= cb["host_cpus"]
    assert len(d["timing"]["region_values"]) == 3
