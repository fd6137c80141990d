"""Data-parallel path end to end on ONE GPU: two ranks (both on cuda:0, gloo all-reduce of the
CUDA gradient arena) train for two epochs across the phase switch; a single process trains the
same schedule with the global batch.  Parameters must agree (same reduced gradient up to fp32
summation order across the shard boundary), and the two replicas must be bit-identical."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, io, contextlib, torch
root = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from physicsvae_amd import parallel
rank, world, _ = parallel.init_from_env(backend=os.environ.get("PVAE_TEST_BACKEND", "gloo"))
from oracle import refpath as R
from util import make_trainer
arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")          # 117 windows
per_gpu = int(sys.argv[3])
with contextlib.redirect_stdout(io.StringIO()):
    tr = make_trainer(arch, data, per_gpu, m_world=1, device="cuda:0", eps_fn=R.eps_stream(2, 8))
tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, 1), 3))
if "PVAE_TEST_EXPECT_COMM" in os.environ:
    assert tr.engine.has_comm == (os.environ["PVAE_TEST_EXPECT_COMM"] == "1"), tr.engine.has_comm
losses = [tr.train()["mean_train_loss"] for _ in range(2)]
sd = {k: v.cpu() for k, v in tr.model.state_dict().items()}
torch.save({"sd": sd, "losses": losses, "steps": dict(tr.optimizer.net_steps)}, out + ".%d" % rank)
if world > 1:
    torch.distributed.barrier()
print("DONE", rank, losses)
'''


def _run(tmp_path, world, per_gpu, tag, **extra_env):
    script = tmp_path / "dp_worker.py"
    script.write_text(WORKER)
    out = str(tmp_path / tag)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", WORLD_SIZE=str(world), **extra_env)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, out, str(per_gpu)],
                              env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(out + ".%d" % r) for r in range(world)]


def test_two_ranks_equal_one_process_with_global_batch(tmp_path):
    dp = _run(tmp_path, 2, 16, "dp")            # 2 ranks x 16 rows: global batch 32, ragged tail 117 % 32 = 21
    single = _run(tmp_path, 1, 32, "single")[0]
    a, b = dp
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k            # replicas stay bit-identical
    assert a["steps"] == single["steps"]
    assert a["losses"] == pytest.approx(single["losses"], rel=1e-5)
    for k, v in single["sd"].items():
        if k.startswith("_value_branch"):
            continue
        err = float((a["sd"][k] - v).norm() / (v.norm() + 1e-30))
        assert err < 2e-3, (k, err)


@pytest.mark.parametrize("transport", ["rccl", "torch"])
def test_rccl_collective_path_single_rank_is_bit_identical(tmp_path, transport):
    """The data-parallel step driven through the real RCCL library with one rank: the reduction is
    the identity, so any mis-ordering between the backward launches, the collective and Adam would
    show up as a difference from the fused single-GPU step, which it must match bit for bit.
    transport "rccl": the library's own communicator, everything on one stream inside
    `pvae_dp_train_step`; "torch": staged backward + torch.distributed's nccl backend (bucketed
    asynchronous all-reduce on RCCL's stream, Adam per bucket after `wait()`)."""
    rccl = _run(tmp_path, 1, 32, "rccl", PVAE_DP_ALWAYS_REDUCE="1", PVAE_TEST_BACKEND="nccl",
                PVAE_DP_TRANSPORT=transport, PVAE_TEST_EXPECT_COMM="1" if transport == "rccl" else "0")[0]
    single = _run(tmp_path, 1, 32, "plain")[0]
    assert rccl["steps"] == single["steps"] and rccl["losses"] == single["losses"]
    for k, v in single["sd"].items():
        assert torch.equal(rccl["sd"][k], v), k
