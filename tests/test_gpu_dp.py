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


def test_in_library_exchange_calls(golden):
    """pvae_comm_* / pvae_dp_train_step / pvae_allreduce_grads driven directly (one-rank
    communicator, no torch.distributed at all): a data-parallel step equals the fused step bit for
    bit in both phases, and an EMPTY shard (rows = 0, ragged last global batch) contributes zeros
    and applies the same Adam update as every other rank."""
    import numpy as np  # noqa: F401
    from oracle import refpath as R
    from physicsvae_amd import _lib
    from physicsvae_amd.engine import make_step_params
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_trainer
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
    data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")
    tr = make_trainer(arch, data, 32, device="cuda")
    sd = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    with pytest.raises(RuntimeError):
        eng.allreduce_grads(0, 64)                                  # no communicator yet
    eng.comm_init(0, 1, eng.comm_unique_id())
    assert eng.has_comm
    eps = R.eps_stream(2, 8)(0, (32, 8))
    for phase, world, nets in ((_lib.PHASE_WORLD, True, [_lib.NET_WM]), (_lib.PHASE_JOINT, False, [_lib.NET_TE, _lib.NET_MD])):
        c = R.phase_coeffs(world)
        res = []
        for dp in (False, True):
            tr.model.load_state_dict(sd)
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            out = torch.zeros(5, device="cuda")
            for t in (1, 2, 3):
                sp = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                      s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=32)
                (eng.dp_train_step if dp else eng.train_step)(phase, 32 * (t - 1), 32, sp, eps=eps, loss_out=out)
            res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        # empty shard: zero gradient through the collective, then Adam
        p0, m0, v0 = eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()
        sp = make_step_params(lr=5e-4, adam_t=(4, 4, 4), global_rows=17)
        eng.grads.fill_(3.0)
        eng.dp_train_step(phase, 0, 0, sp)
        got = eng.params.clone()
        eng.params.copy_(p0); eng.exp_avg.copy_(m0); eng.exp_avg_sq.copy_(v0)
        eng.segment(eng.grads, nets).zero_()
        eng.adam(nets, sp)
        assert torch.equal(eng.params, got)
        assert not torch.equal(got, p0)                             # the moments still move the weights
    eng.comm_destroy()
    assert not eng.has_comm


def test_bench_multi_rank_path_runs_end_to_end(tmp_path):
    """`bench.py --gpus 2` exactly as the driver launches it (torch.distributed.run, one process per
    rank), with both ranks on this box's single GPU through the test hooks (gloo instead of RCCL,
    which refuses two ranks on one device): the N > 1 branches of the benchmark -- sharding,
    barrier-bracketed timing with the max over ranks, the instrumented roofline pass, rank-0-only
    JSON -- run and produce a well-formed line."""
    import json
    env = dict(os.environ, PVAE_LOCAL_DEVICE="0", PVAE_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29571", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "20", "--warmup", "3"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert r.stdout.strip().splitlines()[-1] == lines[0]          # and it is the last line on stdout
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 512 and d["config"]["parallelism"] == "dp2"
    assert d["value"] == pytest.approx(512 * 20 / (d["ms_per_step"] * 20 * 1e-3), rel=1e-6)
    assert "cpu_baseline" not in d and d["roofline"]["bound"] == "mfma" and d["vs_baseline"] is None


def test_in_library_exchange_with_lookahead(golden):
    """The data-parallel step over a lookahead-2 unroll (stacked time steps, one weight-gradient
    launch per layer, per-net all-reduce, flat Adam) equals the fused single-GPU step bit for bit."""
    from oracle import refpath as R
    from physicsvae_amd import _lib
    from physicsvae_amd.engine import make_step_params
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_trainer
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
    data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")
    tr = make_trainer(arch, data, 32, device="cuda", extra={"lookahead": 2})
    sd = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.comm_init(0, 1, eng.comm_unique_id())
    es = R.eps_stream(2, 8)
    eps = torch.stack([es(t, (32, 8)) for t in range(2)])
    for phase, world in ((_lib.PHASE_WORLD, True), (_lib.PHASE_JOINT, False)):
        c = R.phase_coeffs(world)
        res = []
        for dp in (False, True):
            tr.model.load_state_dict(sd)
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            out = torch.zeros(5, device="cuda")
            for t in (1, 2):
                sp = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                      s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=32)
                (eng.dp_train_step if dp else eng.train_step)(phase, 32 * (t - 1), 32, sp, eps=eps, loss_out=out)
            res.append((eng.params.clone(), eng.exp_avg.clone(), out.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
    eng.comm_destroy()
