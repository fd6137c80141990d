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
    # (with more than one rank an unset dp_exchange means "auto" since round 4 -- on this box that would pick a
    #  peer-mapped form; these tests are about the torch.distributed transport and the library's own schedule)
    extra_env.setdefault("PVAE_DP_EXCHANGE", "default")
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
    assert d["n_gpus"] == 2 and d["steps_requested"] == 20 and d["steps"] >= 200 and d["warmup"] == 3 and d["scaling"] == "weak"
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


@pytest.mark.parametrize("lookahead", [1, 2])
@pytest.mark.parametrize("bucket_mb,delay_us", [(0.0, 0), (0.01, 0), (0.01, 400), (0.05, 150), (6.0, 400)])
def test_overlapped_exchange_ordering(golden, lookahead, bucket_mb, delay_us):
    """The bucketed exchange of pvae_dp_train_step runs on the library's own stream.  With every
    bucket size (per layer ... per stack ... in line) and with the reductions artificially slowed
    down by a spin kernel (so a missing event wait would let Adam or the next forward pass run on
    half-finished data), several consecutive steps -- including an empty shard -- give the same
    parameters and moments, bit for bit, as the single-stream fused step."""
    from oracle import refpath as R
    from physicsvae_amd import _lib
    from physicsvae_amd.engine import make_step_params
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_trainer
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 3), wm=(128, 3))
    data = R.synth_demo(0, 3, 60, 23, 7, kind="dynamics")
    tr = make_trainer(arch, data, 32, device="cuda", extra={"lookahead": lookahead})
    sd = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.comm_init(0, 1, eng.comm_unique_id())
    eng.comm_config(bucket_mb, delay_us)
    es = R.eps_stream(2, 8)
    eps = torch.stack([es(t, (32, 8)) for t in range(lookahead)]) if lookahead > 1 else es(0, (32, 8))
    for phase, world, nets in ((_lib.PHASE_WORLD, True, [_lib.NET_WM]), (_lib.PHASE_JOINT, False, [_lib.NET_TE, _lib.NET_MD])):
        c = R.phase_coeffs(world)
        res = []
        for dp in (False, True):
            tr.model.load_state_dict(sd)
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            out = torch.zeros(5, device="cuda")
            for t in range(1, 6):
                sp = make_step_params(lr=5e-4, adam_t=(t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                      s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=32)
                if t == 3:                         # an empty shard in the middle of the run
                    if dp:
                        eng.dp_train_step(phase, 0, 0, sp)
                    else:
                        eng.segment(eng.grads, nets).zero_()
                        eng.adam(nets, sp)
                    continue
                first = 32 * ((t - 1) % 3)
                if dp:
                    eng.dp_train_step(phase, first, 32, sp, eps=eps, loss_out=out, next_span=(32 * (t % 3), 32))
                else:
                    eng.train_step(phase, first, 32, sp, eps=eps, loss_out=out)
            res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
    eng.comm_destroy()


def test_exchange_buckets_at_bench_sizes():
    """BASELINE configs[1] sizes (4x1024 stacks, batch 256): 6 MiB buckets split every stack's
    exchange in two; the overlapped run and the in-line run both equal the fused single-stream step
    bit for bit.  (Prints the one-rank step times: what the hand-offs cost with nothing to hide.)"""
    import contextlib, io, time
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_demo import make_trainer as mk, synth_demo
    from physicsvae_amd.engine import make_step_params
    data = synth_demo(0, 4, 400, 197, 45)
    with contextlib.redirect_stdout(io.StringIO()):
        tr = mk(data, 256, "cuda")
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.comm_init(0, 1, eng.comm_unique_id())
    p0 = eng.params.clone()
    out = torch.zeros(5, device="cuda")
    side = torch.cuda.Stream()                 # not the NULL stream (see pvae_comm_config)
    side.wait_stream(torch.cuda.current_stream())
    for name in ("world", "joint"):
        w = name == "world"
        tr.model.set_learnable_task_encoder(not w); tr.model.set_learnable_motor_decoder(not w)
        tr.model.set_learnable_world_model(w)
        tr.read_loss_fn_coeff(world=w)
        phase, nets = tr.phase()
        res, rate = [], {}
        torch.cuda.synchronize()
        torch.cuda.set_stream(side)
        for mode in ("fused", "overlap", "inline"):
            eng.params.copy_(p0); eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            if mode != "fused":
                eng.comm_config(6.0 if mode == "overlap" else 0.0, 0)
            n = 60
            for rep in range(2):                   # second pass is the timed one
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    sp = make_step_params(lr=1e-3, adam_t=(i + 1, i + 1, i + 1), a_rec=tr.a_rec_coeff, kl=tr.vae_kl_coeff,
                                          s_rec=tr.s_rec_coeff, cyc=tr.vae_cycle_coeff, global_rows=256, seed=7,
                                          offset=i * 65536)
                    first = 256 * (i % 5)
                    if mode == "fused":
                        eng.train_step(phase, first, 256, sp, loss_out=out)
                    else:
                        eng.dp_train_step(phase, first, 256, sp, loss_out=out, next_span=(256 * ((i + 1) % 5), 256))
                torch.cuda.synchronize()
                rate[mode] = (time.perf_counter() - t0) / n * 1e6
                if rep == 0:
                    res.append((eng.params.clone(), eng.exp_avg.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
        print("%s: fused %.1f us/step, dp overlapped %.1f, dp in line %.1f" % (name, rate["fused"], rate["overlap"], rate["inline"]))
    torch.cuda.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream())
    eng.comm_destroy()


def test_sharded_exchange_two_ranks_equals_allreduce_exchange(tmp_path):
    """PVAE_DP_SHARDED=1 (each rank applies Adam to its 1/N slice of every reduced bucket, the updated
    parameter slices are all-gathered): two ranks on one GPU end with bit-identical replicas that also
    equal, bit for bit, what the all-reduce + replicated-Adam exchange produces on the same schedule
    (ragged tail with an empty shard, phase switch)."""
    sharded = _run(tmp_path, 2, 16, "sharded", PVAE_DP_SHARDED="1")
    plain = _run(tmp_path, 2, 16, "plain2")
    a, b = sharded
    for k in a["sd"]:
        assert torch.equal(a["sd"][k], b["sd"][k]), k
        assert torch.equal(a["sd"][k], plain[0]["sd"][k]), k
    assert a["losses"] == plain[0]["losses"] and a["steps"] == plain[0]["steps"]


def test_sharded_exchange_through_rccl_single_rank_is_bit_identical(tmp_path):
    """The in-library sharded step (ncclReduceScatter -> Adam on the owned slice -> ncclAllGather, all on
    the compute stream) with one rank: both collectives are the identity and the slice is the whole
    bucket, so the run must equal the fused single-GPU step bit for bit."""
    rccl = _run(tmp_path, 1, 32, "rccl_sharded", PVAE_DP_ALWAYS_REDUCE="1", PVAE_TEST_BACKEND="nccl",
                PVAE_DP_TRANSPORT="rccl", PVAE_TEST_EXPECT_COMM="1", PVAE_DP_SHARDED="1")[0]
    single = _run(tmp_path, 1, 32, "plain1")[0]
    assert rccl["steps"] == single["steps"] and rccl["losses"] == single["losses"]
    for k, v in single["sd"].items():
        assert torch.equal(rccl["sd"][k], v), k


@pytest.mark.parametrize("form", ["inline", "bucketed", "sharded"])
def test_rccl_forms_at_baseline_size_equal_the_fused_step(form):
    """The three RCCL forms of pvae_dp_train_step -- ncclAllReduce in line, in 6 MiB buckets on the exchange stream,
    ncclReduceScatter + owner Adam + ncclAllGather -- at BASELINE configs[3]'s per-GPU sizes (256 rows, dims 197 / 45, 4x1024
    stacks: 14.4 MB / 28.2 MB exchanged per step) with a ONE-rank communicator, both phases, four optimizer steps with the
    gather prefetched: every call site, bucket boundary and stream hand-over the form has executes at full size, and the
    result equals the fused single-GPU step's bit for bit (one rank: the sum is the rank's own gradient, and the flat Adam
    kernel is the fused update's arithmetic).  No box with more than one GPU has been available: this is what CAN run of
    tm:131-161 on N GPUs before a node sees it."""
    import numpy as np
    from physicsvae_amd import _lib
    from physicsvae_amd.engine import make_step_params
    from physicsvae_amd.train_physics_vae import WindowDataset
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from synth_demo import make_trainer, synth_demo
    Db, Da, B, T, E, K = 197, 45, 256, 1001, 2, 4
    torch.manual_seed(1)
    tr = make_trainer(synth_demo(0, 1, 4, Db, Da), B, "cuda", width=1024, depth=4, latent=32)
    eng = tr.engine
    gen = torch.Generator(device="cuda").manual_seed(0)
    states = torch.randn(E * T, Db, generator=gen, device="cuda")
    actions = torch.randn(E * T, Da, generator=gen, device="cuda").clamp_(-3, 3)
    rows_idx = (torch.arange(E, device="cuda")[:, None] * T + torch.arange(T - 1, device="cuda")[None, :]).reshape(-1)
    ds = WindowDataset(np.zeros((2, Db), np.float32), np.zeros((2, Da), np.float32), np.zeros(1, np.int32))
    ds._dev = (states, actions, rows_idx.to(torch.int32))
    ds.window_row = np.empty(E * (T - 1), dtype=np.int8)
    eng.bind_dataset(*ds.device_arrays(eng.device))
    eng.comm_init(0, 1, eng.comm_unique_id())
    eng.comm_mode("sharded" if form == "sharded" else "allreduce")
    eng.comm_config(6.0 if form == "bucketed" else 0.0)
    start = eng.params.clone()
    for phase, world in ((_lib.PHASE_WORLD, True), (_lib.PHASE_JOINT, False)):
        res = []
        for dp in (False, True):
            eng.params.copy_(start); eng.params_changed(); eng.invalidate_staging()
            eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
            out = torch.zeros(K, 5, device="cuda")
            for t in range(K):
                sp = make_step_params(lr=5e-4, adam_t=(t + 1,) * 3, a_rec=0.0 if world else 1.0, kl=0.0 if world else 1.0,
                                      s_rec=1.0 if world else 0.0, cyc=0.0 if world else 1e-3, global_rows=B)
                sp.rng_seed, sp.rng_offset = 7, t * 65536
                nxt = ((t + 1) * B, B) if t + 1 < K else None
                (eng.dp_train_step if dp else eng.train_step)(phase, t * B, B, sp, loss_out=out[t], next_span=nxt)
            torch.cuda.synchronize()
            res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b), (form, phase)
        assert float(res[0][3][0, 0]) != float(res[0][3][K - 1, 0]) and not torch.equal(res[0][0], start)
    eng.comm_destroy()
