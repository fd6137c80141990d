"""CPU-side checks (no GPU): C-ABI library loads and exports every declared symbol, arena
layout, module/state_dict/ checkpoint layout against the reference captures, dataset
windowing against the oracle, the minibatch schedule, and the data-parallel plumbing with
a world_size-2 gloo group."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib, parallel
from physicsvae_amd import torch_models as TM
from physicsvae_amd import train_physics_vae as T
from util import arch_from_meta, make_trainer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "pvae.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # drop comments
    declared = set(re.findall(r"\b(pvae_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.pvae_abi_version() == _lib.ABI_VERSION == 12


def test_layout_queries_without_gpu():
    lib = _lib.load()
    cfg = _lib.Config(197, 45, 32, 1024, 4, 1024, 4, 1024, 4, 256, 1)
    assert lib.pvae_num_layers(C.byref(cfg)) == 15
    info = _lib.LayerInfo()
    end = 0
    for i in range(15):
        assert lib.pvae_layer(C.byref(cfg), i, C.byref(info)) == 0
        assert info.ld % 64 == 0 and info.n_out_pad % 64 == 0
        assert info.ld >= info.n_in and info.n_out_pad >= info.n_out
        assert info.w_offset == end                      # densely packed, in order
        assert info.b_offset == info.w_offset + info.n_out_pad * info.ld
        end = info.b_offset + info.n_out_pad
    assert lib.pvae_arena_floats(C.byref(cfg)) == end
    off, cnt = C.c_int64(), C.c_int64()
    tot = 0
    for net in range(3):
        assert lib.pvae_net_segment(C.byref(cfg), net, C.byref(off), C.byref(cnt)) == 0
        assert off.value == tot
        tot += cnt.value
    assert tot == end
    assert lib.pvae_workspace_bytes(C.byref(cfg)) > 0


def test_bad_config_is_an_error_not_a_crash():
    lib = _lib.load()
    cfg = _lib.Config(0, 45, 32, 1024, 4, 1024, 4, 1024, 4, 256, 1)
    assert lib.pvae_num_layers(C.byref(cfg)) < 0
    assert b"bad config" in lib.pvae_last_error()
    ctx = C.c_void_p()
    assert lib.pvae_create(C.byref(cfg), C.byref(ctx)) < 0


def test_compute_without_gpu_fails_loudly():
    data = R.synth_demo(0, 2, 20, 7, 3)
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    tr = make_trainer(arch, data, batch=8, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        tr.step()


@pytest.mark.parametrize("name", ["single_tiny", "single_c1", "single_c2", "single_default"])
def test_module_state_dict_matches_reference_layout(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"])
    tr = make_trainer(arch, data, batch, device="cpu")
    sd = tr.model.state_dict()
    assert list(sd.keys()) == list(g["sd_keys"])
    assert [list(v.shape) + [0] * (2 - v.dim()) for v in sd.values()] == g["sd_shapes"].tolist()
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])
    assert all(v.dtype == torch.float32 for v in sd.values())
    # normc init: hidden rows norm 1, output rows 0.01, bias 0 (tpv:184-189)
    for k, v in sd.items():
        if k.endswith("weight"):
            n_layers = len([q for q in sd if q.startswith(k.split(".")[0]) and q.endswith("weight")])
            std = 0.01 if int(k.split(".")[2]) == n_layers - 1 else 1.0
            np.testing.assert_allclose(v.norm(dim=1).numpy(), std, rtol=1e-4)
        else:
            assert float(v.abs().max()) == 0.0
    # loading the oracle's weights writes through to the flat arena; padding stays zero
    ref = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    tr.model.load_state_dict(ref)
    views = tr.engine.named_views()
    for k in views:
        assert torch.equal(views[k], ref[k])
    used = sum(v.numel() for k, v in ref.items() if not k.startswith("_value_branch"))
    assert int((tr.engine.params != 0).sum()) <= used
    assert tr.engine.params.abs().sum().item() == pytest.approx(
        sum(v.abs().double().sum().item() for k, v in ref.items() if not k.startswith("_value_branch")), rel=1e-5)
    # freezing state after setup (tpv:326-329)
    assert tr.model.learnable_nets() == [_lib.NET_WM]
    n_train = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    wm_vb = sum(v.numel() for k, v in ref.items() if k.startswith(("_world_model", "_value_branch")))
    assert n_train == wm_vb


def test_checkpoint_files_match_reference_and_roundtrip(golden, tmp_path):
    g = golden("single_default")
    arch = arch_from_meta(g)
    data = R.synth_demo(0, 2, 100, arch["Db"], arch["Da"])
    tr = make_trainer(arch, data, 32, device="cpu")
    ref = R.perturb_biases(R.init_state_dict(arch, 1), 3)
    tr.model.load_state_dict(ref)
    ret = tr.save_checkpoint(str(tmp_path))
    assert os.path.basename(ret) == str(g["ckpt_return_basename"])
    assert sorted(os.listdir(tmp_path)) == list(g["ckpt_files"])
    for f in g["ckpt_files"]:
        obj = torch.load(os.path.join(tmp_path, str(f)))          # weights_only default: plain tensors
        if str(f) == "task_encoder.pt":
            assert list(obj.keys()) == list(g["ckpt_te_outer_keys"])
            obj = obj["task_encoder"]
        assert list(obj.keys()) == list(g["ckpt_keys::" + str(f)])
        for v in obj.values():
            assert v.device.type == "cpu" and v.is_contiguous() and v.dtype == torch.float32
    # files load into the oracle's module (same layout as the reference's class, pinned above)
    m = R.RefModel(arch)
    m.load_state_dict(torch.load(os.path.join(tmp_path, "model.pth")))
    m._world_model.load_state_dict(torch.load(os.path.join(tmp_path, "world_model.pt")))
    m._motor_decoder.load_state_dict(torch.load(os.path.join(tmp_path, "motor_decoder.pt")))
    m._task_encoder.load_state_dict(torch.load(os.path.join(tmp_path, "task_encoder.pt"))["task_encoder"])
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k])
    # and back: restore() of a fresh trainer reproduces the weights; per-net loaders too
    tr2 = make_trainer(arch, data, 32, device="cpu")
    tr2.restore(ret)
    for k, v in tr2.model.state_dict().items():
        assert torch.equal(v, ref[k])
    tr3 = make_trainer(arch, data, 32, device="cpu")
    tr3.model.load_weights_world_model(os.path.join(tmp_path, "world_model.pt"))
    tr3.model.load_weights_motor_decoder(os.path.join(tmp_path, "motor_decoder.pt"))
    tr3.model.load_weights_task_encoder(os.path.join(tmp_path, "task_encoder.pt"))
    for k, v in tr3.model.state_dict().items():
        if not k.startswith("_value_branch"):
            assert torch.equal(v, ref[k])


def test_window_dataset_equals_oracle_windows(tmp_path):
    data = R.synth_demo(0, 3, 21, 7, 3)
    pkl = str(tmp_path / "d.pkl")
    R.write_demo(pkl, data)
    X, Y = R.build_windows(data)
    ds = T.load_dataset_for_PhysicsVAE([pkl])
    assert len(ds) == len(X) == 60
    np.testing.assert_array_equal(ds.X, X)
    np.testing.assert_array_equal(ds.Y, Y)
    x5, y5 = ds[5]
    assert x5.shape == (1, 14) and x5.dtype == torch.float32
    np.testing.assert_array_equal(x5.numpy(), X[5].astype(np.float32))
    # states are stored once: R rows instead of 2N
    assert ds.states.shape == (63, 7) and ds.states.dtype == np.float32
    # cap semantics (tpv:137-138) and multi-file merge (tpv:94-114)
    capped = T.load_dataset_for_PhysicsVAE([pkl], num_samples=25)
    np.testing.assert_array_equal(capped.X, X[:25])
    both = T.load_dataset_for_PhysicsVAE([pkl, pkl])
    assert len(both) == 120
    np.testing.assert_array_equal(both.X[60:], X)
    bad = dict(data, dim_action=99)
    pkl2 = str(tmp_path / "bad.pkl")
    R.write_demo(pkl2, bad)
    with pytest.raises(AssertionError):
        T.merge_dataset([pkl, pkl2])


def test_packed_file_roundtrip_and_trainer_config(tmp_path):
    data = R.synth_demo(0, 3, 21, 7, 3)
    pkl = str(tmp_path / "d.pkl")
    R.write_demo(pkl, data)
    ds = T.load_dataset_for_PhysicsVAE([pkl])
    pvd = str(tmp_path / "d.pvd")
    T.save_packed(ds, pvd, meta=ds.meta)
    back = T.load_dataset_for_PhysicsVAE([pvd])
    assert not back.states.flags.owndata                          # mapped, not copied
    np.testing.assert_array_equal(back.X, ds.X)
    np.testing.assert_array_equal(back.Y, ds.Y)
    assert back.meta["dim_action"] == 3 and back.meta["exp_std"] == 0.05
    # caps and multi-file merges behave like the pickle path
    X, _ = R.build_windows(data)
    np.testing.assert_array_equal(T.load_dataset_for_PhysicsVAE([pvd], num_samples=25).X, X[:25])
    np.testing.assert_array_equal(T.load_dataset_for_PhysicsVAE([pvd, pvd]).X[60:], X)
    assert T.inspect_dataset(pvd) == T.inspect_dataset(pkl) == (14, 7, 7, 3)
    # the converter CLI
    out = str(tmp_path / "cli.pvd")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pack_demo.py"), pkl, "-o", out],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    np.testing.assert_array_equal(T.load_packed(out).X, ds.X)
    with pytest.raises(ValueError):
        T.load_packed(pkl)


@pytest.mark.parametrize("L", [2, 3])
def test_lookahead_windows_equal_oracle_windows(tmp_path, golden, L):
    """tpv:131-156 with lookahead > 1: T-L windows per episode, never across an episode boundary;
    the pickle path, the packed-file path (widened from its lookahead-1 window list) and the
    loader's [B, L, .] tensors all equal the oracle's (hence the reference's) windows."""
    data = R.synth_demo(0, 3, 21, 7, 3, kind="dynamics")
    pkl = str(tmp_path / "d.pkl")
    R.write_demo(pkl, data)
    X, Y = R.build_windows(data, lookahead=L)
    ds = T.load_dataset_for_PhysicsVAE([pkl], lookahead=L)
    assert len(ds) == 3 * (21 - L) and ds.lookahead == L
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    np.testing.assert_array_equal(ds.X, f32(X))
    np.testing.assert_array_equal(ds.Y, f32(Y))
    x5, y5 = ds[5]
    assert x5.shape == (L, 14) and y5.shape == (L, 3)
    np.testing.assert_array_equal(x5.numpy(), X[5].astype(np.float32))
    pvd = str(tmp_path / "d.pvd")
    T.save_packed(T.load_dataset_for_PhysicsVAE([pkl]), pvd)
    np.testing.assert_array_equal(T.load_dataset_for_PhysicsVAE([pvd], lookahead=L).X, f32(X))
    np.testing.assert_array_equal(T.load_dataset_for_PhysicsVAE([pkl], num_samples=25, lookahead=L).X, f32(X[:25]))
    if L == 3:                                    # window / batch counts of the reference capture
        g = golden("look3_tiny")
        d2 = R.synth_demo(0, 2, 15, 7, 3, kind="dynamics")
        arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
        tr = make_trainer(arch, d2, batch=8, device="cpu", extra={"lookahead": 3})
        assert len(tr.train_loader.dataset) == int(g["n_windows"])
        assert len(tr.train_loader) == int(g["n_batches"])
        assert list(tr.train_loader.spans())[-1][1] == int(g["last_batch_size"])
        np.testing.assert_array_equal(R.tensor_digest(list(tr.train_loader)[-1][0]), g["loader_last_x_digest"])


def test_loader_schedule_is_sequential_with_partial_last_batch():
    data = R.synth_demo(0, 2, 14, 7, 3)
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    tr = make_trainer(arch, data, batch=8, device="cpu")
    X, Y = R.build_windows(data)
    ref = list(R.make_loader(X, Y, 8))
    ours = list(tr.train_loader)
    assert len(tr.train_loader) == len(ref) == 4
    assert list(tr.train_loader.spans()) == [(0, 8), (8, 8), (16, 8), (24, 2)]
    for (xa, ya), (xb, yb) in zip(ours, ref):
        assert torch.equal(xa, xb) and torch.equal(ya, yb)


def test_shuffled_loader_draws_the_reference_order(golden):
    """`shuffle_data: True` (tm:166-175, 181; upstream's config spells the key "suffle_data", so only a user who fixes it
    gets here): every pass over the loader is a fresh permutation drawn exactly as torch's DataLoader + RandomSampler draw
    it -- under the seed of the capture, the five epochs' sample orders equal the indices the REFERENCE's dataset was asked
    for (tests/golden/shuffle_tiny.npz), and the host-side batches are those windows."""
    g = golden("shuffle_tiny")
    n_ep, n_steps, batch, m_world, n_epochs, seed = [int(v) for v in g["meta"][9:15]]
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    data = R.synth_demo(0, n_ep, n_steps, 7, 3, kind="dynamics")
    tr = make_trainer(arch, data, batch=batch, device="cpu", extra={"shuffle_data": True})
    ld = tr.train_loader
    assert ld.shuffle and len(ld) == 8 and list(ld.spans())[-1] == (56, 4)
    X, Y = R.build_windows(data)
    torch.manual_seed(seed)
    for e in range(n_epochs):
        if e % 2 == 0:
            order = ld.new_epoch()                                   # what run_epoch does
        else:
            batches = list(ld)                                       # the host-side iterator draws the same way
            order = ld.order
            assert torch.equal(torch.cat([b[0] for b in batches]), torch.from_numpy(X[order.numpy()]).float())
            assert torch.equal(torch.cat([b[1] for b in batches]), torch.from_numpy(Y[order.numpy()]).float())
        np.testing.assert_array_equal(order.numpy(), g["order_epoch%d" % e])
        assert sorted(order.tolist()) == list(range(60))
    seq = make_trainer(arch, data, batch=batch, device="cpu").train_loader
    assert not seq.shuffle and seq.new_epoch() is None


def test_phase_machine_and_adam_counters():
    data = R.synth_demo(0, 2, 14, 7, 3)
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    tr = make_trainer(arch, data, batch=8, m_world=2, device="cpu")
    assert tr.phase() == (_lib.PHASE_WORLD, [_lib.NET_WM])
    assert (tr.s_rec_coeff, tr.a_rec_coeff, tr.vae_kl_coeff, tr.vae_cycle_coeff) == (1.0, 0.0, 0.0, 0.0)
    sp = tr.step_params([_lib.NET_WM], 8, True)
    assert list(sp.adam_t)[:3] == [1, 1, 1] and tr.optimizer.net_steps[_lib.NET_WM] == 1
    tr.iter = 2                                   # as if two world epochs had run
    tr.model.set_learnable_task_encoder(True)
    tr.model.set_learnable_motor_decoder(True)
    tr.model.set_learnable_world_model(False)
    tr.read_loss_fn_coeff(world=False)
    assert tr.phase() == (_lib.PHASE_JOINT, [_lib.NET_TE, _lib.NET_MD])
    assert (tr.s_rec_coeff, tr.a_rec_coeff, tr.vae_kl_coeff, tr.vae_cycle_coeff) == (0.0, 1.0, 1.0, 1e-3)
    sp = tr.step_params([_lib.NET_TE, _lib.NET_MD], 8, True)
    assert list(sp.adam_t)[:3] == [1, 1, 1]       # TE/MD start at t = 1, WM stays where it was
    sp = tr.step_params([_lib.NET_TE, _lib.NET_MD], 8, True)
    assert list(sp.adam_t)[:3] == [2, 2, 1]
    # StepLR drives HipAdam through param_groups, once per epoch
    sched = TM.get_lr_scheduler(tr.optimizer, "step", {"step_size": 2, "gamma": 0.7})
    lrs = []
    for _ in range(5):
        lrs.append(tr.optimizer.lr)
        sched.step()
    np.testing.assert_allclose(lrs, [R.lr_for_epoch(e, step_size=2) for e in range(1, 6)], rtol=1e-12)
    assert TM.get_lr_scheduler(tr.optimizer, None, None) is None
    assert TM.get_lr_scheduler(tr.optimizer, "cosine", {"T_max": 10}) is not None


def test_unsupported_configs_are_refused():
    data = R.synth_demo(0, 2, 14, 7, 3)
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    with pytest.raises(NotImplementedError):
        make_trainer(arch, data, 8, device="cpu", extra={"latent_prior_type": "von_mises_fisher"})     # rmt:624-625
    with pytest.raises(NotImplementedError):
        make_trainer(arch, data, 8, device="cpu", extra={"loss": "CrossEntropy"})
    assert make_trainer(arch, data, 8, device="cpu", extra={"loss": "L1"}).loss_name == "L1"


def test_shard_arithmetic_matches_reference_order():
    n, b = 1000, 64
    for world in (1, 2, 4, 8):
        seen = []
        steps = parallel.DataParallel(0, world).global_steps(n, b)
        for g in range(steps):
            tot = 0
            for r in range(world):
                first, rows, grows = parallel.DataParallel(r, world).shard(g, n, b)
                seen.extend(range(first, first + rows))
                tot += rows
            assert tot == grows
        assert seen == list(range(n))             # every window once, in sequential order


DP_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from physicsvae_amd import parallel
rank, world, _ = parallel.init_from_env(backend="gloo")
dp = parallel.DataParallel.from_env()
assert (dp.rank, dp.world) == (rank, 2)
# each rank contributes sum-scaled shard gradients; SUM all-reduce == full-batch mean gradient
torch.manual_seed(0)
per_sample = torch.randn(10, 33)                 # per-sample gradient contributions
first, rows, grows = dp.shard(0, 10, 6)          # ragged: 6 + 4
local = per_sample[first:first + rows].sum(0) / grows
dp.all_reduce(local)
assert torch.allclose(local, per_sample.mean(0), atol=1e-6), (local - per_sample.mean(0)).abs().max()
first, rows, grows = dp.shard(1, 13, 6)          # second global batch: rank 0 has 1 row, rank 1 none
assert (rows, grows) == ((1, 1) if rank == 0 else (0, 1))
dist.barrier()
print("OK", rank)
'''


def test_data_parallel_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "OK %d" % r in o


AUTOTUNE_WORKER = r'''
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from physicsvae_amd import parallel
rank, world, _ = parallel.init_from_env(backend="gloo")

class Engine:                                   # what autotune_exchange touches of an engine, on the CPU
    device = torch.device("cpu")
    def __init__(self):
        self.params, self.exp_avg, self.exp_avg_sq = torch.arange(8.), torch.zeros(8), torch.ones(8)
        self.form, self.has_p2p, self.closed, self.cleared = None, False, 0, 0
    def invalidate_staging(self): pass
    def p2p_status(self): return (0, 0, 0)
    def p2p_close(self): self.has_p2p = False; self.closed += 1
    def p2p_clear_errors(self): self.cleared += 1

class DP(parallel.DataParallel):                # three "forms": fast everywhere / slow on rank 1 / breaks the replicas
    def set_exchange_form(self, engine, form):
        if form == "bucketed":
            return "not here"
        engine.form = form
        if form.startswith("p2p"):
            engine.has_p2p = True                # (attach_p2p maps the peers again after a close)
        return None

eng, dp = Engine(), DP(rank, world)
cost = {"inline": (0.012, 0.012), "sharded": (0.001, 0.012), "p2p": (0.0005, 0.0005), "p2p_push": (0.002, 0.003)}
def run_steps(n):
    for i in range(n):
        if eng.form == "sharded" and rank == 1 and n > 1 and i == 2:
            raise RuntimeError("boom on rank 1 only")                        # a rank-local failure in the timed steps
        time.sleep(cost[eng.form][rank])
        eng.params += 1.0 if eng.form != "p2p" else float(rank + 1)          # "p2p" lets the replicas drift apart
        eng.exp_avg += 0.5
chosen, report = dp.autotune_exchange(eng, run_steps, steps=5, warm=1)
assert chosen == "p2p_push", (chosen, report)                                # fastest by the SLOWEST rank among the valid ones
assert report["bucketed"] == {"skipped": "not here"} and report["p2p"]["replicas_identical"] is False
assert report["inline"]["us_per_step"] > report["p2p_push"]["us_per_step"]
# the rank-local failure is agreed on by BOTH ranks (nobody is left waiting in a barrier), the candidate is dropped
assert "error" in report["sharded"] and ("boom" in report["sharded"]["error"]) == (rank == 1), report["sharded"]
# the peer-mapped candidate whose replicas diverged was torn down (error word cleared, peers unmapped) and the winner
# attached again from clean flags
assert eng.closed == 1 and eng.cleared >= 2 and eng.has_p2p
assert torch.equal(eng.params, torch.arange(8.)) and torch.equal(eng.exp_avg, torch.zeros(8))   # state restored
assert eng.form == "p2p_push"
dist.barrier()
print("OK", rank)
'''


def test_exchange_autotune_decides_alike_on_every_rank_gloo_world2(tmp_path):
    """`DataParallel.autotune_exchange` with two gloo ranks and a stand-in engine: forms that are unavailable are
    skipped, a form whose replicas diverge is disqualified and torn down, a failure on ONE rank drops the candidate on
    both without a hang, the slowest rank's time decides (so both ranks choose the same form), and the snapshot of
    parameters and moments is restored."""
    script = tmp_path / "autotune_worker.py"
    script.write_text(AUTOTUNE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "OK %d" % r in o


def test_graft_entry_build_runs_without_gpu():
    """The driver's "does it build" check: compiles (or finds fresh) the gfx950 library, loads it,
    checks the ABI version and imports the package -- all without a GPU."""
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "built" in r.stdout


def test_bench_algorithmic_flops_match_the_survey():
    """the per-sample work `roofline.achieved` is priced with."""
    sys.path.insert(0, ROOT)
    import bench
    w, j = bench.algorithmic_flops_per_sample(197, 45, 32, 1024, 4)
    assert (w, j) == (2 * 10_537_984, 2 * 27_506_688)
    w5, j5 = bench.algorithmic_flops_per_sample(400, 90, 32, 1024, 4)
    assert (w5, j5) == (2 * 11_669_504, 2 * 29_607_936)
    assert bench.PEAK_F32_MFMA_TFLOPS == pytest.approx(256 * 2.4e9 * 256 / 1e12, rel=1e-3)


def test_constructor_reproduces_the_reference_initialisation_bit_for_bit(golden):
    """Same torch seed -> same weights as the reference's own constructor: the stacks are built in the
    reference's order (task encoder, motor decoder, world model, value branch) and `normc_` consumes
    the torch RNG exactly like ray's normc_initializer, so a user who seeds torch gets the run they
    would have got upstream.  (Digests of the reference's state dict under torch.manual_seed(0).)"""
    g = golden("anchor_c1")
    arch = R.make_arch(197, 45, latent=32, te=(256, 2), md=(256, 2), wm=(256, 2))
    data = R.synth_demo(0, 1, 8, 197, 45)                      # dims only; the init does not see the data
    torch.manual_seed(0)
    tr = make_trainer(arch, data, 64, m_world=2, device="cpu")
    sd = tr.model.state_dict()
    assert list(sd.keys()) == list(g["sd_keys"])
    for k, v in sd.items():
        np.testing.assert_array_equal(R.tensor_digest(v.cpu()), g["init_digest::" + k], err_msg=k)


def test_checkpoint_interop_with_the_reference_both_directions(golden, tmp_path):
    """(a) The five files the REFERENCE wrote (stored byte for byte in the fixture) load through our
    restore / load_weights* and give exactly the weights it held.  (b) The fixture was only written
    after the reference's own load_checkpoint / load_weights / per-net loaders had accepted OUR five
    files and reproduced our weights (asserted in oracle/gen_golden.py, flags recorded); the forward
    pass the reference then computed is reproduced by the oracle from those weights."""
    g = golden("ckpt_interop_tiny")
    arch = arch_from_meta(g)
    data = R.synth_demo(0, 2, 14, arch["Db"], arch["Da"], kind="iid")
    names = sorted(k.split("::")[1] for k in g.files if k.startswith("ref_file::"))
    assert names == ["model.pt", "model.pth", "motor_decoder.pt", "task_encoder.pt", "world_model.pt"]
    for f in names:
        (tmp_path / f).write_bytes(g["ref_file::" + f].tobytes())
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)           # what the reference held
    tr = make_trainer(arch, data, 8, device="cpu")
    tr.restore(str(tmp_path / "model.pth"))
    assert all(torch.equal(v, sd[k]) for k, v in tr.model.state_dict().items())
    tr2 = make_trainer(arch, data, 8, device="cpu")
    tr2.model.load_weights(str(tmp_path / "model.pt"))
    assert all(torch.equal(v, sd[k]) for k, v in tr2.model.state_dict().items())
    tr3 = make_trainer(arch, data, 8, device="cpu")
    tr3.model.load_weights_task_encoder(str(tmp_path / "task_encoder.pt"))
    tr3.model.load_weights_motor_decoder(str(tmp_path / "motor_decoder.pt"))
    tr3.model.load_weights_world_model(str(tmp_path / "world_model.pt"))
    assert all(torch.equal(v, sd[k]) for k, v in tr3.model.state_dict().items() if not k.startswith("_value_branch"))
    # (b)
    for k in ("load_checkpoint", "load_weights", "per_net_loaders"):
        assert bool(g["reference_accepts::" + k])
    sd2 = R.perturb_biases(R.init_state_dict(arch, seed=7), seed=9)
    m = R.RefModel(arch)
    m.load_state_dict(sd2)
    m.eps_source = lambda shape: torch.zeros(shape)
    with torch.no_grad():
        logits = m(torch.from_numpy(g["obs"]))
    np.testing.assert_allclose(logits.numpy(), g["reference_logits_after_loading_our_files"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.cur_future_state.numpy(), g["reference_future_state"], rtol=1e-6, atol=1e-7)


def test_multi_file_merge_and_num_data_cap_match_the_reference(golden, tmp_path):
    """Two --data_train files (tpv:94-114) and the --num_data cap (tpv:137-138), captured from the
    reference: window / batch counts and the loader's first and last minibatches -- for the pickle
    path and for packed .pvd files."""
    g = golden("ingest_tiny")
    arch = arch_from_meta(g)
    d1 = R.synth_demo(0, 2, 14, arch["Db"], arch["Da"], kind="iid")
    d2 = R.synth_demo(5, 3, 11, arch["Db"], arch["Da"], kind="iid")
    p1, p2 = str(tmp_path / "a.pkl"), str(tmp_path / "b.pkl")
    R.write_demo(p1, d1)
    R.write_demo(p2, d2)
    q1, q2 = str(tmp_path / "a.pvd"), str(tmp_path / "b.pvd")
    T.save_packed(T.load_dataset_for_PhysicsVAE([p1]), q1, meta={k: d1[k] for k in T.META_KEYS})
    T.save_packed(T.load_dataset_for_PhysicsVAE([p2]), q2, meta={k: d2[k] for k in T.META_KEYS})
    for files in ([p1, p2], [q1, q2]):
        for tag, num in (("all", None), ("cap", 37)):
            ds = T.load_dataset_for_PhysicsVAE(files, num_samples=num)
            loader = TM.WindowLoader(ds, 8)
            batches = list(loader)
            assert len(ds) == int(g[tag + "_n_windows"]) and len(loader) == int(g[tag + "_n_batches"])
            assert batches[-1][0].shape[0] == int(g[tag + "_last_batch_size"])
            for b, nm in ((0, "first"), (len(batches) - 1, "last")):
                np.testing.assert_array_equal(R.tensor_digest(batches[b][0]), g["%s_%s_x_digest" % (tag, nm)])
                np.testing.assert_array_equal(R.tensor_digest(batches[b][1]), g["%s_%s_y_digest" % (tag, nm)])
    d3 = dict(d2, dim_action=arch["Da"] + 1)                          # meta mismatch is refused like upstream
    p3 = str(tmp_path / "c.pkl")
    R.write_demo(p3, d3)
    with pytest.raises(AssertionError):
        T.load_dataset_for_PhysicsVAE([p1, p3])


def test_pretrained_weight_options_of_the_constructor(tmp_path):
    """rmt:709-727: `load_weights`, `*_load_weights` keys of custom_model_config (the trainer's
    --world_model flag feeds world_model_load_weights, tpv:247) load at construction."""
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    data = R.synth_demo(0, 2, 14, 7, 3)
    src = make_trainer(arch, data, 8, device="cpu")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=11), seed=12)
    src.model.load_state_dict(sd)
    src.save_checkpoint(str(tmp_path))
    # --world_model: only the world model comes from the file
    td = str(tmp_path / "d.pkl")
    R.write_demo(td, data)
    argv = ["--data_train", td, "--batch_size", "8", "--max_iter_world_model", "0", "--latent_dim", "4",
            "--TE_width", "16", "--TE_depth", "2", "--MD_width", "24", "--MD_depth", "2",
            "--world_model_width", "32", "--world_model_depth", "2", "--world_model", str(tmp_path / "world_model.pt")]
    T.args = T.arg_parser().parse_args(argv)
    cfg = T.get_trainer_config(T.args)
    cfg["model"]["custom_model_config"]["device"] = "cpu"
    tr = T.TrainModel(cfg)
    got = tr.model.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in got if k.startswith("_world_model"))
    assert not all(torch.equal(got[k], sd[k]) for k in got if k.startswith("_task_encoder"))
    # load_weights: everything; per-net keys with their learnable flags
    cfg2 = T.get_trainer_config(T.args)
    cmc = cfg2["model"]["custom_model_config"]
    cmc.update(device="cpu", world_model_load_weights=None, load_weights=str(tmp_path / "model.pt"),
               task_encoder_load_weights=str(tmp_path / "task_encoder.pt"), task_encoder_learnable=False)
    tr2 = T.TrainModel(cfg2)
    assert all(torch.equal(v, sd[k]) for k, v in tr2.model.state_dict().items())


# ------------------------------------------------------------------------------------------
# tune stand-in: grid trials, rank-gated logs / checkpoints
# ------------------------------------------------------------------------------------------
class _CountingTrainable(__import__("physicsvae_amd.tune", fromlist=["Trainable"]).Trainable):
    def setup(self, config):
        self.scale = config["scale"] * config["shift"]

    def step(self):
        return {"mean_train_loss": self.scale / (self.training_iteration + 1), "mean_test_loss": 0.0}

    def save_checkpoint(self, checkpoint_dir):
        path = os.path.join(checkpoint_dir, "model.pth")
        torch.save({"scale": self.scale}, path)
        return path


def test_tune_run_executes_every_grid_trial(tmp_path):
    """`--vae_kl_coeff 0.1` in the reference sweeps [1.0, 0.1] (tpv:51-53): tune.run executes the
    cartesian product of the grid leaves as consecutive trials and the analysis picks the best."""
    from physicsvae_amd import tune
    cfg = {"scale": tune.grid_search([3.0, 1.0]), "shift": tune.grid_search([2.0]), "plain": 7}
    assert [(c["scale"], c["shift"]) for c in tune.expand_grid(cfg)] == [(3.0, 2.0), (1.0, 2.0)]
    an = tune.run(_CountingTrainable, config=cfg, stop={"training_iteration": 3}, checkpoint_freq=2,
                  checkpoint_at_end=True, local_dir=str(tmp_path), name="t", verbose=0)
    assert len(an.trials) == 2 and all(len(t.results) == 3 for t in an.trials)
    assert [len(t.checkpoints) for t in an.trials] == [2, 2]          # iteration 2 and the end (3)
    best = an.get_best_logdir(metric="mean_train_loss", mode="min")
    assert best == an.trials[1].logdir
    ck = an.get_best_checkpoint(logdir=best)
    assert torch.load(ck)["scale"] == 2.0
    assert len(open(os.path.join(best, "result.json")).read().splitlines()) == 3


def test_cli_coefficient_flags(tmp_path):
    """The reference's append-to-default flags keep their sweep meaning; the single-value flags
    (ours) run one trial with the value given."""
    from physicsvae_amd import tune
    pkl = str(tmp_path / "d.pkl")
    R.write_demo(pkl, R.synth_demo(0, 2, 14, 7, 3))
    a = T.arg_parser().parse_args(["--data_train", pkl, "--vae_kl_coeff", "0.1"])
    cfg = T.get_trainer_config(a)
    assert cfg["vae_kl_coeff"] == {"grid_search": [1.0, 0.1]} and cfg["vae_cycle_coeff"] == {"grid_search": [1e-3]}
    assert len(tune.expand_grid(cfg)) == 2
    a = T.arg_parser().parse_args(["--data_train", pkl, "--kl_coeff", "0.1", "--cycle_coeff", "0.01"])
    cfg = T.get_trainer_config(a)
    assert cfg["vae_kl_coeff"] == {"grid_search": [0.1]} and cfg["vae_cycle_coeff"] == {"grid_search": [0.01]}
    assert len(tune.expand_grid(cfg)) == 1
    # a fresh parser still has the reference's defaults (argparse must not have mutated them)
    a = T.arg_parser().parse_args(["--data_train", pkl])
    assert a.vae_kl_coeff == [1.0] and a.vae_cycle_coeff == [1e-3]


TUNE_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from physicsvae_amd import parallel, tune
rank, world, _ = parallel.init_from_env(backend="gloo")
class Tr(tune.Trainable):
    def setup(self, config): self.k = config["k"]
    def step(self): return {"mean_train_loss": float(self.k), "mean_test_loss": 0.0}
    def save_checkpoint(self, d):
        p = os.path.join(d, "model.pth"); torch.save({"k": self.k, "writer": dist.get_rank()}, p); return p
an = tune.run(Tr, config={"k": tune.grid_search([2, 5])}, stop={"training_iteration": 2}, checkpoint_at_end=True,
              local_dir=sys.argv[2], name="job", verbose=0)
ck = an.get_best_checkpoint(logdir=an.get_best_logdir("mean_train_loss", "min"))
got = torch.load(ck)
assert got == {"k": 2, "writer": 0}, got
print("CKPT", rank, ck)
dist.barrier()
'''


def test_tune_run_under_two_ranks_writes_once(tmp_path):
    script = tmp_path / "tune_worker.py"
    script.write_text(TUNE_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29537", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path / "out")],
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    paths = {o.strip().splitlines()[-1].split(" ", 2)[2] for o in outs}
    assert len(paths) == 1                                        # both ranks agree on ONE checkpoint
    dirs = os.listdir(tmp_path / "out" / "job")
    assert len(dirs) == 2, dirs                                   # one directory per trial, not per rank


def test_module_identity_moves_on_cpu():
    data = R.synth_demo(0, 2, 14, 7, 3)
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    m = make_trainer(arch, data, 8, device="cpu").model
    assert m.to("cpu") is m and m.float() is m and m.cpu() is m
    with pytest.raises(RuntimeError):
        m.double()


# ------------------------------------------------------------------------------------------
# latent priors other than N(0, I) (specified in oracle/refpath.py: the reference crashes on them)
# ------------------------------------------------------------------------------------------
def test_prior_kinds_layout_and_state_dict():
    lib = _lib.load()
    base = (7, 3, 4, 16, 2, 24, 2, 32, 2, 8, 1)
    n0 = lib.pvae_num_layers(C.byref(_lib.Config(*base, 0, 0, 0)))
    assert lib.pvae_num_layers(C.byref(_lib.Config(*base, 1, 12, 1))) == n0 + 2          # + prior stack 7 -> 12 -> 4
    assert lib.pvae_num_layers(C.byref(_lib.Config(*base, 2, 0, 0))) == n0
    assert lib.pvae_num_layers(C.byref(_lib.Config(*base, 3, 0, 0))) == n0               # no prior (rmt:622-623)
    assert lib.pvae_num_layers(C.byref(_lib.Config(*base, 4, 0, 0))) < 0                 # unknown kind
    assert lib.pvae_num_layers(C.byref(_lib.Config(7, 3, 4, 16, 2, 24, 2, 32, 2, 8, 2, 1, 12, 1))) < 0   # needs lookahead 1
    assert b"lookahead" in lib.pvae_last_error()
    # arena order TE | MD | PR | WM, densely packed; the hypersphere encoder emits Z values, not 2Z
    cfg = _lib.Config(*base, 1, 12, 1)
    info, end, nets = _lib.LayerInfo(), 0, []
    for i in range(n0 + 2):
        assert lib.pvae_layer(C.byref(cfg), i, C.byref(info)) == 0
        assert info.w_offset == end
        end = info.b_offset + info.n_out_pad
        nets.append(info.net)
    assert nets == [0] * 3 + [1] * 3 + [3] * 2 + [2] * 3
    cfg2 = _lib.Config(*base, 2, 0, 0)
    assert lib.pvae_layer(C.byref(cfg2), 2, C.byref(info)) == 0 and (info.net, info.n_out) == (0, 4)
    data = R.synth_demo(0, 2, 14, 7, 3)
    for prior in R.PRIORS[1:]:
        arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2), prior=prior, pr=(12, 1))
        tr = make_trainer(arch, data, 8, device="cpu")
        sd = tr.model.state_dict()
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == R.state_dict_spec(arch)
        # the reference-shaped state dict round-trips through the arena views
        ref_sd = R.perturb_biases(R.init_state_dict(arch, 1), 3)
        tr.model.load_state_dict(ref_sd)
        for k, v in tr.model.state_dict().items():
            assert torch.equal(v.cpu(), ref_sd[k]), k
        # phase machine: the learned prior follows the encoder
        want_joint = [_lib.NET_TE, _lib.NET_MD] + ([_lib.NET_PR] if prior == R.PRIORS[1] else [])
        assert tr.phase() == (_lib.PHASE_WORLD, [_lib.NET_WM])
        tr.iter = tr.max_iter_world_model = 0
        tr.model.set_learnable_task_encoder(True); tr.model.set_learnable_motor_decoder(True)
        tr.model.set_learnable_world_model(False); tr.model.set_learnable_latent_prior(True)
        assert tr.phase() == (_lib.PHASE_JOINT, want_joint)
        # tpv:440-467: five files, and a sixth -- latent_prior.pt -- exactly when the model has a learned prior mean
        import tempfile
        d = tempfile.mkdtemp()
        tr.save_checkpoint(d)
        five = ["model.pt", "model.pth", "motor_decoder.pt", "task_encoder.pt", "world_model.pt"]
        if prior == R.PRIORS[1]:
            assert sorted(os.listdir(d)) == ["latent_prior.pt"] + five
            got = torch.load(os.path.join(d, "latent_prior.pt"))
            want = {k[len("_latent_prior."):]: v for k, v in ref_sd.items() if k.startswith("_latent_prior.")}
            assert list(got) == list(want) and all(torch.equal(got[k], want[k]) for k in want)
            tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, 2), 5))
            tr.model.load_weights_latent_prior(os.path.join(d, "latent_prior.pt"))       # rmt:925-928
            assert all(torch.equal(tr.model.state_dict()["_latent_prior." + k].cpu(), want[k]) for k in want)
        else:
            assert sorted(os.listdir(d)) == five


def test_cond_rel_windows_match_the_reference_capture(golden, tmp_path):
    """load_dataset_for_PhysicsVAE(cond="rel") (tpv:149-150: x = [s_t | s_{t+1} - s_t], float64 subtraction,
    then the Dataset's float32 copy): our packed-once dataset (states + a row-aligned `next_states` array)
    yields the reference's own windows bit for bit, at lookahead 1 and 2; the oracle's build_windows too."""
    g = golden("ingest_rel_tiny")
    arch = arch_from_meta(g)
    data = R.synth_demo(3, 3, 12, arch["Db"], arch["Da"], kind="iid", quantum=0.0)
    pkl = str(tmp_path / "a.pkl")
    R.write_demo(pkl, data)
    for L in (1, 2):
        ds = T.load_dataset_for_PhysicsVAE([pkl], lookahead=L, cond="rel")
        assert len(ds) == int(g["L%d_n_windows" % L])
        for i, nm in ((0, "first"), (len(ds) - 1, "last")):
            x, y = ds[i]
            np.testing.assert_array_equal(x.numpy(), g["L%d_%s_x" % (L, nm)])
            np.testing.assert_array_equal(y.numpy(), g["L%d_%s_y" % (L, nm)])
        X, Y = R.build_windows(data, lookahead=L, cond="rel")
        np.testing.assert_array_equal(R.tensor_digest(torch.from_numpy(X)), g["L%d_X_digest" % L])
        np.testing.assert_array_equal(R.tensor_digest(torch.from_numpy(Y)), g["L%d_Y_digest" % L])
        np.testing.assert_array_equal(torch.from_numpy(X).float().numpy(), ds.X.astype(np.float32))
        # the states themselves are still stored once; only the differences are an extra array
        assert ds.states.shape == ds.next_states.shape == (36, arch["Db"])
    with pytest.raises(NotImplementedError):
        T.load_dataset_for_PhysicsVAE([pkl], cond="delta")
    with pytest.raises(NotImplementedError):
        T.save_packed(ds, str(tmp_path / "a.pvd"))


def test_bench_classifies_the_kernels_of_a_committed_trace():
    """bench.py files rocprofv3 kernel names under the library's profiler categories by name; the committed
    round-2 kernel stats must all land where the HIP-event pass puts them (a renamed kernel would silently
    drop out of `roofline`)."""
    import csv
    sys.path.insert(0, ROOT)
    import bench
    seen = {}
    with open(os.path.join(ROOT, "profiles", "r02_kernel_stats_joint.csv"), newline="") as f:
        for r in csv.DictReader(f):
            if "pvae::" in r["Name"]:
                seen[r["Name"].split("(")[0]] = bench._cat_of(r["Name"])
    assert any(c == 3 for c in seen.values()) and any(c == 0 for c in seen.values())
    for name, c in seen.items():
        if "bwd_pair_kernel" in name:
            assert c == 3, name
        elif "wgrad_pair_kernel" in name or "gemm_wgrad_reg_kernel" in name:
            assert c == 2, name
        elif "gemm_splitk_reg16_kernel<true," in name:
            assert c == 5, name                  # the narrow layers on 16x16 tiles: a kernel (and a category) of their own
        elif "<true," in name:
            assert c == 0, name
        elif "<false," in name:
            assert c == 1, name
    assert bench._cat_of("reparam_kernel(float const*, int)") is None
    a = bench.parse_args(["--gpus", "2", "--steps", "7"])
    assert (a.gpus, a.steps, a.phase, a.config) == (2, 7, "joint", "c2")


def test_tune_run_resume_continues_from_the_newest_checkpoint(tmp_path):
    """`--resume` (tpv:500): a second tune.run on the same experiment name picks the trial directory up at
    its newest checkpoint -- weights through `restore`, the iteration counter from the directory name --
    and runs on to the stop criterion, appending to result.json."""
    from physicsvae_amd import tune

    class Tr(_CountingTrainable):
        def load_checkpoint(self, path):
            self.scale = torch.load(path)["scale"] + 100.0         # visible proof that restore ran

    cfg = {"scale": 3.0, "shift": 2.0}
    a1 = tune.run(Tr, config=cfg, stop={"training_iteration": 2}, checkpoint_at_end=True, local_dir=str(tmp_path),
                  name="exp", verbose=0)
    assert len(a1.results) == 2 and a1.checkpoints[-1].endswith(os.path.join("checkpoint_000002", "model.pth"))
    a2 = tune.run(Tr, config=cfg, stop={"training_iteration": 5}, checkpoint_at_end=True, local_dir=str(tmp_path),
                  name="exp", verbose=0, resume=True)
    assert a2.logdir == a1.logdir and [r["training_iteration"] for r in a2.results] == [3, 4, 5]
    assert a2.results[0]["mean_train_loss"] == pytest.approx(106.0 / 3)          # restored scale 6 + 100, iteration 3
    assert len(open(os.path.join(a2.logdir, "result.json")).read().splitlines()) == 5
    assert os.listdir(tmp_path / "exp") == [os.path.basename(a1.logdir)]


def test_python_constants_mirror_the_header():
    """ABI version, exchange-mode codes and the peer-mapped exchange's blob size / rank limit: the numbers the ctypes
    binding hard-codes are the header's."""
    header = open(os.path.join(ROOT, "include", "pvae.h")).read()
    defs = dict(re.findall(r"#define\s+(PVAE_[A-Z0-9_]+)\s+(\d+)", header))
    assert int(defs["PVAE_ABI_VERSION"]) == _lib.ABI_VERSION
    assert int(defs["PVAE_P2P_BLOB_BYTES"]) == _lib.P2P_BLOB_BYTES and int(defs["PVAE_P2P_MAX_RANKS"]) == _lib.P2P_MAX_RANKS
    enums = dict((k, int(v)) for k, v in re.findall(r"(PVAE_EXCHANGE_[A-Z0-9_]+)\s*=\s*(\d+)", header))
    assert enums == {"PVAE_EXCHANGE_ALLREDUCE": _lib.EXCHANGE_ALLREDUCE, "PVAE_EXCHANGE_SHARDED": _lib.EXCHANGE_SHARDED,
                     "PVAE_EXCHANGE_P2P": _lib.EXCHANGE_P2P, "PVAE_EXCHANGE_LOCAL": _lib.EXCHANGE_LOCAL,
                     "PVAE_EXCHANGE_P2P_PUSH": _lib.EXCHANGE_P2P_PUSH}


def test_exchange_choice_is_validated_before_any_gpu_work():
    """`dp_exchange` / PVAE_DP_EXCHANGE: an unknown form is refused by name (every rank must choose the same, so a typo
    must not fall through to the default)."""
    src = open(os.path.join(ROOT, "physicsvae_amd", "torch_models.py")).read()
    assert 'not in (None, "inline", "bucketed", "sharded", "p2p", "p2p_push", "auto")' in src
    from physicsvae_amd import parallel
    dp = parallel.DataParallel(0, 1)
    assert not dp.collective and dp.attach_p2p(type("E", (), {"ctx": None, "has_p2p": False})()) is False


def test_tune_resume_picks_the_newest_run_of_an_experiment(tmp_path):
    """An experiment directory accumulates one group of trial directories per run (default --name): resume continues
    the NEWEST run, not the first ever made, and never mixes single- and multi-trial stamps."""
    from physicsvae_amd import tune
    exp = tmp_path / "exp"
    for d in ("trial_20260101_000000", "trial_20260301_120000_00", "trial_20260301_120000_01",
              "trial_20260201_000000_00", "other"):
        os.makedirs(exp / d)
    assert tune._newest_run(str(exp)) == ["trial_20260301_120000_00", "trial_20260301_120000_01"]
    os.makedirs(exp / "trial_20260401_000000")
    assert tune._newest_run(str(exp)) == ["trial_20260401_000000"]
    assert tune._newest_run(str(tmp_path)) == []


def test_no_prior_mode_layout_matches_the_reference_capture(golden):
    """latent_prior_type = False (rmt:622-623): the reference builds an encoder with Z outputs; same keys and
    shapes here, and its state dict round-trips through the arena views."""
    g = golden("noprior_tiny")
    arch = dict(arch_from_meta(g), prior=False)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    tr = make_trainer(arch, data, batch, device="cpu")
    sd = tr.model.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["sd_keys"]]
    for (k, v), shp in zip(sd.items(), g["sd_shapes"]):
        assert list(v.shape) == [int(x) for x in shp[: v.dim()]], k
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == R.state_dict_spec(arch)
    assert tr.latent_prior_type is False


# ------------------------------------------------------------------------------------------
# trainer-config keys a user edits in the dict: "act_fn" (tpv:262) and "weight_decay" (tpv:253)
# ------------------------------------------------------------------------------------------
def test_act_fn_and_weight_decay_reach_the_model_and_the_step_params():
    lib = _lib.load()
    base = (7, 3, 4, 16, 2, 24, 2, 32, 2, 8, 1, 0, 0, 0)
    n0 = lib.pvae_num_layers(C.byref(_lib.Config(*base)))                      # a zeroed act_kind = relu
    for kind in _lib.ACT_KINDS.values():
        assert lib.pvae_num_layers(C.byref(_lib.Config(*base, kind))) == n0
    assert lib.pvae_num_layers(C.byref(_lib.Config(*base, 4))) < 0 and b"act_kind" in lib.pvae_last_error()
    data = R.synth_demo(0, 2, 14, 7, 3)
    for act, mod in (("tanh", torch.nn.Tanh), ("sigmoid", torch.nn.Sigmoid), ("elu", torch.nn.ELU)):
        arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2), act=act)
        tr = make_trainer(arch, data, 8, device="cpu", extra={"weight_decay": 0.01})
        assert tr.engine.cfg.act_kind == _lib.ACT_KINDS[act]
        for net in (tr.model._task_encoder, tr.model._motor_decoder, tr.model._world_model):
            assert isinstance(net._model[0]._model[1], mod) and len(net._model[2]._model) == 1     # linear output layer
        assert isinstance(tr.model._value_branch._model[0]._model[1], torch.nn.ReLU)               # value_fn_layers: untouched
        assert [(k, tuple(v.shape)) for k, v in tr.model.state_dict().items()] == R.state_dict_spec(arch)
        assert tr.step_params([_lib.NET_WM], 8, True).weight_decay == pytest.approx(0.01)
    with pytest.raises(NotImplementedError):                                   # needs the pre-activation backward
        make_trainer(R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2)), data, 8, device="cpu",
                     extra={"act_fn": "swish"})
    from physicsvae_amd.model import PhysicsVAE
    from physicsvae_amd import train_physics_vae as T
    tr = make_trainer(R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2)), data, 8, device="cpu")
    cfg = dict(tr.config["model"]["custom_model_config"])
    cfg["world_model_layers"] = T.gen_layers(32, 2, act_hidden="tanh")          # a stack with its own activation:
    m = PhysicsVAE(cfg["observation_space"], cfg["action_space"], 6, {"custom_model_config": cfg}, "m")    # layer by layer
    assert isinstance(m._world_model._model[0]._model[1], torch.nn.Tanh) and isinstance(m._task_encoder._model[0]._model[1], torch.nn.ReLU)
    assert [l["act"] for l in m.engine.layers if l["net"] == _lib.NET_WM] == [_lib.ACT_KINDS["tanh"]] * 2 + [_lib.ACT_LINEAR]


def test_stacks_with_per_layer_widths_and_activations(golden):
    """FC accepts any list of fc layers, each with its own hidden_size / activation / init_weight (rmt:234-270,
    get_activation_fn rmt:30-46, get_initializer rmt:220-232); gen_layers (tpv:180-192) emits only the uniform case.
    The layout (pvae_config.layer_width / layer_act), the module tree and the state-dict keys / shapes against the
    reference's own model built from the same lists (capture single_mixed_tiny); what stays refused."""
    lib = _lib.load()
    g = golden("single_mixed_tiny")
    arch = arch_from_meta(g)
    assert arch["md"] == [(32, "elu"), (16, "relu"), (24, "sigmoid")] and arch["wm"][1] == (24, "linear")
    # --- the C layout: zeroed rows = the uniform stack; a row overrides width and activation per hidden layer
    cfg = _lib.Config(7, 3, 4, 16, 2, 32, 3, 40, 3, 8, 1, 0, 0, 0, 0)
    info = _lib.LayerInfo()

    def layers_of(c):
        out = []
        for i in range(lib.pvae_num_layers(C.byref(c))):
            _lib.check(lib.pvae_layer(C.byref(c), i, C.byref(info)))
            out.append((info.net, info.index, info.n_in, info.n_out, info.act))
        return out
    uniform = layers_of(cfg)
    assert [l[3] for l in uniform if l[0] == _lib.NET_MD] == [32, 32, 32, 3] and {l[4] for l in uniform} == {0, _lib.ACT_LINEAR}
    for net, key in ((_lib.NET_TE, "te"), (_lib.NET_MD, "md"), (_lib.NET_WM, "wm")):
        for i, (w, a) in enumerate(arch[key]):
            cfg.layer_width[net][i] = w
            cfg.layer_act[net][i] = 1 + _lib.LAYER_ACTS[a]
    got = layers_of(cfg)
    dims = R.net_layer_dims(arch)
    for net, name, key in ((_lib.NET_TE, "_task_encoder", "te"), (_lib.NET_MD, "_motor_decoder", "md"), (_lib.NET_WM, "_world_model", "wm")):
        mine = [l for l in got if l[0] == net]
        assert [(l[2], l[3]) for l in mine] == dims[name]
        assert [l[4] for l in mine] == [_lib.LAYER_ACTS[a] for _, a in arch[key]] + [_lib.ACT_LINEAR]
    n_floats = lib.pvae_arena_floats(C.byref(cfg))
    assert n_floats > 0
    bad = _lib.Config(7, 3, 4, 16, 2, 32, 3, 40, 3, 8, 1, 0, 0, 0, 0)
    bad.layer_act[_lib.NET_MD][1] = 7
    assert lib.pvae_num_layers(C.byref(bad)) < 0 and b"layer_act" in lib.pvae_last_error()
    bad.layer_act[_lib.NET_MD][1] = 0
    bad.layer_width[_lib.NET_WM][0] = -4
    assert lib.pvae_num_layers(C.byref(bad)) < 0 and b"layer_width" in lib.pvae_last_error()
    bad.layer_width[_lib.NET_WM][0] = 65536          # (row strides travel in 16 bits in the fused backward launches)
    assert lib.pvae_num_layers(C.byref(bad)) < 0 and b"wider than 65535" in lib.pvae_last_error()
    bad.layer_width[_lib.NET_WM][0] = 65472
    assert lib.pvae_num_layers(C.byref(bad)) > 0
    # --- the module tree through the trainer (our `*_layers` trainer keys carry the lists)
    data = R.synth_demo(0, 2, 14, 7, 3)
    tr = make_trainer(arch, data, 8, device="cpu")
    sd = tr.model.state_dict()
    assert list(sd.keys()) == list(g["sd_keys"])
    assert [list(v.shape) + [0] * (2 - v.dim()) for v in sd.values()] == g["sd_shapes"].tolist()
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"])
    mods = {"relu": torch.nn.ReLU, "tanh": torch.nn.Tanh, "sigmoid": torch.nn.Sigmoid, "elu": torch.nn.ELU}
    for name, key in (("_task_encoder", "te"), ("_motor_decoder", "md"), ("_world_model", "wm")):
        slims = list(getattr(tr.model, name)._model)
        for slim, (w, a) in zip(slims, arch[key]):
            assert slim._model[0].out_features == w
            assert (len(slim._model) == 1) if a == "linear" else isinstance(slim._model[1], mods[a])
    assert tr.engine.cfg.act_kind == 0 and tr.engine.cfg.layer_width[_lib.NET_MD][2] == 24
    for name, key in (("_task_encoder", "te"), ("_motor_decoder", "md"), ("_world_model", "wm")):
        n = len(arch[key])                        # normc rows: 1.0 hidden, 0.01 output (the lists' init_weight)
        for i in range(n + 1):
            norms = sd["%s._model.%d._model.0.weight" % (name, i)].norm(dim=1)
            assert torch.allclose(norms, torch.full_like(norms, 0.01 if i == n else 1.0), rtol=1e-4)
    # --- init_weight per layer: a normc std of one's own, xavier with a gain; unknown names fail at construction
    from physicsvae_amd.model import PhysicsVAE
    cmc = dict(tr.config["model"]["custom_model_config"])
    lay = [dict(l) for l in cmc["task_encoder_layers"]]
    lay[0]["init_weight"] = {"name": "normc", "std": 0.5}
    lay[1]["init_weight"] = {"name": "xavier_uniform", "gain": 2.0}
    cmc["task_encoder_layers"] = lay
    m = PhysicsVAE(cmc["observation_space"], cmc["action_space"], 6, {"custom_model_config": cmc}, "m")
    w0, w1 = m._task_encoder._model[0]._model[0].weight, m._task_encoder._model[1]._model[0].weight
    assert torch.allclose(w0.norm(dim=1), torch.full((16,), 0.5), atol=1e-5)
    bound = 2.0 * (6.0 / (16 + 24)) ** 0.5
    assert 0.5 * bound < float(w1.detach().abs().max()) <= bound
    for edit in ({"init_weight": {"name": "orthogonal"}}, {"activation": "swish"}, {"type": "bn"}, {"hidden_size": "output"}):
        lay2 = [dict(l) for l in lay]
        lay2[0].update(edit)
        cmc2 = dict(cmc, task_encoder_layers=lay2)
        with pytest.raises(NotImplementedError):
            PhysicsVAE(cmc2["observation_space"], cmc2["action_space"], 6, {"custom_model_config": cmc2}, "m")


def test_cli_flags_for_act_fn_and_weight_decay(tmp_path):
    from physicsvae_amd import train_physics_vae as T
    pkl = str(tmp_path / "d.pkl")
    R.write_demo(pkl, R.synth_demo(0, 2, 14, 7, 3))
    a = T.arg_parser().parse_args(["--data_train", pkl, "--act_fn", "elu", "--weight_decay", "0.01"])
    cfg = T.get_trainer_config(a)
    assert cfg["act_fn"] == "elu" and cfg["weight_decay"] == 0.01
    a = T.arg_parser().parse_args(["--data_train", pkl])
    cfg = T.get_trainer_config(a)
    assert cfg["act_fn"] == "relu" and cfg["weight_decay"] == 0.0              # tpv:262, 253


def test_launcher_with_more_ranks_than_gpus_shares_the_devices(monkeypatch):
    """`parallel.init_from_env` under `torchrun --nproc-per-node 8` on a box with fewer GPUs: the ranks take the
    devices round-robin and rendezvous over gloo (RCCL refuses two ranks on one device); with enough GPUs nothing
    changes.  (Device count faked; no process group is created at WORLD_SIZE = 1.)"""
    from physicsvae_amd import parallel
    for k in ("PVAE_LOCAL_DEVICE", "PVAE_BENCH_SHARED_GPU", "PVAE_DIST_BACKEND", "PVAE_DP_ALWAYS_REDUCE"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("LOCAL_RANK", "5")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert parallel.init_from_env() == (0, 1, 1)                     # 5 % 2
    assert os.environ["PVAE_LOCAL_DEVICE"] == "1" and os.environ["PVAE_BENCH_SHARED_GPU"] == "1"
    monkeypatch.delenv("PVAE_LOCAL_DEVICE")
    monkeypatch.delenv("PVAE_BENCH_SHARED_GPU")
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert parallel.init_from_env() == (0, 1, 5)
    assert "PVAE_LOCAL_DEVICE" not in os.environ and "PVAE_BENCH_SHARED_GPU" not in os.environ


def test_build_command_keeps_the_kernarg_preload_switch():
    """The contraction kernels take what a workgroup needs for its first tile load as leading scalar parameters
    (pvae_gemm.h PVAE_GA_PARAMS / PVAE_GA2_PARAMS) so that gfx950 preloads them into SGPRs at wave launch; that only
    happens when hipcc is given -amdgpu-kernarg-preload-count (0.26 us per dependent launch otherwise)."""
    import inspect
    from physicsvae_amd import build as B
    assert "-amdgpu-kernarg-preload-count=16" in B.FLAGS and B.FLAGS[B.FLAGS.index("-amdgpu-kernarg-preload-count=16") - 1] == "-mllvm"
    assert "FLAGS" in inspect.getsource(B.build)                 # every translation unit is compiled with them
    hdr = open(os.path.join(ROOT, "physicsvae_amd", "csrc", "pvae_gemm.h")).read()
    for kernel in ("gemm_splitk_ws_kernel(PVAE_GA_PARAMS(a_)", "gemm_splitk_reg16_kernel(PVAE_GA_PARAMS(a_)",
                   "bwd_pair_kernel(PVAE_GA2_PARAMS", "wgrad_pair_kernel(PVAE_GA2_PARAMS"):
        assert kernel in hdr, kernel


def test_input_subsets_layout_matches_the_reference_capture(golden):
    """task_encoder_inputs / motor_decoder_inputs (rmt:470, 485): the reference builds narrower first layers; same keys
    and shapes here -- as column windows of the full-width weight blocks, whose other columns are structural zeros that
    neither the initialisation nor a state-dict load touches."""
    g = golden("subsets_tiny")
    base = arch_from_meta(g["meta"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, base["Db"], base["Da"], kind="iid")
    Db, Z = base["Db"], base["Z"]
    for ci, combo in enumerate(g["combos"]):
        te_in, md_in = (tuple(part.split("+")) for part in str(combo).split("/"))
        arch = R.with_inputs(base, te_in, md_in)
        tr = make_trainer(arch, data, batch, device="cpu")
        sd = tr.model.state_dict()
        assert list(sd.keys()) == [str(k) for k in g["c%d_sd_keys" % ci]]
        for (k, v), shp in zip(sd.items(), g["c%d_sd_shapes" % ci]):
            assert list(v.shape) == [int(x) for x in shp[: v.dim()]], k
        assert [(k, tuple(v.shape)) for k, v in sd.items()] == R.state_dict_spec(arch)
        eng = tr.engine
        want = {_lib.NET_TE: {("body", "task"): (0, 2 * Db), ("body",): (0, Db), ("task",): (Db, Db)}[te_in],
                _lib.NET_MD: {("body", "task"): (0, Db + Z), ("body",): (0, Db), ("task",): (Db, Z)}[md_in]}
        tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
        for info in eng.layers:
            if info["index"] == 0 and info["net"] in want:
                assert (info["col0"], info["n_in"]) == want[info["net"]]
                assert info["ld"] == 64 * ((sum({_lib.NET_TE: (Db, Db), _lib.NET_MD: (Db, Z)}[info["net"]]) + 63) // 64)
                blk = eng.params[info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]].view(info["n_out_pad"], info["ld"])
                inside = blk[: info["n_out"], info["col0"]: info["col0"] + info["n_in"]]
                assert float(inside.abs().min()) > 0.0
                assert int((blk != 0).sum()) == inside.numel()                    # everything outside the window is zero
            else:
                assert info["col0"] == 0
    with pytest.raises(NotImplementedError):
        make_trainer(dict(base, te_inputs=("task", "body")), data, batch, device="cpu")


def test_dp_buckets_follow_the_backward_plan():
    """torch_models.dp_buckets: slices finished stage by stage (last layer first, adjacent below one another) merge into one
    bucket per stack, close at a stack boundary, at a gap, and at the size limit; each bucket names the stage after which
    it is final -- what every rank, also one with an empty shard, turns into its sequence of collectives."""
    plan = [(2, 0, 0),                       # an input-gradient stage: finishes nothing
            (1, 900, 100), (1, 700, 200), (1, 600, 100),          # decoder: three layers, last to first
            (4, 1000, 50),                   # helper (another stack)
            (0, 300, 300), (0, 100, 200), (0, 0, 100)]            # encoder
    assert TM.dp_buckets(plan) == [(1, 600, 1000, 3), (4, 1000, 1050, 4), (0, 0, 600, 7)]
    assert TM.dp_buckets(plan, limit=250) == [(1, 700, 1000, 2), (1, 600, 700, 3), (4, 1000, 1050, 4), (0, 300, 600, 5),
                                              (0, 0, 300, 7)]
    assert TM.dp_buckets([(0, 500, 100), (0, 100, 100)]) == [(0, 500, 600, 0), (0, 100, 200, 1)]      # a gap closes too
    assert TM.dp_buckets([]) == [] and TM.dp_buckets([(2, 0, 0)]) == []
