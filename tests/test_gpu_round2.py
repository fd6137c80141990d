"""Round-2 GPU tests: bench.py launched plainly (N = 1 and N = 2 self-launch), HIP-graph rollout
replays interleaved with prefetched training steps, identity device moves of the module, and the
gather kernel on a demonstration set whose byte offsets cross 2^31."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from util import arch_from_meta, make_trainer

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
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5
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
    assert ref["steps"] == 3 and ref["global_batch"] == 512 and ref["max_rel_loss_diff"] < 2e-4 and ref["update_rel_l2_diff"] < 2e-2
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
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "weak"
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
    assert cb["value"] >= cb["value_1thread"] > 0 and cb["threads_best"] in [int(k) for k in cb["sweep"]]
    assert len(d["timing"]["region_values"]) == 3


def test_graph_replays_between_prefetched_train_steps_do_not_touch_the_training_batch(golden):
    """GraphedInfer writes the staging panels that were current when it was captured; a prefetched
    train step swaps current and alternate panels.  Replays interleaved with prefetched steps (odd
    and even counts in between) must leave training bit-identical to an engine that never
    prefetches and never replays, and a replay must not leave a stale 'staged' minibatch behind."""
    g = golden("train_tiny")
    arch = arch_from_meta(g["meta"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    obs = torch.randn(1, 2 * arch["Db"], generator=torch.Generator().manual_seed(5)).to(DEV)
    res = []
    for replay in (False, True):
        tr = make_trainer(arch, data, batch, m_world=1, device=DEV, extra={"prefetch_gather": replay})
        tr.model.load_state_dict(sd)
        eng = tr.engine
        eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
        gi = eng.graphed_infer(1, want_s2=True, noise=False) if replay else None
        tr.model.set_learnable_task_encoder(True); tr.model.set_learnable_motor_decoder(True)
        tr.model.set_learnable_world_model(False); tr.read_loss_fn_coeff(world=False)
        phase, nets = tr.phase()
        spans = list(tr.train_loader.spans())
        out = torch.zeros(len(spans), 5, device=DEV)
        for i, (first, rows) in enumerate(spans):
            sp = tr.step_params(nets, rows, True)
            e = R.eps_stream(7, arch["Z"])(i, (rows, arch["Z"]))
            nxt = spans[(i + 1) % len(spans)]
            eng.train_step(phase, first, rows, sp, eps=e, loss_out=out[i], next_span=nxt if replay else None)
            if replay and i % 3 != 2:               # 1, then 2 prefetched steps between replays: both parities
                a_hat = gi(obs)[0].clone()
                assert torch.equal(a_hat, eng.infer(obs, noise=False, want_s2=True)[0])
        res.append((out.clone(), eng.params.clone(), eng.exp_avg.clone()))
        if replay:
            # gather -> replay -> forward_backward must fail loudly, not run on overwritten panels
            eng.gather(0, batch)
            gi(obs)
            with pytest.raises(RuntimeError, match="staged"):
                eng.forward_backward(phase, batch, tr.step_params(nets, batch, False), backward=False)
    (la, pa, ma), (lb, pb, mb) = res
    assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(ma, mb)


def test_identity_device_moves_of_the_module(golden):
    """`model.to(device)` with the device it already lives on (what the reference's trainer and
    RLlib call), `.cuda()` and `.float()` are no-ops; a real move is refused."""
    g = golden("single_tiny")
    arch = arch_from_meta(g["meta"])
    data = R.synth_demo(0, 2, 20, arch["Db"], arch["Da"], kind="iid")
    tr = make_trainer(arch, data, 8, device=DEV)          # "cuda" without an index
    m = tr.model
    ptr = m.engine.params.data_ptr()
    assert m.to("cuda") is m and m.cuda() is m and m.float() is m
    assert m.to(torch.device("cuda", torch.cuda.current_device())) is m and m.to(m.engine.device) is m
    assert m.engine.params.data_ptr() == ptr
    assert next(m._world_model.parameters()).data_ptr() >= ptr         # still views of the arena
    with pytest.raises(RuntimeError):
        m.to("cpu")
    with pytest.raises(RuntimeError):
        m.double()


def test_gather_windows_across_the_2GiB_byte_boundary():
    """A demonstration set of 1.5e6 state rows x 400 floats = 2.4 GB: byte offsets of the rows a
    window reads cross 2^31 (row 1 342 177).  The panels the gather kernel fills for windows at the
    start, straddling the boundary and at the very end equal the oracle's definition of a window
    (x = [s_t | s_{t+1}], y = a_t; tpv:133-156) bit for bit."""
    from physicsvae_amd.engine import Arch, HipEngine
    Db, Da, B = 400, 90, 512
    rows_total = 1_500_000
    eng = HipEngine(Arch(Db, Da, 32, (64, 1), (64, 1), (64, 1)), B, device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    states = torch.randn(rows_total, Db, generator=gen, device=DEV)
    actions = torch.randn(rows_total, Da, generator=gen, device=DEV)
    assert states.numel() * 4 > 2 ** 31
    # windows: every row but the last of each 1000-row "episode" (episode boundaries are skipped)
    idx = torch.arange(rows_total, device=DEV)
    window_row = idx[(idx % 1000) != 999].to(torch.int32)
    eng.bind_dataset(states, actions, window_row)
    n = window_row.numel()
    boundary_row = 2 ** 31 // (Db * 4)                   # first row whose bytes start beyond 2^31
    first_over = int(torch.searchsorted(window_row, torch.tensor(boundary_row, device=DEV, dtype=torch.int32)))
    for first, rows in ((0, B), (first_over - B // 2, B), (n - B, B), (n - 37, 37)):
        eng.gather(first, rows)
        torch.cuda.synchronize()
        r = window_row[first: first + rows].long()
        te_in = eng.panel("in", _lib.NET_TE)[:rows]
        wm_in = eng.panel("in", _lib.NET_WM)[:rows]
        md_in = eng.panel("in", _lib.NET_MD)[:rows]
        assert torch.equal(te_in[:, :Db], states[r]) and torch.equal(te_in[:, Db: 2 * Db], states[r + 1])
        assert torch.equal(wm_in[:, :Db], states[r]) and torch.equal(wm_in[:, Db: Db + Da], actions[r])
        assert torch.equal(md_in[:, :Db], states[r])
        assert torch.equal(eng.panel("s2")[:rows, :Db], states[r + 1])
        assert torch.equal(eng.panel("act_t")[:rows, :Da], actions[r])
        assert float(te_in[:, 2 * Db:].abs().max()) == 0.0            # pad columns are zeros
        if rows < B:
            assert float(eng.panel("in", _lib.NET_TE)[rows: (rows + 31) // 32 * 32].abs().max()) == 0.0


@pytest.mark.parametrize("stop_after", [3, 1, 2])
def test_trainer_state_resume_is_bit_exact(golden, tmp_path, stop_after):
    """`save_trainer_state` / `resume_trainer_state` (ours; upstream resumes weights only, tm:215-216):
    a run interrupted after `stop_after` epochs (world -> joint switch at 2: after it, before it, exactly at
    it) and resumed from its checkpoint directory continues bit for bit like the uninterrupted run: weights,
    Adam moments, step counts, StepLR position, eps stream position, and the PHASE."""
    g = golden("train_tiny")
    arch = arch_from_meta(g["meta"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    kw = dict(m_world=2, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, arch["Z"]))
    full = make_trainer(arch, data, batch, extra={"save_trainer_state": True}, **kw)
    full.model.load_state_dict(sd)
    for _ in range(stop_after):
        full.train()
    ck = full.save_checkpoint(str(tmp_path))
    assert os.path.exists(tmp_path / "trainer_state.pt")
    learnable_at_save = sorted(full.model.learnable_nets())
    want = [full.train()["mean_train_loss"] for _ in range(3)]
    res = make_trainer(arch, data, batch, extra={"resume_trainer_state": True}, **kw)
    res.restore(ck)
    assert res.iter == stop_after
    # the restore itself re-entered the phase the state was saved in (joint when saved after the switch)
    assert sorted(res.model.learnable_nets()) == learnable_at_save
    assert (res.a_rec_coeff > 0) == (stop_after > 2)
    got = [res.train()["mean_train_loss"] for _ in range(3)]
    assert got == want
    assert torch.equal(res.engine.params, full.engine.params)
    assert torch.equal(res.engine.exp_avg, full.engine.exp_avg) and torch.equal(res.engine.exp_avg_sq, full.engine.exp_avg_sq)
    assert res.optimizer.net_steps == full.optimizer.net_steps and res.optimizer.lr == full.optimizer.lr


def test_state_independent_log_std_is_a_parameter_the_loss_never_touches(golden):
    """log_std_type "state_independent" (rmt:178-181): a learnable tensor in the state dict that the
    supervised loss cannot reach (tpv:356-359 slices the action half), so training leaves it alone."""
    g = golden("single_tiny")
    arch = arch_from_meta(g["meta"])
    data = R.synth_demo(0, 2, 30, arch["Db"], arch["Da"], kind="dynamics")
    from physicsvae_amd import train_physics_vae as T
    tr = make_trainer(arch, data, 8, m_world=1, device=DEV)
    cfg = dict(tr.config)
    cfg["model"]["custom_model_config"]["log_std_type"] = "state_independent"
    tr = T.TrainModel(cfg)
    key = "_motor_decoder._model.%d.log_std" % (arch["md"][1] + 1)
    sd = tr.model.state_dict()
    assert key in sd and sd[key].shape == (arch["Da"],)
    before = sd[key].clone()
    tr.train(); tr.train()                     # world epoch, joint epoch
    assert torch.equal(tr.model.state_dict()[key], before)
    with pytest.raises(AssertionError):
        tr.model.set_exploration_std(0.05)     # rmt:191-193
    logits, _ = tr.model({"obs": torch.zeros(2, 2 * arch["Db"], device=DEV)})
    assert torch.allclose(logits[:, arch["Da"]:].cpu(), before.cpu().expand(2, -1))


@pytest.mark.parametrize("rows", [1, 2, 3, 4])
def test_fused_rollout_path_equals_the_staged_forward(golden, rows):
    """pvae_infer at <= 4 rows (input assembly, sampler and output copies inside the layer launches: 7
    launches instead of 9, 10 instead of 14 with the world model) against the stage-by-stage module
    forward (forward_encoder / forward_decoder / forward_world over padded panels), with supplied draws,
    Philox draws and noise off; the staged training minibatch survives a rollout call."""
    g = golden("single_default")
    arch = arch_from_meta(g["meta"])
    data = R.synth_demo(0, 2, 40, arch["Db"], arch["Da"], kind="dynamics")
    tr = make_trainer(arch, data, 32, device=DEV)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    m, eng = tr.model, tr.engine
    obs = torch.randn(rows, 2 * arch["Db"], generator=torch.Generator().manual_seed(rows)).to(DEV)
    eps = torch.randn(rows, arch["Z"], generator=torch.Generator().manual_seed(9)).to(DEV)
    for noise, e in ((True, eps), (False, None)):
        m.latent_prior_noise = noise
        want, _ = m._forward_staged(obs, [], None, eps=e)
        want_s2, want_z = m._cur_future_state.clone(), m.task_encoder_variable().clone()
        a_hat, s2, z = eng.infer(obs, eps=e, noise=noise, want_s2=True)
        assert float((a_hat - want[:, : arch["Da"]]).abs().max()) < 1e-6
        assert float((s2 - want_s2).abs().max()) < 1e-6 and float((z - want_z).abs().max()) < 1e-6
        a2, none, z2 = eng.infer(obs, eps=e, noise=noise, want_s2=False)
        assert none is None and torch.equal(a2, a_hat) and torch.equal(z2, z)
        logits, _ = m.forward({"obs_flat": obs}, [], None, eps=e)            # the module's forward is the same call
        assert torch.equal(logits[:, : arch["Da"]], a_hat)
        assert float((m._cur_task_encoder_mu - eng.read("mu", rows)).abs().max()) == 0.0
    # Philox draws: keyed by (seed, offset), reproducible, standard-normal sized
    m.latent_prior_noise = True
    z1 = eng.infer(obs, noise=True, seed=5, offset=11)[2].clone()
    z2 = eng.infer(obs, noise=True, seed=5, offset=11)[2].clone()
    z3 = eng.infer(obs, noise=True, seed=5, offset=12)[2].clone()
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    # a rollout call between gather and forward_backward leaves the staged minibatch alone
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    c = R.phase_coeffs(True)
    from physicsvae_amd.engine import make_step_params
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=32)
    eng.gather(0, 32)
    l0 = eng.forward_backward(_lib.PHASE_WORLD, 32, sp, backward=False).clone()
    eng.gather(0, 32)
    eng.infer(obs, noise=False)
    l1 = eng.forward_backward(_lib.PHASE_WORLD, 32, sp, backward=False).clone()
    assert torch.equal(l0, l1)
    # the same in the JOINT phase with the backward pass (the encoder / decoder forwards pick their kernels by
    # the staged row count: a rollout call must not shrink it to its own 1-4 rows)
    cj = R.phase_coeffs(False)
    spj = make_step_params(lr=5e-4, a_rec=cj["a_rec_coeff"], kl=cj["vae_kl_coeff"], s_rec=cj["s_rec_coeff"],
                           cyc=cj["vae_cycle_coeff"], global_rows=32)
    eps = torch.randn(32, arch["Z"], device=DEV)
    eng.gather(0, 32)
    j0 = eng.forward_backward(_lib.PHASE_JOINT, 32, spj, eps=eps, backward=True).clone()
    g0 = eng.grads.clone()
    eng.grads.zero_()
    eng.gather(0, 32)
    eng.infer(obs, noise=False)
    j1 = eng.forward_backward(_lib.PHASE_JOINT, 32, spj, eps=eps, backward=True).clone()
    assert torch.equal(j0, j1)
    assert torch.equal(g0, eng.grads)
    assert float(g0.abs().sum()) > 0


def test_gather_of_cond_rel_windows_is_bit_exact(golden, tmp_path):
    """cond = "rel" datasets on the device: the gather kernel reads the second half of x and the target
    s2 from the row-aligned `next_states` array; the panels equal the dataset's windows bit for bit and
    a training epoch over them runs (the world model then learns state DIFFERENCES)."""
    from physicsvae_amd import train_physics_vae as T
    g = golden("ingest_rel_tiny")
    arch = arch_from_meta(g["meta"])
    Db, Da = arch["Db"], arch["Da"]
    data = R.synth_demo(3, 3, 12, Db, Da, kind="iid", quantum=0.0)
    pkl = str(tmp_path / "a.pkl")
    R.write_demo(pkl, data)
    tr = make_trainer(arch, data, 8, m_world=1, device=DEV)
    ds = T.load_dataset_for_PhysicsVAE([pkl], cond="rel")
    tr.train_loader.dataset = ds
    eng = tr.engine
    eng.bind_dataset(*ds.device_arrays(eng.device))
    for first, rows in ((0, 8), (25, 8), (32, 1)):
        eng.gather(first, rows)
        torch.cuda.synchronize()
        xs = torch.stack([ds[i][0][0] for i in range(first, first + rows)]).to(DEV)      # [rows, 2Db]
        ys = torch.stack([ds[i][1][0] for i in range(first, first + rows)]).to(DEV)
        assert torch.equal(eng.panel("in", _lib.NET_TE)[:rows, : 2 * Db], xs)
        assert torch.equal(eng.panel("s2")[:rows, :Db], xs[:, Db:])
        assert torch.equal(eng.panel("in", _lib.NET_WM)[:rows, :Db], xs[:, :Db])
        assert torch.equal(eng.panel("in", _lib.NET_WM)[:rows, Db: Db + Da], ys)
    r1, r2 = tr.train(), tr.train()
    assert np.isfinite(r1["mean_train_loss"]) and np.isfinite(r2["mean_train_loss"])
    # against the oracle on the same windows
    X, Y = R.build_windows(data, cond="rel")
    x, y = next(iter(R.make_loader(X, Y, 8)))
    sd = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}
    want = R.loss_and_grads(arch, sd, x, y, None, world=True)
    c = R.phase_coeffs(True)
    from physicsvae_amd.engine import make_step_params
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=8)
    eng.gather(0, 8)
    got = eng.forward_backward(_lib.PHASE_WORLD, 8, sp, backward=False).cpu()
    assert float(got[0]) == pytest.approx(float(want["total"]), rel=1e-5)


@pytest.mark.parametrize("rows", [1, 2, 3, 4, 6, 33])
def test_module_forward_writes_logits_in_one_call_and_value_branch_runs_on_the_gemv_chain(golden, rows):
    """`PhysicsVAE.forward` = one `pvae_infer_logits` call: [a_hat | log_std] written by the launch that produces
    the action (no concatenation, no host -> device copy of a constant log_std per call), `set_exploration_std`
    seen by the next call; `value_function()` under no_grad = `pvae_mlp_forward` on the value branch's own torch
    parameters, equal to the torch module (which still serves autograd callers)."""
    g = golden("single_default")
    arch = arch_from_meta(g["meta"])
    data = R.synth_demo(0, 2, 40, arch["Db"], arch["Da"], kind="dynamics")
    tr = make_trainer(arch, data, 64, device=DEV)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    m, eng, Da = tr.model, tr.engine, arch["Da"]
    m.latent_prior_noise = False
    obs = torch.randn(rows, 2 * arch["Db"], generator=torch.Generator().manual_seed(rows)).to(DEV)
    a_hat, s2, z = eng.infer(obs, noise=False, want_s2=True)
    for std in (0.1, 0.05):
        m.set_exploration_std(std)
        with torch.no_grad():
            logits, _ = m.forward({"obs_flat": obs}, [], None)
            val = m.value_function()
        assert logits.shape == (rows, 2 * Da) and torch.equal(logits[:, :Da], a_hat)
        assert torch.allclose(logits[:, Da:], torch.full((rows, Da), float(np.log(std)), device=DEV), rtol=0, atol=1e-7)
        assert m._st._cur_future_state is None                        # the prediction is lazy by default ...
        assert torch.equal(m._cur_future_state, s2) and torch.equal(m.task_encoder_variable(), z)     # ... and exact when read
        assert torch.equal(m.forward({"obs_flat": obs}, [], None)[0], logits)
        m.rollout_predicts_state = True                               # as upstream: with every forward
        with torch.no_grad():
            m.forward({"obs_flat": obs}, [], None)
        assert torch.equal(m._st._cur_future_state, s2)
        m.rollout_predicts_state = False
        with torch.no_grad():
            m.forward({"obs_flat": obs}, [], None)
        assert m._cur_future_state is None
        m.rollout_predicts_state = "lazy"
        want = m._value_branch(obs).squeeze(1)                       # torch path (autograd on)
        assert val.shape == (rows,) and not val.requires_grad and want.requires_grad
        assert float((val - want.detach()).abs().max()) < 1e-6 * max(1.0, float(want.abs().max()))
    # deferred reads see the observation the ACTION was computed on, also when the caller recycles its buffer
    # right after the call (the forward keeps the library's own copy of the rows, workspace kind 7)
    buf = obs.clone()
    with torch.no_grad():
        m.forward({"obs_flat": buf}, [], None)
        buf.normal_()                                                # the caller's next observation, in place
        assert torch.equal(m._cur_future_state, s2)
        assert torch.equal(m.value_function(), val)
        assert torch.equal(m._cur_task_encoder_mu, eng.read("mu", rows))
    # any stack of dense layers, any row stride: a transposed-storage weight and a strided input
    w1 = torch.randn(19, 40, device=DEV)[:, :37]                     # rows 40 floats apart, 37 used
    b1 = torch.randn(19, device=DEV)
    w2, b2 = torch.randn(5, 19, device=DEV), torch.randn(5, device=DEV)
    x = torch.randn(rows, 50, device=DEV)[:, :37]
    for act, fn in (("relu", torch.relu), ("tanh", torch.tanh), ("elu", torch.nn.functional.elu), ("sigmoid", torch.sigmoid)):
        got = eng.mlp_forward(x, [(w1, b1), (w2, b2)], act=act)
        want = fn(x @ w1.t() + b1) @ w2.t() + b2
        assert float((got - want).abs().max()) < 2e-5


def test_state_independent_log_std_reaches_the_logits(golden):
    g = golden("single_tiny")
    arch = arch_from_meta(g["meta"])
    data = R.synth_demo(0, 2, 14, arch["Db"], arch["Da"], kind="iid")
    tr = make_trainer(arch, data, 8, device=DEV, extra={})
    cfg = dict(tr.config["model"]["custom_model_config"], log_std_type="state_independent", device=DEV)
    from physicsvae_amd.model import PhysicsVAE
    m = PhysicsVAE(cfg["observation_space"], cfg["action_space"], 2 * arch["Da"], {"custom_model_config": cfg}, "m")
    with torch.no_grad():
        m._motor_decoder._model[-1].log_std.copy_(torch.arange(arch["Da"], dtype=torch.float32) * 0.1 - 1.0)
        logits, _ = m.forward({"obs_flat": torch.zeros(1, 2 * arch["Db"], device=DEV)}, [], None)
    assert torch.allclose(logits[0, arch["Da"]:].cpu(), torch.arange(arch["Da"], dtype=torch.float32) * 0.1 - 1.0)
