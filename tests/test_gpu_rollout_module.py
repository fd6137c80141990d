"""The rollout side of the boundary (rmt:742-771 and the PhysicsVAE module surface RLlib drives): the <= 4-row
rollout path against the staged forward, host observation -> host action, the module forward + value branch (also on
stacks given layer by layer), HIP-graph replays between prefetched training steps, device moves of the module, log_std."""
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


@pytest.mark.parametrize("rows", [1, 2, 4])
def test_rollout_from_host_observation_to_host_action(golden, rows):
    """`HipEngine.infer_host` (rmt:742-771 for a control loop whose environment lives on the CPU,
    envs/rllib_env_imitation.py:215-266): the observation is read from pinned host memory by the first encoder launch,
    the action is written into a pinned buffer by the decoder's last launch, and the host polls that buffer (NaN
    pre-fill) instead of synchronising.  Same kernels as `infer`: equal bit for bit, with and without sampler noise
    (Philox draws keyed by seed / offset), with the log-std half appended; the staged minibatch is left alone."""
    g = golden("single_default")
    arch = arch_from_meta(g)
    data = R.synth_demo(0, 2, 40, arch["Db"], arch["Da"], kind="dynamics")
    tr = make_trainer(arch, data, 32, device=DEV)
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    eng, Da = tr.engine, arch["Da"]
    obs = torch.randn(rows, 2 * arch["Db"], generator=torch.Generator().manual_seed(rows))
    for noise in (False, True):
        for call in range(3):                                   # the pinned buffers are re-used call after call
            want = eng.infer(obs.to(DEV), noise=noise, seed=5, offset=call, want_s2=False)[0].cpu()
            got = eng.infer_host(obs, noise=noise, seed=5, offset=call)
            assert got.device.type == "cpu" and got.shape == (rows, Da) and torch.equal(got, want)
            got_np = eng.infer_host(obs.numpy(), noise=noise, seed=5, offset=call)       # numpy observations too
            assert torch.equal(got_np, want)
    ls = torch.full((Da,), -1.5, device=DEV)
    logits = eng.infer_host(obs, noise=False, log_std=ls)
    assert logits.shape == (rows, 2 * Da) and torch.equal(logits[:, :Da], eng.infer(obs.to(DEV), noise=False, want_s2=False)[0].cpu())
    assert torch.equal(logits[:, Da:], ls.cpu().expand(rows, -1))
    with pytest.raises(ValueError):
        eng.infer_host(torch.randn(5, 2 * arch["Db"]))
    # a staged training minibatch survives the call
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    c = R.phase_coeffs(True)
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=32)
    eng.gather(0, 32)
    l0 = eng.forward_backward(_lib.PHASE_WORLD, 32, sp, backward=False).clone()
    eng.gather(0, 32)
    eng.infer_host(obs, noise=False)
    assert torch.equal(eng.forward_backward(_lib.PHASE_WORLD, 32, sp, backward=False), l0)


@pytest.mark.parametrize("rows", [1, 4, 33])
def test_rollout_and_value_branch_with_per_layer_stacks(golden, rows):
    """PhysicsVAE.forward (rmt:742-771) on stacks given layer by layer -- own width and activation per hidden layer,
    FC's general layer list (rmt:234-270) --, the value branch included (rmt:846-853: `pvae_mlp_forward` with one
    activation per hidden layer when sampling, the torch module under autograd)."""
    from physicsvae_amd.model import PhysicsVAE
    g = golden("single_mixed_c1")
    arch = dict(arch_from_meta(g), vb=[(48, "tanh"), (32, "linear"), (40, "elu")])
    data = R.synth_demo(0, 2, 40, arch["Db"], arch["Da"], kind="dynamics")
    tr = make_trainer(arch, data, 64, device=DEV)
    cmc = dict(tr.config["model"]["custom_model_config"], value_fn_layers=R.fc_layer_list(arch["vb"]))
    m = PhysicsVAE(cmc["observation_space"], cmc["action_space"], 2 * arch["Da"], {"custom_model_config": cmc}, "m")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    m.load_state_dict(sd)
    ref = R.RefModel(arch)
    ref.load_state_dict(sd)
    obs = torch.randn(rows, 2 * arch["Db"], generator=torch.Generator().manual_seed(rows))
    e = R.eps_stream(2, arch["Z"])(0, (rows, arch["Z"]))
    ref.eps_source = lambda shape: e
    want = ref(obs).detach()
    with torch.no_grad():
        logits, _ = m.forward({"obs_flat": obs.to(DEV)}, [], None, eps=e)
        value = m.value_function().cpu()
    assert max_err_scaled(logits.cpu(), want) < 2e-5
    assert max_err_scaled(m._cur_future_state.cpu(), ref.cur_future_state.detach()) < 2e-5
    assert max_err_scaled(m.task_encoder_variable().cpu(), ref.cur_z.detach()) < 2e-5
    assert max_err_scaled(value, ref.cur_value.detach()) < 1e-4
    v_torch, _ = m.forward_value_branch(obs.to(DEV))                 # autograd on: the plain module
    assert v_torch.requires_grad and max_err_scaled(v_torch.detach().cpu().reshape(-1), ref.cur_value.detach().reshape(-1)) < 1e-4
    a_hat, s2, z = m.engine.infer(obs.to(DEV), eps=e.to(DEV))
    assert max_err_scaled(a_hat.cpu(), want[:, : arch["Da"]]) < 2e-5 and max_err_scaled(s2.cpu(), ref.cur_future_state.detach()) < 2e-5
