"""`motor_decoder_helper_enable` (rmt:490-498, 670-680, 833-835): a second stack on the motor decoder's input whose tanh
output, scaled by `motor_decoder_helper_range`, is added to the action half of the logits -- the residual policy that RL
fine-tuning puts on top of a frozen decoder.  train_physics_vae.py never builds it, so it is part of the ROLLOUT path
(PhysicsVAE.forward / forward_decoder); checked against captures of the reference's own model (tests/golden/helper_*.npz,
oracle/gen_golden.py case_helper) and against the oracle's restatement."""
import os

import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import train_physics_vae as T

TINY = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
DFLT = R.make_arch(197, 45)


def _helper_model(arch, device, tmp_path, enable=True, batch=8):
    """The module as a rollout user builds it: the trainer's model config with the helper switched on."""
    data = R.synth_demo(0, 2, 14, arch["Db"], arch["Da"])
    pkl = os.path.join(str(tmp_path), "demo.pkl")
    R.write_demo(pkl, data)
    argv = ["--data_train", pkl, "--batch_size", str(batch), "--latent_dim", str(arch["Z"])]
    for flag, key in (("TE", "te"), ("MD", "md"), ("world_model", "wm")):
        argv += ["--%s_width" % flag, str(arch[key][0]), "--%s_depth" % flag, str(arch[key][1])]
    T.args = T.arg_parser().parse_args(argv)
    cfg = T.get_trainer_config(T.args)
    for k, v in list(cfg.items()):
        if isinstance(v, dict) and "grid_search" in v:
            cfg[k] = v["grid_search"][0]
    T.update_model_config(cfg)
    cfg["model"]["custom_model_config"].update(motor_decoder_helper_enable=enable, device=device)
    return cfg, T.create_model(cfg)


def _weights(arch):
    h = R.with_helper(arch)
    sd = R.perturb_biases(R.init_state_dict(h, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(h["mh"])
    sd[k_out] = sd[k_out] * 60.0                      # (as the capture: an output layer of norm 0.01 would hide the term)
    return h, sd


@pytest.mark.parametrize("name,arch", [("helper_tiny", TINY), ("helper_default", DFLT)])
def test_helper_state_dict_layout_and_files_match_the_reference(golden, name, arch, tmp_path):
    g = golden(name)
    cfg, m = _helper_model(arch, "cpu", tmp_path)
    sd = m.state_dict()
    assert list(sd.keys()) == list(g["sd_keys"])           # the helper registers between motor decoder and world model
    assert [list(v.shape) + [0] * (2 - v.dim()) for v in sd.values()] == g["sd_shapes"].tolist()
    assert m._motor_decoder_helper_range == float(g["helper_range"]) == 0.5
    h, ref = _weights(arch)
    assert [k for k, _ in R.state_dict_spec(h)] == list(sd.keys())
    m.load_state_dict(ref)
    for k, v in m.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k
    # its own file (rmt:891-893, 907-910), the switch (rmt:942-946), normc init of a fresh module
    f = os.path.join(str(tmp_path), "helper.pt")
    m.save_weights_motor_decoder_helper(f)
    assert list(torch.load(f).keys()) == list(g["helper_file_keys"])
    cfg2, m2 = _helper_model(arch, "cpu", tmp_path)
    last = len(h["mh"])
    for i, mod in enumerate(m2._motor_decoder_helper._model):
        np.testing.assert_allclose(mod._model[0].weight.detach().norm(dim=1).numpy(), 0.01 if i == last else 1.0, rtol=1e-4)
    assert isinstance(m2._motor_decoder_helper._model[last]._model[-1], torch.nn.Tanh)
    m2.load_weights_motor_decoder_helper(f)
    for (k, a), (_, b) in zip(m2._motor_decoder_helper.state_dict().items(), m._motor_decoder_helper.state_dict().items()):
        assert torch.equal(a, b), k
    m2.set_learnable_motor_decoder_helper(False)
    assert not any(p.requires_grad for p in m2._motor_decoder_helper.parameters())
    # without the switch nothing changes: no module, no keys
    _, m0 = _helper_model(arch, "cpu", tmp_path, enable=False)
    assert m0._motor_decoder_helper is None and not any("helper" in k for k in m0.state_dict())
    m0.save_weights_motor_decoder_helper(os.path.join(str(tmp_path), "none.pt"))     # (a no-op upstream too)
    assert not os.path.exists(os.path.join(str(tmp_path), "none.pt"))


def test_helper_config_is_checked_like_upstream_and_the_trainer_takes_a_helper_model(tmp_path):
    cfg, _ = _helper_model(TINY, "cpu", tmp_path)
    cmc = cfg["model"]["custom_model_config"]
    bad = [dict(l) for l in cmc["motor_decoder_helper_layers"]]
    bad[-1]["activation"] = "linear"
    cmc2 = dict(cmc, motor_decoder_helper_layers=bad)
    with pytest.raises(AssertionError):                                        # rmt:672
        T.create_model(dict(cfg, model=dict(cfg["model"], custom_model_config=cmc2)))
    with pytest.raises(AssertionError):                                        # rmt:673
        T.create_model(dict(cfg, model=dict(cfg["model"], custom_model_config=dict(cmc, motor_decoder_helper_range=0.0))))
    # the supervised trainer trains the helper inside a_hat, as upstream does (tests/test_gpu_helper_training.py); the
    # helper is a fifth stack of the arena, between the motor decoder and the learned prior / world model
    data = R.synth_demo(0, 2, 14, 7, 3)
    from util import make_trainer
    from physicsvae_amd import _lib
    tr = make_trainer(TINY, data, 8, device="cpu")
    assert tr.model._motor_decoder_helper is None and tr.engine.segments[_lib.NET_MH][1] == 0
    full = dict(tr.config)
    full["model"] = dict(full["model"], custom_model_config=dict(full["model"]["custom_model_config"], motor_decoder_helper_enable=True))
    th = T.TrainModel(full)
    eng = th.engine
    assert th.model._motor_decoder_helper is not None and eng.segments[_lib.NET_MH][1] > 0
    assert eng.segments[_lib.NET_MD][0] + eng.segments[_lib.NET_MD][1] == eng.segments[_lib.NET_MH][0]
    assert eng.segments[_lib.NET_MH][0] + eng.segments[_lib.NET_MH][1] == eng.segments[_lib.NET_WM][0]
    assert [(k, tuple(v.shape)) for k, v in th.model.state_dict().items()] == R.state_dict_spec(R.with_helper(TINY))
    assert th.phase() == (_lib.PHASE_WORLD, [_lib.NET_WM])                  # (the helper has no gradient there, lookahead 1)
    th.model.set_learnable_task_encoder(True); th.model.set_learnable_motor_decoder(True); th.model.set_learnable_world_model(False)
    assert th.phase() == (_lib.PHASE_JOINT, [_lib.NET_TE, _lib.NET_MD, _lib.NET_MH])
    assert th.optimizer.next_counts([_lib.NET_TE, _lib.NET_MD, _lib.NET_MH]) == [1, 1, 1, 1, 1]
    th.model.set_learnable_motor_decoder_helper(False)
    assert th.phase() == (_lib.PHASE_JOINT, [_lib.NET_TE, _lib.NET_MD])
    assert th.optimizer.next_counts([_lib.NET_TE, _lib.NET_MD])[4] == 0     # adam_t[PVAE_NET_MH] = 0: frozen for that step
    # lookahead > 1 (tpv:367-428): the WORLD phase reaches the helper too -- the state the world model continues from is its
    # own prediction under the helped action -- so it joins the world phase's trainable stacks there
    full["lookahead"] = 2
    t2 = T.TrainModel(full)
    assert t2.engine.lookahead == 2 and t2.phase() == (_lib.PHASE_WORLD, [_lib.NET_WM, _lib.NET_MH])
    assert t2.optimizer.next_counts([_lib.NET_WM, _lib.NET_MH]) == [1, 1, 1, 1, 1]


@pytest.mark.gpu
@pytest.mark.parametrize("name,arch", [("helper_tiny", TINY), ("helper_default", DFLT)])
def test_helper_forward_matches_the_reference_and_the_oracle(golden, name, arch, tmp_path):
    """PhysicsVAE.forward and forward_decoder of a helper model against the reference's own outputs at the same weights,
    observations and draws: logits (action = decoder + range * tanh-stack, then log_std), z, the world model's prediction
    -- which must see the HELPED action (rmt:758) -- and the value; 2e-5 of max.  Then the autograd route: with gradients
    enabled the helper's term comes from the torch module, so that its parameters receive the gradient a policy-gradient
    learner needs, and equals the kernel route's."""
    g = golden(name)
    _, m = _helper_model(arch, "cuda", tmp_path)
    h, sd = _weights(arch)
    m.load_state_dict(sd)
    m.eval()
    obs, eps = torch.from_numpy(g["obs"]), torch.from_numpy(g["eps"])
    Da = arch["Da"]

    def close(a, b, tol=2e-5):
        b = torch.as_tensor(b)
        return float((a.detach().cpu() - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    for noise, tag in ((False, "mean"), (True, "noise")):
        m.latent_prior_noise = noise
        with torch.no_grad():
            logits, _ = m.forward({"obs_flat": obs.cuda()}, [], None, eps=eps.cuda())
        assert logits.shape == (obs.shape[0], 2 * Da)
        assert close(logits, g[tag + "_logits"]), (tag, float((logits.cpu() - torch.from_numpy(g[tag + "_logits"])).abs().max()))
        assert close(m.task_encoder_variable(), g[tag + "_z"])
        assert close(m._cur_future_state, g[tag + "_future_state"])
        assert close(m.value_function(), g[tag + "_value"])
    # the helper's term is really there: the same module without it answers differently
    ref = R.RefModel(h)
    ref.load_state_dict(sd)
    with torch.no_grad():
        zin = torch.cat([obs[:, : arch["Db"]], eps], dim=-1)
        term = h["mh_range"] * ref._motor_decoder_helper(zin)
        plain = ref._motor_decoder(zin)
    assert float(term.abs().max()) > 0.5 * float(plain.abs().max())
    with torch.no_grad():
        dec, _ = m.forward_decoder(obs[:, : arch["Db"]].cuda(), eps.cuda())
    assert close(dec, g["decoder_logits"]) and close(dec[:, :Da], plain + term)
    # autograd: the gradient of a function of the action reaches the helper's parameters (and only flows through it)
    m.set_learnable_motor_decoder_helper(True)
    for p in m._motor_decoder_helper.parameters():
        p.grad = None
    with torch.enable_grad():
        dec2, _ = m.forward_decoder(obs[:, : arch["Db"]].cuda(), eps.cuda())
        assert torch.allclose(dec2.detach(), dec, rtol=0, atol=2e-6)
        (dec2[:, :Da] ** 2).sum().backward()
    ref_h = R.RefModel(h)
    ref_h.load_state_dict(sd)
    a = (plain + h["mh_range"] * ref_h._motor_decoder_helper(zin))
    ((plain + h["mh_range"] * ref_h._motor_decoder_helper(zin)) ** 2).sum().backward()
    for (k, p), (_, q) in zip(m._motor_decoder_helper.named_parameters(), ref_h._motor_decoder_helper.named_parameters()):
        assert p.grad is not None and float((p.grad.cpu() - q.grad).abs().max()) <= 1e-4 * max(1e-6, float(q.grad.abs().max())), k
    # the resident server serves helper models too (tests/test_gpu_rollout_server.py): the served action is the launched one
    m.latent_prior_noise = False
    with torch.no_grad():
        want, _ = m.forward({"obs_flat": obs[:1].cuda()}, [], None)
    m.start_rollout_server(idle_ms=500.0, lifetime_s=30.0)
    try:
        with torch.no_grad():
            got, _ = m.forward({"obs_flat": obs[:1].cpu()}, [], None)
    finally:
        m.stop_rollout_server()
    assert torch.equal(got.cpu(), want.cpu())
