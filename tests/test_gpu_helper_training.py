"""Supervised training of a model with `motor_decoder_helper_enable` (rmt:490-498, 670-680, 833-835).

The reference's loss sees the helper's term inside a_hat (a_hat = decoder + range * tanh-stack), and nothing in its trainer
ever freezes the helper (tpv:326-329, 347-350 switch encoder, decoder and world model only): the joint phase trains it with
the decoder, the world phase leaves it without a gradient (lookahead 1).  Here the helper is a fifth stack of the arena
(PVAE_NET_MH) between the decoder's and the world model's; the step against the capture of the reference itself
(`helper_train_tiny.npz`, oracle/gen_golden.py case_helper_train) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import arch_from_meta, make_trainer, max_err_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _trainer(arch, data, batch, **kw):
    """make_trainer with the helper switched on in custom_model_config (a dict edit, as upstream)."""
    from physicsvae_amd import train_physics_vae as T
    orig = T.update_model_config

    def update_model_config(trainer_config):
        orig(trainer_config)
        cmc = trainer_config["model"]["custom_model_config"]
        cmc["motor_decoder_helper_enable"] = True
        if arch.get("mh_hidden"):
            cmc["motor_decoder_helper_layers"] = arch["mh_hidden"]
    T.update_model_config = update_model_config
    try:
        return make_trainer(arch, data, batch, **kw)
    finally:
        T.update_model_config = orig


def _weights(arch):
    h = R.with_helper(arch)
    sd = R.perturb_biases(R.init_state_dict(h, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(h["mh"])
    sd[k_out] = sd[k_out] * 60.0
    return h, sd


def _sp(world, rows, t=1):
    c = R.phase_coeffs(world)
    return make_step_params(lr=5e-4, adam_t=(t, t, t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                            cyc=c["vae_cycle_coeff"], global_rows=rows)


def test_one_minibatch_matches_the_reference_capture(golden):
    g = golden("helper_train_tiny")
    base = arch_from_meta(g["meta"])
    h, sd = _weights(base)
    assert h["mh_range"] == float(g["helper_range"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, base["Db"], base["Da"], kind="dynamics")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    tr = _trainer(base, data, batch, device=DEV)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    eps = torch.from_numpy(g["eps"])
    for world in (True, False):
        tag = "world" if world else "joint"
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, batch, _sp(world, batch), eps=eps.to(DEV),
                                    fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(g[tag + "_total"]), rel=1e-5)
        want = R.loss_and_grads(h, sd, x, y, eps, world)
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        gv = eng.named_views(eng.grads)
        keys = [str(k) for k in g[tag + "_grad_keys"]]
        assert any(k.startswith("_motor_decoder_helper") for k in keys) == (not world)
        for k in keys:
            ref = torch.from_numpy(g["%s_grad::%s" % (tag, k)])
            assert max_err_scaled(gv[k].cpu(), ref) < 1e-4, k
            assert max_err_scaled(gv[k].cpu(), want["grads"][k]) < 1e-4, k
    # a_hat as read back is the HELPED action (what the loss and the world model saw)
    ref_m = R.RefModel(h)
    ref_m.load_state_dict(sd)
    ref_m.eps_source = lambda shape: eps
    with torch.no_grad():
        logits = ref_m(x[:, 0, :])
    assert max_err_scaled(eng.read("a_hat", batch).cpu(), logits[:, : base["Da"]]) < 2e-5
    assert max_err_scaled(eng.read("s2_hat", batch).cpu(), ref_m.cur_future_state) < 2e-5


def test_training_run_matches_the_reference_capture(golden):
    """The reference's own loop (2 world + 3 joint epochs, StepLR tick every 2) at the same weights, minibatches and eps
    stream: epoch losses, final weights of every stack incl. the helper, per-stack Adam step counts."""
    g = golden("helper_train_tiny")
    base = arch_from_meta(g["meta"])
    h, sd = _weights(base)
    n_ep, n_steps, batch, m_world, n_epochs = [int(v) for v in g["meta"][9:14]]
    data = R.synth_demo(0, n_ep, n_steps, base["Db"], base["Da"], kind="dynamics")
    tr = _trainer(base, data, batch, m_world=m_world, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, base["Z"]))
    tr.model.load_state_dict(sd)
    before = {k: v.clone() for k, v in tr.model.state_dict().items()}
    ours = []
    for e in range(n_epochs):
        ours.append(tr.train()["mean_train_loss"])
        if e + 1 == m_world:                         # the world phase leaves the helper (and TE / MD) alone
            now = tr.model.state_dict()
            for k in before:
                if not k.startswith(("_world_model", "_value_branch")):
                    assert torch.equal(now[k], before[k]), k
    np.testing.assert_allclose(ours, g["epoch_losses"], rtol=1e-3)
    for k, v in tr.model.state_dict().items():
        if not k.startswith("_value_branch"):
            assert max_err_scaled(v.cpu(), g["final::" + k]) < 5e-3, k
    nb = len(tr.train_loader)
    assert tr.optimizer.net_steps[_lib.NET_WM] == nb * m_world
    assert tr.optimizer.net_steps[_lib.NET_MD] == tr.optimizer.net_steps[_lib.NET_MH] == nb * (n_epochs - m_world)
    steps = dict(zip((str(k) for k in g["adam_keys"]), g["adam_steps"]))
    assert steps["_motor_decoder_helper._model.0._model.0.weight"] == tr.optimizer.net_steps[_lib.NET_MH]
    # checkpoint round trip (model.pth holds the helper; the reference writes no separate helper file, tpv:440-467)
    import os
    import tempfile
    d = tempfile.mkdtemp()
    tr.save_checkpoint(d)
    tr2 = _trainer(base, data, batch, device=DEV)
    tr2.restore(os.path.join(d, "model.pth"))
    for k, v in tr.model.state_dict().items():
        assert torch.equal(v.cpu(), tr2.model.state_dict()[k].cpu()), k


@pytest.mark.parametrize("dims", ["baseline", "tiny_tanh"])
def test_fused_steps_track_the_oracle(dims):
    """Several fused optimizer steps of the joint phase at the BASELINE dims (197 / 45, the default helper 2x128 relu) and on
    a tiny model whose helper has tanh hidden layers: parameters and both Adam moments against the oracle's graph +
    torch.optim.Adam; then with the helper frozen (`set_learnable_motor_decoder_helper(False)`, adam_t[PVAE_NET_MH] = 0)
    its tensors do not move while its term still shapes every other gradient."""
    if dims == "baseline":
        base = R.make_arch(197, 45, latent=32, te=(128, 2), md=(128, 2), wm=(128, 2))
        rows, n_steps = 64, 200
    else:
        base = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
        base = dict(base, mh_hidden=[{"type": "fc", "hidden_size": 24, "activation": "tanh"},
                                     {"type": "fc", "hidden_size": 16, "activation": "relu"},
                                     {"type": "fc", "hidden_size": "output", "activation": "tanh"}])
        rows, n_steps = 8, 30
    h = R.with_helper(base, hidden=[(24, "tanh"), (16, "relu")]) if dims == "tiny_tanh" else R.with_helper(base)
    sd = R.perturb_biases(R.init_state_dict(h, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(h["mh"])
    sd[k_out] = sd[k_out] * 60.0
    data = R.synth_demo(0, 2, n_steps, base["Db"], base["Da"], kind="dynamics")
    X, Y = R.build_windows(data)
    tr = _trainer(base, data, rows, device=DEV)
    eng = tr.engine
    es = R.eps_stream(2, base["Z"])
    for frozen in (False, True):
        tr.model.load_state_dict(sd)
        eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
        ref = R.RefModel(h)
        ref.load_state_dict(sd)
        ref.set_learnable("_world_model", False)
        ref.set_learnable("_value_branch", False)
        if frozen:
            ref.set_learnable("_motor_decoder_helper", False)
        opt = torch.optim.Adam([p for p in ref.parameters()], lr=5e-4)
        K = 4
        for i in range(K):
            x, y = X[i * rows: (i + 1) * rows], Y[i * rows: (i + 1) * rows]
            x, y = torch.as_tensor(x, dtype=torch.float32), torch.as_tensor(y, dtype=torch.float32)
            eps = es(i, (rows, base["Z"]))
            ref.eps_source = lambda shape, e=eps: e
            opt.zero_grad()
            total, _ = R.compute_loss(ref, x, y, R.phase_coeffs(False))
            total.backward()
            opt.step()
            sp = _sp(False, rows, t=i + 1)
            if frozen:
                sp.adam_t[_lib.NET_MH] = 0
            eng.set_batch(x, y)
            loss = eng.forward_backward(_lib.PHASE_JOINT, rows, sp, eps=eps.to(DEV), fused_adam=True).cpu()
            assert float(loss[0]) == pytest.approx(float(total), rel=2e-4)
        ref_sd = ref.state_dict()
        got = tr.model.state_dict()
        for k, v in got.items():
            if k.startswith(("_value_branch", "_world_model")):
                continue
            if frozen and k.startswith("_motor_decoder_helper"):
                assert torch.equal(v.cpu(), sd[k]), k
            else:
                assert max_err_scaled(v.cpu(), ref_sd[k]) < 2e-3, k
                assert not torch.equal(v.cpu(), sd[k]), k


def test_data_parallel_step_equals_fused_step_with_a_helper():
    """One-rank peer-mapped exchange: gradient store, exchange (helper | decoder | encoder buckets), flat Adam == the fused
    step, bit for bit; rollout forward of the trained model through the library's own path (pvae_infer_logits adds the
    helper's term) == the sub-module route."""
    base = R.make_arch(197, 45, latent=32, te=(128, 2), md=(128, 2), wm=(128, 2))
    h, sd = _weights(base)
    rows = 64
    data = R.synth_demo(0, 2, 200, 197, 45, kind="dynamics")
    X, Y = R.build_windows(data)
    tr = _trainer(base, data, rows, device=DEV)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.p2p_open(0, 1, [eng.p2p_export()])
    eng.comm_mode("p2p")
    es = R.eps_stream(2, 32)
    outs = []
    for dp in (False, True):
        tr.model.load_state_dict(sd)
        eng.exp_avg.zero_(); eng.exp_avg_sq.zero_(); eng.invalidate_staging()
        for i in range(3):
            sp = _sp(False, rows, t=i + 1)
            eps = es(i, (rows, 32)).to(DEV)
            if dp:
                eng.dp_train_step(_lib.PHASE_JOINT, i * rows, rows, sp, eps=eps, next_span=((i + 1) * rows, rows))
            else:
                eng.train_step(_lib.PHASE_JOINT, i * rows, rows, sp, eps=eps, next_span=((i + 1) * rows, rows))
        torch.cuda.synchronize()
        outs.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    eng.p2p_close()
    assert not torch.equal(outs[0][0], torch.zeros_like(outs[0][0]))
    tr.model.latent_prior_noise = False
    obs = torch.as_tensor(X[:3, 0, :], dtype=torch.float32).to(DEV)
    with torch.no_grad():
        logits, _ = tr.model.forward({"obs_flat": obs}, [], None)
        zb, zt, _ = tr.model.forward_encoder(obs)
        lg, _ = tr.model.forward_decoder(zb, zt)
    assert max_err_scaled(logits.cpu(), lg.cpu()) < 2e-5
    ref = R.RefModel(h)
    ref.load_state_dict({k: v.cpu() for k, v in tr.model.state_dict().items()})
    ref.latent_prior_noise = False
    with torch.no_grad():
        want = ref(obs.cpu())
    assert max_err_scaled(logits.cpu(), want) < 2e-5
    assert max_err_scaled(tr.model._cur_future_state.cpu(), ref.cur_future_state) < 2e-5


@pytest.mark.parametrize("md_in", [("task",), ("body",)])
def test_helper_on_a_decoder_input_subset(md_in):
    """The helper reads what the decoder reads (rmt:646-653, 674-680): with `motor_decoder_inputs` a subset, its first layer
    is the same column window of the shared input panel.  One minibatch of both phases and four fused joint steps against
    the oracle (restated model: narrower first layers for decoder AND helper)."""
    base = R.with_inputs(R.make_arch(197, 45, latent=32, te=(128, 2), md=(128, 2), wm=(128, 2)), ("body", "task"), md_in)
    h = R.with_helper(base)
    sd = R.perturb_biases(R.init_state_dict(h, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(h["mh"])
    sd[k_out] = sd[k_out] * 60.0
    rows = 64
    data = R.synth_demo(0, 2, 200, 197, 45, kind="dynamics")
    X, Y = R.build_windows(data)
    tr = _trainer(base, data, rows, device=DEV)
    assert [(k, tuple(v.shape)) for k, v in tr.model.state_dict().items()] == R.state_dict_spec(h)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    x, y = next(iter(R.make_loader(X, Y, rows)))
    es = R.eps_stream(2, 32)
    eps = es(0, (rows, 32))
    for world in (True, False):
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, rows, _sp(world, rows), eps=eps.to(DEV),
                                    fused_adam=False).cpu()
        keep = R.relu_kink_margin(h, sd, x, y, eps, world) > 4e-6
        want = R.loss_and_grads(h, sd, x, y, eps, world)
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        if bool(keep.all()):
            gv = eng.named_views(eng.grads)
            for k, gr in want["grads"].items():
                assert max_err_scaled(gv[k].cpu(), gr) < 1e-4, k
    ref = R.RefModel(h)
    ref.load_state_dict(sd)
    ref.set_learnable("_world_model", False)
    ref.set_learnable("_value_branch", False)
    opt = torch.optim.Adam(list(ref.parameters()), lr=5e-4)
    eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
    for i in range(4):
        xi = torch.as_tensor(X[i * rows: (i + 1) * rows], dtype=torch.float32)
        yi = torch.as_tensor(Y[i * rows: (i + 1) * rows], dtype=torch.float32)
        e = es(i, (rows, 32))
        ref.eps_source = lambda shape, e=e: e
        opt.zero_grad()
        total, _ = R.compute_loss(ref, xi, yi, R.phase_coeffs(False))
        total.backward()
        opt.step()
        eng.set_batch(xi, yi)
        loss = eng.forward_backward(_lib.PHASE_JOINT, rows, _sp(False, rows, t=i + 1), eps=e.to(DEV), fused_adam=True).cpu()
        assert float(loss[0]) == pytest.approx(float(total.detach()), rel=2e-4)
    ref_sd = ref.state_dict()
    for k, v in tr.model.state_dict().items():
        if not k.startswith(("_value_branch", "_world_model")):
            assert max_err_scaled(v.cpu(), ref_sd[k]) < 2e-3, k
    # structural zeros of BOTH first layers on the shared panel
    for info in eng.layers:
        if info["index"] == 0 and info["net"] in (_lib.NET_MD, _lib.NET_MH):
            blk = eng.params[info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]].view(info["n_out_pad"], info["ld"])
            out = torch.cat([blk[:, : info["col0"]].reshape(-1), blk[:, info["col0"] + info["n_in"]:].reshape(-1)])
            assert float(out.abs().max()) == 0.0


def test_helper_with_lookahead_two_matches_the_reference_capture(golden):
    """`motor_decoder_helper_enable` with lookahead 2 (tpv:367-428; rmt:833-835 inside every unrolled step), against the
    reference's own compute_loss / backward / training loop (tests/golden/helper_train_look2_tiny.npz).  One minibatch in
    both phases: total and every gradient -- in the WORLD phase the helper has one too (the state the world model continues
    from is its own prediction under the helped action, tpv:417-421), the frozen decoder and encoder none.  Then the
    five-epoch run across the switch: epoch losses, final weights, and Adam's counters -- the helper's runs from the first
    epoch, the decoder's from the switch."""
    g = golden("helper_train_look2_tiny")
    base = arch_from_meta(g["meta"])
    h, sd = _weights(base)
    n_ep, n_steps, batch, m_world, n_epochs, L = [int(v) for v in g["meta"][9:15]]
    assert L == 2
    data = R.synth_demo(0, n_ep, n_steps, base["Db"], base["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    tr = _trainer(base, data, batch, device=DEV, extra={"lookahead": L})
    tr.model.load_state_dict(sd)
    eng = tr.engine
    es = R.eps_stream(2, base["Z"])
    eps = torch.stack([es(t, (batch, base["Z"])) for t in range(L)])
    for world in (True, False):
        tag = "world" if world else "joint"
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, batch, _sp(world, batch), eps=eps.to(DEV),
                                    fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(g[tag + "_total"]), rel=1e-5)
        gv = eng.named_views(eng.grads)
        keys = [str(k) for k in g[tag + "_grad_keys"]]
        assert any(k.startswith("_motor_decoder_helper") for k in keys)
        if world:
            assert not any(k.startswith(("_motor_decoder.", "_task_encoder")) for k in keys)
        for k in keys:
            ref = torch.from_numpy(g["%s_grad::%s" % (tag, k)])
            assert max_err_scaled(gv[k].cpu(), ref) < 1e-4, (tag, k)
    # the reference's loop
    tr = _trainer(base, data, batch, m_world=m_world, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, base["Z"]),
                  extra={"lookahead": L})
    tr.model.load_state_dict(sd)
    before = {k: v.clone() for k, v in tr.model.state_dict().items()}
    ours = []
    for e in range(n_epochs):
        ours.append(tr.train()["mean_train_loss"])
        if e + 1 == m_world:                         # the world phase moved the world model AND the helper, nothing else
            now = tr.model.state_dict()
            for k in before:
                moved = not torch.equal(now[k], before[k])
                assert moved == k.startswith(("_world_model", "_motor_decoder_helper")), k
    np.testing.assert_allclose(ours, g["epoch_losses"], rtol=1e-3)
    for k, v in tr.model.state_dict().items():
        if not k.startswith("_value_branch"):
            assert max_err_scaled(v.cpu(), g["final::" + k]) < 5e-3, k
    nb = len(tr.train_loader)
    steps = dict(zip((str(k) for k in g["adam_keys"]), g["adam_steps"]))
    assert tr.optimizer.net_steps[_lib.NET_WM] == nb * m_world == steps["_world_model._model.0._model.0.weight"]
    assert tr.optimizer.net_steps[_lib.NET_MH] == nb * n_epochs == steps["_motor_decoder_helper._model.0._model.0.weight"]
    assert tr.optimizer.net_steps[_lib.NET_MD] == nb * (n_epochs - m_world) == steps["_motor_decoder._model.0._model.0.weight"]


def test_helper_with_lookahead_fused_equals_stored_gradient_plus_flat_adam():
    """lookahead 3 with a helper at wider dims: the fused step (weight gradients with Adam in their launches, the paired
    schedule of step 0) and the gradient-store step followed by the flat Adam kernel agree bit for bit in both phases, over
    three steps; with the helper frozen (adam_t[PVAE_NET_MH] = 0) its tensors do not move."""
    base = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3))
    h, sd = _weights(base)
    data = R.synth_demo(0, 3, 40, 23, 7, kind="dynamics")
    tr = _trainer(base, data, 32, device=DEV, extra={"lookahead": 3})
    eng = tr.engine
    X, Y = R.build_windows(data, lookahead=3)
    x, y = next(iter(R.make_loader(X, Y, 32)))
    es = R.eps_stream(2, 8)
    eps = torch.stack([es(t, (32, 8)) for t in range(3)]).to(DEV)
    mh_off, mh_cnt = eng.segments[_lib.NET_MH]
    for world in (True, False):
        phase = _lib.PHASE_WORLD if world else _lib.PHASE_JOINT
        nets = ([_lib.NET_WM] if world else [_lib.NET_TE, _lib.NET_MD]) + [_lib.NET_MH]
        for frozen in (False, True):
            res = []
            for fused in (True, False):
                tr.model.load_state_dict(sd)
                eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
                for t in (1, 2, 3):
                    sp = _sp(world, 32, t)
                    if frozen:
                        sp.adam_t[_lib.NET_MH] = 0
                    eng.set_batch(x, y)
                    eng.forward_backward(phase, 32, sp, eps=eps, fused_adam=fused)
                    if not fused:
                        eng.adam([n for n in nets if not (frozen and n == _lib.NET_MH)], sp)
                res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()))
            for a, b in zip(*res):
                assert torch.equal(a, b), (world, frozen)
            p0 = eng.params.clone()
            tr.model.load_state_dict(sd)
            moved = not torch.equal(res[0][0][mh_off: mh_off + mh_cnt], eng.params[mh_off: mh_off + mh_cnt])
            assert moved == (not frozen), (world, frozen)
            del p0
