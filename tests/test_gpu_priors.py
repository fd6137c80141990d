"""The two latent priors the reference sketches but cannot run ("normal_state_mean_one_std",
"hypersphere_uniform"; rmt:614-635, 795-819, tpv:390-409), built to the specification in
oracle/refpath.py (PRIORS): the HIP path against that restatement -- one minibatch (losses, forward
internals, every gradient, both launch schedules), a training run across the phase switch, the
data-parallel step, and the module's rollout forward.  No reference capture can exist for these two
options (upstream raises before the first loss), so the oracle alone is the checker here."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import make_trainer, max_err_scaled, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"

ARCHS = {
    "tiny": dict(dims=(23, 7, 8), te=(64, 2), md=(96, 2), wm=(128, 3), pr=(48, 1), batch=32, demo=(3, 40)),
    "c1": dict(dims=(197, 45, 32), te=(256, 2), md=(256, 2), wm=(256, 2), pr=(256, 2), batch=64, demo=(2, 100)),
}


def _setup(prior, size):
    a = ARCHS[size]
    Db, Da, Z = a["dims"]
    arch = R.make_arch(Db, Da, latent=Z, te=a["te"], md=a["md"], wm=a["wm"], prior=prior, pr=a["pr"])
    data = R.synth_demo(0, a["demo"][0], a["demo"][1], Db, Da, kind="dynamics")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, a["batch"])))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    eps = R.eps_stream(2, Z)(0, (x.shape[0], Z))
    return arch, data, x, y, sd, eps


@pytest.mark.parametrize("paired", ["1", "0"])
@pytest.mark.parametrize("size", ["tiny", "c1"])
@pytest.mark.parametrize("prior", R.PRIORS[1:])
def test_single_batch_matches_the_specification(prior, size, paired, monkeypatch):
    monkeypatch.setenv("PVAE_PAIR", paired)          # fused gradient hand-overs vs stand-alone glue kernels
    arch, data, x, y, sd, eps = _setup(prior, size)
    tr = make_trainer(arch, data, x.shape[0], device=DEV)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    margin = R.relu_kink_margin(arch, sd, x, y, eps, False)
    keep = margin > 4e-6
    x, y, eps = x[keep], y[keep], eps[keep]
    rows = x.shape[0]
    want = R.loss_and_grads(arch, sd, x, y, eps, world=False)
    c = R.phase_coeffs(False)
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=rows)
    eng.set_batch(x, y)
    eng.grads.fill_(float("nan"))
    loss = eng.forward_backward(_lib.PHASE_JOINT, rows, sp, eps=eps, fused_adam=False).cpu()
    assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
    for i, k in enumerate(("loss_a", "loss_kl", "loss_s", "loss_cyc")):
        assert float(loss[1 + i]) == pytest.approx(float(want[k]), rel=2e-5, abs=1e-8), k
    assert max_err_scaled(eng.read("z", rows).cpu(), want["z"]) < 2e-5
    assert max_err_scaled(eng.read("a_hat", rows).cpu(), want["a_hat"]) < 2e-5
    assert max_err_scaled(eng.read("s2_hat", rows).cpu(), want["future_state"]) < 2e-5
    if prior == R.PRIORS[1]:
        assert max_err_scaled(eng.read("mu", rows).cpu(), want["mu"]) < 2e-5
        assert max_err_scaled(eng.read("logvar", rows).cpu(), want["logvar"]) < 2e-5
        assert max_err_scaled(eng.read("prior_mu", rows).cpu(), want["prior_mu"]) < 2e-5
        assert torch.equal(eng.read("eps", rows).cpu(), eps)
    else:
        assert max_err_scaled(eng.read("eps", rows).cpu(), want["prior_mu"]) < 2e-6      # the unit prior sample
        assert torch.allclose(eng.read("z", rows).norm(dim=1).cpu(), torch.ones(rows), atol=1e-5)
    gv = eng.named_views(eng.grads)
    trainable = [k for k in sd if k.startswith(("_task_encoder", "_motor_decoder", "_latent_prior"))]
    assert sorted(want["grads"].keys()) == sorted(trainable)
    for k, gr in want["grads"].items():
        ours = gv[k].cpu()
        assert torch.isfinite(ours).all(), k
        assert max_err_scaled(ours, gr) < 1e-4, k
        assert rel_err(ours, gr) < 2e-4, k
    # the world phase does not involve the prior at all
    cw = R.phase_coeffs(True)
    spw = make_step_params(lr=5e-4, a_rec=cw["a_rec_coeff"], kl=cw["vae_kl_coeff"], s_rec=cw["s_rec_coeff"],
                           cyc=cw["vae_cycle_coeff"], global_rows=rows)
    wantw = R.loss_and_grads(arch, sd, x, y, eps, world=True)
    eng.set_batch(x, y)                      # (the joint forward left a_hat in the world model's action columns)
    lossw = eng.forward_backward(_lib.PHASE_WORLD, rows, spw, fused_adam=False).cpu()
    assert float(lossw[0]) == pytest.approx(float(wantw["total"]), rel=1e-5)


def test_zero_prior_mean_reproduces_the_default_prior():
    """mu_p == 0 (output layer of the prior stack zeroed) must give the default prior's losses and
    encoder/decoder gradients: the specification keeps the default's normalisation."""
    arch1, data, x, y, sd1, eps = _setup(R.PRIORS[1], "tiny")
    for k in sd1:
        if k.startswith("_latent_prior._model.1."):           # pr = (48, 1): layer 1 is the output layer
            sd1[k] = torch.zeros_like(sd1[k])
    arch0 = dict(arch1, prior=R.PRIORS[0])
    sd0 = {k: v for k, v in sd1.items() if not k.startswith("_latent_prior")}
    rows = x.shape[0]
    c = R.phase_coeffs(False)
    sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                          cyc=c["vae_cycle_coeff"], global_rows=rows)
    out = []
    for arch, sd in ((arch1, sd1), (arch0, sd0)):
        tr = make_trainer(arch, data, rows, device=DEV)
        tr.model.load_state_dict(sd)
        tr.engine.set_batch(x, y)
        loss = tr.engine.forward_backward(_lib.PHASE_JOINT, rows, sp, eps=eps, fused_adam=False).cpu()
        gv = {k: v.cpu().clone() for k, v in tr.engine.named_views(tr.engine.grads).items()
              if k.startswith(("_task_encoder", "_motor_decoder"))}
        out.append((loss, gv))
    assert torch.allclose(out[0][0], out[1][0], rtol=1e-6, atol=1e-9)
    for k in out[1][1]:
        assert max_err_scaled(out[0][1][k], out[1][1][k]) < 1e-6, k


@pytest.mark.parametrize("prior", R.PRIORS[1:])
def test_training_run_tracks_the_specification(prior):
    """Two world epochs + three joint epochs (phase switch, lazy Adam state for TE / MD / prior, StepLR
    tick) against the oracle's trainer with the same eps stream."""
    arch, data, x, y, sd, eps = _setup(prior, "tiny")
    batch, m_world, n_epochs = 32, 2, 5
    X, Y = R.build_windows(data)
    ref = R.RefTrainer(arch, sd, X, Y, batch, max_iter_world_model=m_world, lr_step=2,
                       eps_fn=R.eps_stream(2, arch["Z"]))
    tr = make_trainer(arch, data, batch, m_world=m_world, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, arch["Z"]))
    tr.model.load_state_dict(sd)
    ours, theirs = [], []
    for e in range(n_epochs):
        theirs.append(ref.step()["mean_train_loss"])
        ours.append(tr.train()["mean_train_loss"])
    np.testing.assert_allclose(ours, theirs, rtol=1e-3)
    nb = len(tr.train_loader)
    assert tr.optimizer.net_steps[_lib.NET_WM] == nb * m_world
    assert tr.optimizer.net_steps[_lib.NET_TE] == nb * (n_epochs - m_world)
    assert tr.optimizer.net_steps[_lib.NET_PR] == (nb * (n_epochs - m_world) if prior == R.PRIORS[1] else 0)
    ref_sd = ref.model.state_dict()
    for k, v in tr.model.state_dict().items():
        if k.startswith("_value_branch"):
            continue
        assert max_err_scaled(v.cpu(), ref_sd[k]) < 5e-3, k
    # the five checkpoint files + the prior's own file round-trip
    import os
    import tempfile
    d = tempfile.mkdtemp()
    tr.save_checkpoint(d)                        # (writes latent_prior.pt itself when the model has one, tpv:462-466)
    assert os.path.exists(os.path.join(d, "latent_prior.pt")) == (prior == R.PRIORS[1])
    tr2 = make_trainer(arch, data, batch, device=DEV)
    tr2.restore(os.path.join(d, "model.pth"))
    for k, v in tr.model.state_dict().items():
        assert torch.equal(v.cpu(), tr2.model.state_dict()[k].cpu()), k
    if prior == R.PRIORS[1]:
        got = torch.load(os.path.join(d, "latent_prior.pt"))
        assert list(got) == [k[len("_latent_prior."):] for k in sd if k.startswith("_latent_prior")]


@pytest.mark.parametrize("prior", R.PRIORS[1:])
def test_data_parallel_step_equals_fused_step(prior):
    """One-rank RCCL communicator: the data-parallel step (gradient store, in-place reduction, flat Adam
    over TE | MD | prior) equals the fused single-GPU step bit for bit."""
    arch, data, x, y, sd, eps = _setup(prior, "tiny")
    tr = make_trainer(arch, data, 32, device=DEV)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.comm_init(0, 1, eng.comm_unique_id())
    c = R.phase_coeffs(False)
    res = []
    for dp in (False, True):
        tr.model.load_state_dict(sd)
        eng.exp_avg.zero_(); eng.exp_avg_sq.zero_()
        out = torch.zeros(5, device=DEV)
        for t in (1, 2, 3):
            sp = make_step_params(lr=5e-4, adam_t=(t, t, t, t), a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"],
                                  s_rec=c["s_rec_coeff"], cyc=c["vae_cycle_coeff"], global_rows=32)
            (eng.dp_train_step if dp else eng.train_step)(_lib.PHASE_JOINT, 32 * (t - 1), 32, sp, eps=eps, loss_out=out)
        res.append((eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), out.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    eng.comm_destroy()


@pytest.mark.parametrize("rows", [1, 5])
@pytest.mark.parametrize("prior", R.PRIORS[1:])
def test_module_forward_matches_the_specification(prior, rows):
    arch, data, x, y, sd, eps = _setup(prior, "tiny")
    tr = make_trainer(arch, data, 32, device=DEV)
    tr.model.load_state_dict(sd)
    ref = R.RefModel(arch)
    ref.load_state_dict(sd)
    obs = x[:rows, 0, :]
    e = eps[:rows]
    ref.eps_source = lambda shape: e
    with torch.no_grad():
        want = ref(obs)
    m = tr.model
    logits, _ = m.forward({"obs_flat": obs.to(DEV)}, [], None, eps=e.to(DEV))
    assert max_err_scaled(logits.cpu(), want) < 2e-5
    assert max_err_scaled(m._cur_future_state.cpu(), ref.cur_future_state) < 2e-5
    assert max_err_scaled(m.task_encoder_variable().cpu(), ref.cur_z) < 2e-5
    assert max_err_scaled(m._cur_latent_prior_mu.cpu(), ref.cur_prior_mu) < 2e-5
    a_hat, s2, z = tr.engine.infer(obs.to(DEV), eps=e.to(DEV), noise=True, want_s2=True)
    assert max_err_scaled(a_hat.cpu(), want[:, : arch["Da"]]) < 2e-5 and max_err_scaled(z.cpu(), ref.cur_z) < 2e-5


@pytest.mark.parametrize("paired", ["1", "0"])
def test_no_prior_mode_matches_the_reference_capture(golden, paired, monkeypatch):
    """latent_prior_type = False -- a mode the reference runs (rmt:622-623, 815-816): the encoder's Z outputs go
    to the decoder as they are, no sampling, no KL term.  Total, internals and every gradient of one minibatch
    against the capture of the reference itself (`noprior_tiny.npz`) and against the oracle, both phases."""
    from util import arch_from_meta
    monkeypatch.setenv("PVAE_PAIR", paired)
    g = golden("noprior_tiny")
    arch = dict(arch_from_meta(g["meta"]), prior=False)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = make_trainer(arch, data, batch, device=DEV)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    for world in (True, False):
        tag = "world" if world else "joint"
        c = R.phase_coeffs(world)
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=batch)
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, batch, sp, fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(g[tag + "_total"]), rel=1e-5)
        assert float(loss[2]) == 0.0                                           # no KL term
        want = R.loss_and_grads(arch, sd, x, y, None, world)
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        if not world:
            assert max_err_scaled(eng.read("z", batch).cpu(), g["joint_z"]) < 2e-5
            assert max_err_scaled(eng.read("s2_hat", batch).cpu(), g["joint_future_state"]) < 2e-5
        gv = eng.named_views(eng.grads)
        assert sorted(str(k) for k in g[tag + "_grad_keys"]) == sorted(want["grads"].keys())
        for k in g[tag + "_grad_keys"]:
            k = str(k)
            ref = torch.from_numpy(g["%s_grad::%s" % (tag, k)])
            assert max_err_scaled(gv[k].cpu(), ref) < 1e-4, k
    # module forward: the code is the encoder output
    ref_m = R.RefModel(arch)
    ref_m.load_state_dict(sd)
    obs = x[:3, 0, :]
    with torch.no_grad():
        want_logits = ref_m(obs)
    logits, _ = tr.model.forward({"obs_flat": obs.to(DEV)}, [], None)
    assert max_err_scaled(logits.cpu(), want_logits) < 2e-5
    assert max_err_scaled(tr.model.task_encoder_variable().cpu(), ref_m.cur_z) < 2e-5
    r1, r2 = tr.train(), tr.train()
    assert np.isfinite(r1["mean_train_loss"]) and np.isfinite(r2["mean_train_loss"])


@pytest.mark.parametrize("prior", [False, R.PRIORS[2]])
@pytest.mark.parametrize("rows", [1, 3, 4, 6])
def test_rollout_forward_of_the_sample_free_priors(prior, rows):
    """The rollout path (<= 4 rows: input assembly inside the layer launches, 7 launches) for the two priors
    whose code is a function of the encoder output alone -- `False`: z = e; hypersphere: z = e / |e| -- against
    the oracle's module forward, and against the staged path the larger batches take (6 rows)."""
    arch = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2), prior=prior)
    data = R.synth_demo(0, 2, 14, 7, 3, kind="iid")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = make_trainer(arch, data, 8, device=DEV)
    tr.model.load_state_dict(sd)
    ref = R.RefModel(arch)
    ref.load_state_dict(sd)
    obs = torch.randn(rows, 14, generator=torch.Generator().manual_seed(rows))
    e = torch.randn(rows, 4, generator=torch.Generator().manual_seed(7))
    ref.eps_source = lambda shape: e
    with torch.no_grad():
        want = ref(obs)
    a_hat, s2, z = tr.engine.infer(obs.to(DEV), eps=e.to(DEV), noise=True, want_s2=True)
    assert max_err_scaled(a_hat.cpu(), want[:, :3]) < 2e-5 and max_err_scaled(z.cpu(), ref.cur_z) < 2e-5
    assert max_err_scaled(s2.cpu(), ref.cur_future_state) < 2e-5
    a2, none, z2 = tr.engine.infer(obs.to(DEV), noise=False, want_s2=False)
    assert none is None and max_err_scaled(a2.cpu(), want[:, :3]) < 2e-5
