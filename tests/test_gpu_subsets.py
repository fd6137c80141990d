"""`task_encoder_inputs` / `motor_decoder_inputs` (rmt:470, 485; 607-613, 646-653, 776-783, 822-829): subsets of
["body", "task"] as COLUMN WINDOWS of the full-width first layers (include/pvae.h pvae_config.te_inputs / md_inputs).

The HIP step against the capture of the reference itself (`subsets_tiny.npz`, oracle/gen_golden.py case_subsets) and
against the oracle, at the captured dims and at the BASELINE dims; whole training runs against the oracle's trainer;
the rollout forward (staged, fused <= 4 rows, served) against the oracle's module; lookahead > 1; and the invariant the
design rests on -- the columns outside a window are EXACTLY zero after training (their operand is staged as zeros, so
their gradient is exactly zero and Adam never moves them)."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import arch_from_meta, make_trainer, max_err_scaled

pytestmark = pytest.mark.gpu
DEV = "cuda"
COMBOS = [(("task",), ("body", "task")), (("body",), ("task",)), (("body", "task"), ("body",)), (("task",), ("task",))]


def _outside_window(eng, arena):
    """max |value| of `arena` (parameters, gradients or a moment) in the columns a subset leaves out."""
    worst = 0.0
    for info in eng.layers:
        if info["index"] == 0 and info["net"] in (_lib.NET_TE, _lib.NET_MD):
            blk = arena[info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]].view(info["n_out_pad"], info["ld"])
            for part in (blk[:, : info["col0"]], blk[:, info["col0"] + info["n_in"]:], blk[info["n_out"]:]):
                if part.numel():
                    worst = max(worst, float(part.abs().max()))
    return worst


@pytest.mark.parametrize("ci", range(4))
def test_one_minibatch_matches_the_reference_capture(golden, ci):
    g = golden("subsets_tiny")
    base = arch_from_meta(g["meta"])
    te_in, md_in = (tuple(part.split("+")) for part in str(g["combos"][ci]).split("/"))
    assert (te_in, md_in) == COMBOS[ci]
    arch = R.with_inputs(base, te_in, md_in)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = make_trainer(arch, data, batch, device=DEV)
    tr.model.load_state_dict(sd)
    eng = tr.engine
    pre = "c%d_" % ci
    eps = torch.from_numpy(g[pre + "eps"])
    for world in (True, False):
        tag = pre + ("world" if world else "joint")
        c = R.phase_coeffs(world)
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=batch)
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, batch, sp, eps=eps.to(DEV),
                                    fused_adam=False).cpu()
        assert float(loss[0]) == pytest.approx(float(g[tag + "_total"]), rel=1e-5)
        want = R.loss_and_grads(arch, sd, x, y, eps, world)
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        if not world:        # (the world phase runs the world model alone: its loss terms need nothing else, tpv:331-335)
            assert max_err_scaled(eng.read("z", batch).cpu(), g[tag + "_z"]) < 2e-5
            assert max_err_scaled(eng.read("s2_hat", batch).cpu(), g[tag + "_future_state"]) < 2e-5
        gv = eng.named_views(eng.grads)
        for k in g[tag + "_grad_keys"]:
            k = str(k)
            ref = torch.from_numpy(g["%s_grad::%s" % (tag, k)])
            assert gv[k].shape == ref.shape, k
            assert max_err_scaled(gv[k].cpu(), ref) < 1e-4, k
        # the gradient of a column outside the window is exactly zero (not small: zero)
        nets = (_lib.NET_WM,) if world else (_lib.NET_TE, _lib.NET_MD)
        for info in eng.layers:
            if info["index"] == 0 and info["net"] in nets and info["net"] != _lib.NET_WM:
                blk = eng.grads[info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]].view(info["n_out_pad"], info["ld"])
                out = torch.cat([blk[:, : info["col0"]].reshape(-1), blk[:, info["col0"] + info["n_in"]:].reshape(-1)])
                assert out.numel() == 0 or float(out.abs().max()) == 0.0
    # module forward (staged path at 8 rows, the fused rollout launches at 3): the oracle's module on the same weights
    ref_m = R.RefModel(arch)
    ref_m.load_state_dict(sd)
    ref_m.latent_prior_noise = False
    tr.model.latent_prior_noise = False
    for rows in (3, 8):
        obs = x[:rows, 0, :]
        with torch.no_grad():
            want_logits = ref_m(obs)
        logits, _ = tr.model.forward({"obs_flat": obs.to(DEV)}, [], None)
        assert max_err_scaled(logits.cpu(), want_logits) < 2e-5
        assert max_err_scaled(tr.model.task_encoder_variable().cpu(), ref_m.cur_z) < 2e-5
        assert max_err_scaled(tr.model._cur_future_state.cpu(), ref_m.cur_future_state) < 2e-5
    # the sub-module entry points take what the reference's take (rmt:773-837)
    with torch.no_grad():
        zb, zt, _ = tr.model.forward_encoder(obs.to(DEV))
        lg, _ = tr.model.forward_decoder(zb, zt)
    assert max_err_scaled(lg.cpu(), want_logits) < 2e-5


@pytest.mark.parametrize("te_in,md_in", COMBOS)
def test_training_run_tracks_the_oracle_and_keeps_the_structural_zeros(te_in, md_in):
    """Two world epochs + three joint epochs at the BASELINE dims (197 / 45: the [s_t | .] boundary inside a 16-byte
    chunk) against the oracle's trainer on the same eps stream; afterwards every column outside a window is still
    exactly zero -- in the parameters AND in both Adam moments."""
    arch = R.with_inputs(R.make_arch(197, 45, latent=32, te=(128, 2), md=(128, 2), wm=(128, 2)), te_in, md_in)
    data = R.synth_demo(0, 2, 80, 197, 45, kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    batch, m_world, n_epochs = 32, 2, 5
    X, Y = R.build_windows(data)
    ref = R.RefTrainer(arch, sd, X, Y, batch, max_iter_world_model=m_world, lr_step=2, eps_fn=R.eps_stream(2, 32))
    tr = make_trainer(arch, data, batch, m_world=m_world, device=DEV, lr_step=2, eps_fn=R.eps_stream(2, 32))
    tr.model.load_state_dict(sd)
    ours, theirs = [], []
    for e in range(n_epochs):
        theirs.append(ref.step()["mean_train_loss"])
        ours.append(tr.train()["mean_train_loss"])
    np.testing.assert_allclose(ours, theirs, rtol=1e-3)
    ref_sd = ref.model.state_dict()
    got = tr.model.state_dict()
    assert [(k, tuple(v.shape)) for k, v in got.items()] == [(k, tuple(v.shape)) for k, v in ref_sd.items()]
    for k, v in got.items():
        if not k.startswith("_value_branch"):
            assert max_err_scaled(v.cpu(), ref_sd[k]) < 5e-3, k
    eng = tr.engine
    assert _outside_window(eng, eng.params) == 0.0
    assert _outside_window(eng, eng.exp_avg) == 0.0 and _outside_window(eng, eng.exp_avg_sq) == 0.0
    # checkpoints carry the reference's shapes and load back bit for bit
    import os
    import tempfile
    d = tempfile.mkdtemp()
    tr.save_checkpoint(d)
    te_file = torch.load(os.path.join(d, "task_encoder.pt"))["task_encoder"]
    assert tuple(te_file["_model.0._model.0.weight"].shape) == tuple(ref_sd["_task_encoder._model.0._model.0.weight"].shape)
    tr2 = make_trainer(arch, data, batch, device=DEV)
    tr2.restore(os.path.join(d, "model.pth"))
    for k, v in got.items():
        assert torch.equal(v.cpu(), tr2.model.state_dict()[k].cpu()), k
    assert _outside_window(tr2.engine, tr2.engine.params) == 0.0


def test_lookahead_unroll_with_subsets():
    """lookahead 2: the world model's prediction becomes s_t of the next step only in the panels of the stacks that read
    s_t (tpv:421); against the oracle's unrolled graph."""
    arch = R.with_inputs(R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2)), ("task",), ("task",))
    data = R.synth_demo(0, 2, 14, 7, 3, kind="dynamics")
    L, batch = 2, 8
    X, Y = R.build_windows(data, lookahead=L)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = make_trainer(arch, data, batch, device=DEV, extra={"lookahead": L})
    tr.model.load_state_dict(sd)
    eng = tr.engine
    es = R.eps_stream(2, arch["Z"])
    eps = torch.stack([es(t, (batch, arch["Z"])) for t in range(L)])
    for world in (True, False):
        c = R.phase_coeffs(world)
        sp = make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                              cyc=c["vae_cycle_coeff"], global_rows=batch)
        eng.set_batch(x, y)
        eng.grads.fill_(float("nan"))
        loss = eng.forward_backward(_lib.PHASE_WORLD if world else _lib.PHASE_JOINT, batch, sp, eps=eps.to(DEV),
                                    fused_adam=False).cpu()
        want = R.loss_and_grads(arch, sd, x, y, eps, world)
        assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
        gv = eng.named_views(eng.grads)
        for k, gr in want["grads"].items():
            assert max_err_scaled(gv[k].cpu(), gr) < 1e-4, k
        assert _outside_window(eng, eng.grads) == 0.0 or world


def test_served_forward_and_direct_refusal_with_subsets():
    """The resident rollout kernel serves a subset model through the same structural zeros; the direct first layers
    (pvae_set_direct) decline -- the staged panels are what carries the zeros."""
    arch = R.with_inputs(R.make_arch(197, 45, latent=32, te=(256, 2), md=(256, 2), wm=(256, 2)), ("task",), ("body", "task"))
    data = R.synth_demo(0, 2, 300, 197, 45, kind="dynamics")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = make_trainer(arch, data, 256, device=DEV)
    tr.model.load_state_dict(sd)
    ref_m = R.RefModel(arch)
    ref_m.load_state_dict(sd)
    ref_m.latent_prior_noise = False
    tr.model.latent_prior_noise = False
    X, _ = R.build_windows(data)
    obs = torch.as_tensor(X[:1, 0, :], dtype=torch.float32)
    with torch.no_grad():
        want = ref_m(obs)
    tr.model.start_rollout_server()
    try:
        logits, _ = tr.model.forward({"obs_flat": obs.cpu()}, [], None)
        assert max_err_scaled(logits.cpu(), want) < 2e-5
    finally:
        tr.model.stop_rollout_server()
    eng = tr.engine
    eng.set_direct(True)
    sp = make_step_params(lr=5e-4, a_rec=1.0, kl=1.0, s_rec=0.0, cyc=1e-3, global_rows=256)
    assert not eng.direct_active(_lib.PHASE_JOINT, 256, sp, fused=True)


@pytest.mark.parametrize("te_in,md_in", [(("body",), ("task",)), (("task",), ("body",)), (("body", "task"), ("body",)),
                                         (("task",), ("body", "task"))])
def test_direct_first_layers_decline_every_subset(te_in, md_in):
    """pvae_set_direct with ANY input subset on either stack keeps the staged panels (they carry the structural zeros): also
    when one stack reads only the body and the other only the task block -- PVAE_INPUT_BODY | PVAE_INPUT_TASK of the two
    fields together spells "both", which a guard on their OR let through (advisor, round 5)."""
    wide = R.make_arch(197, 45, latent=32, te=(512, 2), md=(512, 2), wm=(512, 2))     # (first layers on the 32x32 tile kernels)
    arch = R.with_inputs(wide, te_in, md_in)
    data = R.synth_demo(0, 2, 300, 197, 45, kind="dynamics")
    tr = make_trainer(arch, data, 256, device=DEV)
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    eng.set_direct(True)
    sp = make_step_params(lr=5e-4, a_rec=1.0, kl=1.0, s_rec=0.0, cyc=1e-3, global_rows=256)
    assert not eng.direct_active(_lib.PHASE_JOINT, 256, sp, fused=True)
    full = make_trainer(wide, data, 256, device=DEV)
    full.engine.bind_dataset(*full.train_loader.dataset.device_arrays(full.engine.device))
    full.engine.set_direct(True)
    assert full.engine.direct_active(_lib.PHASE_JOINT, 256, sp, fused=True)        # (the guard is about subsets, nothing else)


def test_forward_decoder_between_training_steps_keeps_the_zero_block_exact():
    """motor_decoder_inputs = ["body"]: the z columns of the decoder's input panel must stay zero for its layer-0 weight
    gradient (the sampler writes z to a side panel).  `forward_decoder` -- pvae_net_forward(PVAE_NET_MD) on a caller's
    [s | z] rows -- copies only the WINDOW of those rows into the panel, so a rollout-style call between two optimizer steps
    leaves nothing behind: gradients, parameters and both moments outside the window stay exactly zero."""
    arch = R.with_inputs(R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 2)), ("body", "task"), ("body",))
    data = R.synth_demo(0, 2, 80, 23, 7, kind="dynamics")
    tr = make_trainer(arch, data, 32, m_world=1, device=DEV, eps_fn=R.eps_stream(2, 8))
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    eng = tr.engine
    tr.train()                                     # world epoch
    z_body = torch.randn(32, 23, device=DEV)
    z_task = torch.randn(32, 8, device=DEV) * 3.0
    for _ in range(2):                             # joint epochs with rollout-style decoder calls in between
        with torch.no_grad():
            a, _ = tr.model.forward_decoder(z_body, z_task)
            b, _ = tr.model.forward_decoder(z_body, z_task * 0.0)
        assert torch.equal(a, b)                   # a ["body"] decoder does not see z at all
        tr.train()
        assert _outside_window(eng, eng.grads) == 0.0
        assert _outside_window(eng, eng.params) == 0.0
        assert _outside_window(eng, eng.exp_avg) == 0.0 and _outside_window(eng, eng.exp_avg_sq) == 0.0
