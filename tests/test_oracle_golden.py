"""Pin oracle/refpath.py (our CPU restatement) to outputs captured from the reference
itself (tests/golden/*.npz, produced by oracle/gen_golden.py in the dev container).
Runs anywhere, no GPU, no /root/reference."""
import numpy as np
import pytest
import torch

from oracle import refpath as R


def arch_from_meta(meta):
    """`meta` of a fixture, or the fixture itself (then an edited "act_fn" recorded in it is honoured)."""
    act, stacks = "relu", None
    if hasattr(meta, "files"):
        act = str(meta["act_fn"]) if "act_fn" in meta.files else "relu"
        stacks = meta["stacks"] if "stacks" in meta.files else None          # stacks recorded layer by layer
        meta = meta["meta"]
    Db, Da, Z, tw, td, mw, md_, ww, wd = [int(v) for v in meta[:9]]
    arch = R.make_arch(Db, Da, latent=Z, te=(tw, td), md=(mw, md_), wm=(ww, wd), act=act)
    return R.with_stacks(arch, stacks) if stacks is not None else arch


SINGLE = ["single_tiny", "single_c1", "single_c2", "single_default",
          "single_tiny_tanh", "single_tiny_sigmoid", "single_tiny_elu", "single_c1_tanh",      # these four: "act_fn" edited
          "single_mixed_tiny", "single_mixed_c1"]      # per-layer widths / activations (FC's general layer list, rmt:234-270)


@pytest.mark.parametrize("name", SINGLE)
def test_state_dict_layout_matches_reference(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    spec = R.state_dict_spec(arch)
    assert [k for k, _ in spec] == list(g["sd_keys"])
    shapes = [list(s) + [0] * (2 - len(s)) for _, s in spec]
    assert shapes == g["sd_shapes"].tolist()
    assert sum(int(np.prod(s)) for _, s in spec) == int(g["n_params"])
    # normc property of the reference's own init: hidden rows norm 1, output rows 0.01
    dims = R.net_layer_dims(arch)
    expect = [0.01 if i == len(d) - 1 else 1.0 for d in dims.values() for i in range(len(d))]
    np.testing.assert_allclose(g["init_row_norms_minmax"][:, 0], expect, rtol=1e-5)
    np.testing.assert_allclose(g["init_row_norms_minmax"][:, 1], expect, rtol=1e-5)
    sd = R.init_state_dict(arch, seed=1)
    for k, v in sd.items():
        if k.endswith("weight"):
            i = int(k.split(".")[2])
            net = k.split(".")[0]
            std = 0.01 if i == len(dims[net]) - 1 else 1.0
            np.testing.assert_allclose(v.norm(dim=1).numpy(), std, rtol=1e-5)


def test_tensor_counts_26_24_36(golden):
    assert len(golden("single_default")["sd_keys"]) == 26
    assert len(golden("single_c1")["sd_keys"]) == 24
    assert len(golden("single_c2")["sd_keys"]) == 36


@pytest.mark.parametrize("name", SINGLE)
def test_windows_and_loader_match_reference(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    assert X.dtype == np.float64 and X.shape == (n_ep * (n_steps - 1), 1, 2 * arch["Db"])
    assert len(X) == int(g["n_windows"])
    loader = R.make_loader(X, Y, batch)
    assert len(loader) == int(g["n_batches"])
    assert str(g["sampler"]) == "SequentialSampler"
    batches = list(loader)
    assert batches[-1][0].shape[0] == int(g["last_batch_size"])
    for tag, b in (("first", 0), ("mid", len(batches) // 2), ("last", len(batches) - 1)):
        xb, yb = batches[b]
        np.testing.assert_array_equal(R.tensor_digest(xb), g["loader_%s_x_digest" % tag])
        np.testing.assert_array_equal(R.tensor_digest(yb), g["loader_%s_y_digest" % tag])
        if "loader_%s_x" % tag in g:
            np.testing.assert_array_equal(xb.numpy(), g["loader_%s_x" % tag])
            np.testing.assert_array_equal(yb.numpy(), g["loader_%s_y" % tag])


def test_num_samples_cap_is_exact():
    data = R.synth_demo(0, 3, 20, 5, 2)
    X, _ = R.build_windows(data, num_samples=25)
    assert len(X) == 25
    X2, _ = R.build_windows(data)
    np.testing.assert_array_equal(X, X2[:25])


@pytest.mark.parametrize("name", SINGLE)
def test_single_batch_losses_and_grads_match_reference(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    eps = R.eps_stream(2, arch["Z"])(0, (x.shape[0], arch["Z"]))
    for world in (True, False):
        tag = "world" if world else "joint"
        out = R.loss_and_grads(arch, sd, x, y, eps, world)
        # same ops, same order, same machine class -> essentially exact
        np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
        for k in ("mu", "logvar", "z", "future_state"):
            if tag + "_" + k in g:
                np.testing.assert_allclose(out[k].numpy(), g[tag + "_" + k], rtol=1e-5, atol=1e-6)
            else:
                np.testing.assert_allclose(R.tensor_digest(out[k]), g["%s_%s_digest" % (tag, k)],
                                           rtol=1e-5, atol=1e-6)
        assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
        for k, gr in out["grads"].items():
            ref = g["%s_graddigest::%s" % (tag, k)]
            np.testing.assert_allclose(R.tensor_digest(gr), ref, rtol=2e-4, atol=1e-9)
            full = "%s_grad::%s" % (tag, k)
            if full in g:
                np.testing.assert_allclose(gr.numpy(), g[full], rtol=1e-4, atol=1e-8)
        # loss decomposition
        if world:
            np.testing.assert_allclose(out["loss_s"].numpy(), g["world_total"], rtol=1e-6)
        else:
            tot = out["loss_a"] + out["loss_kl"] + 1e-3 * out["loss_cyc"]
            np.testing.assert_allclose(tot.numpy(), g["joint_total"], rtol=1e-6)
    # value branch does not influence the loss (reference capture with VB weights + 1)
    np.testing.assert_array_equal(g["joint_total_vb_perturbed"], g["joint_total"])


CLEAR = ["single_tiny_clear", "single_c1_clear", "single_default_clear", "single_c2_clear", "single_pyramid_c1_clear"]


def clear_batch(g):
    """Inputs of a `*_clear` fixture: the demo's windows `rows_idx` (all off the ReLU kinks, oracle/gen_golden.py
    case_single_clear) with the draws of those rows."""
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="iid")
    X, Y = R.build_windows(data)
    idx = torch.from_numpy(g["rows_idx"])
    assert len(X) == int(g["n_windows"]) and idx.numel() == batch
    x = torch.from_numpy(np.asarray(X)).float()[idx]
    y = torch.from_numpy(np.asarray(Y)).float()[idx]
    eps = R.eps_stream(2, arch["Z"])(0, (len(X), arch["Z"]))[idx]
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    return arch, data, x, y, eps, sd


def digest_close(got, ref, vmax, tol):
    """A tensor's fingerprint [sum, abs-sum, L2, 16 samples] against the captured one: abs-sum and L2 relative, the
    (cancelling) sum against abs-sum, the samples against the tensor's largest magnitude."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert abs(got[1] - ref[1]) <= tol * ref[1] + 1e-30 and abs(got[2] - ref[2]) <= tol * ref[2] + 1e-30, (got[:3], ref[:3])
    assert abs(got[0] - ref[0]) <= tol * ref[1] + 1e-30, (got[0], ref[0], ref[1])
    assert np.abs(got[3:] - ref[3:]).max() <= tol * vmax + 1e-30, (np.abs(got[3:] - ref[3:]).max(), vmax)


@pytest.mark.parametrize("name", CLEAR)
def test_kink_free_batches_match_reference(golden, name):
    """The minibatches the HIP gradients are held to tightly (tests/test_gpu_parity.py): rows off every ReLU kink, so
    any two fp32 implementations agree to summation noise.  The restatement reproduces the reference's losses,
    internals and every gradient tensor's digest; the recorded rows really clear the margin."""
    g = golden(name)
    arch, data, x, y, eps, sd = clear_batch(g)
    for world in (True, False):
        tag = "world" if world else "joint"
        assert float(R.relu_kink_margin(arch, sd, x, y, eps, world).min()) > float(g["margin"])
        out = R.loss_and_grads(arch, sd, x, y, eps, world)
        np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
        for k in ("mu", "logvar", "z", "future_state"):
            digest_close(R.tensor_digest(out[k]), g["%s_%s_digest" % (tag, k)], float(g["%s_%s_max" % (tag, k)]), 2e-6)
        assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
        for k, gr in out["grads"].items():
            digest_close(R.tensor_digest(gr), g["%s_graddigest::%s" % (tag, k)], float(g["%s_gradmax::%s" % (tag, k)]), 2e-5)


def trained_batch(g):
    """Inputs of the `trained_*` fixture (oracle/gen_golden.py case_trained): the weights the REFERENCE's trainer ended
    with (TE / MD / WM as stored; the value branch never receives a gradient and is regenerated from its seed), the
    kink-free windows `rows_idx` of the learnable demo and their draws."""
    arch = arch_from_meta(g)
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data)
    idx = torch.from_numpy(g["rows_idx"])
    assert len(X) == int(g["n_windows"]) and idx.numel() == batch
    x = torch.from_numpy(np.asarray(X)).float()[idx]
    y = torch.from_numpy(np.asarray(Y)).float()[idx]
    eps = R.eps_stream(5, arch["Z"])(0, (len(X), arch["Z"]))[idx]
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    for k in sd:
        if not k.startswith("_value_branch"):
            sd[k] = torch.from_numpy(g["trained::" + k]).clone()
    return arch, data, x, y, eps, sd


def test_trained_weights_capture_is_where_the_trajectory_lives(golden):
    """What `trained_c1` holds is not initialisation statistics: the world model fitted (epoch loss down 5x), the
    posterior collapsed (KL ~ 5e-5, |mu|, |logvar| < 0.05), the learning rate decayed seven times, output layers moved
    by more than their initial size (decoder 17x, world model 30x), and the per-net Adam counters show the phase machine (WM stopped at the switch, TE / MD started there)."""
    g = golden("trained_c1")
    arch, data, x, y, eps, sd = trained_batch(g)
    m_world, n_epochs, lr_step = [int(v) for v in g["meta"][12:15]]
    losses = g["epoch_losses"]
    assert len(losses) == n_epochs and losses[m_world - 1] < 0.25 * losses[0] and losses[-1] < losses[m_world]
    assert float(g["final_lr"]) == pytest.approx(5e-4 * 0.7 ** (n_epochs // lr_step), rel=1e-12)
    assert 0.0 < float(g["joint_loss_kl"]) < 1e-3
    sd0 = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    for net in ("_task_encoder", "_motor_decoder", "_world_model"):
        k = [k for k in sd if k.startswith(net) and k.endswith("weight")][-1]          # the output layer
        assert float((sd[k] - sd0[k]).norm() / sd0[k].norm()) > 1.0, k          # moved by more than its own initial size
    out = R.loss_and_grads(arch, sd, x, y, eps, False)
    assert float(out["mu"].abs().max()) < 0.05 and float(out["logvar"].abs().max()) < 0.05     # the collapsed posterior
    steps_per_epoch = -(-int(g["n_windows"]) // int(g["meta"][11]))
    steps = dict(zip([str(k) for k in g["adam_keys"]], g["adam_steps"]))
    for k, v in steps.items():
        want = (m_world * steps_per_epoch if k.startswith("_world_model") else
                -1 if k.startswith("_value_branch") else (n_epochs - m_world) * steps_per_epoch)
        assert v == want, (k, v, want)


def test_oracle_matches_reference_at_trained_weights(golden):
    """The restatement against the reference's own compute_loss + backward at the weights the reference's trainer
    ended with: total loss, the KL term on its own, internals, every gradient tensor's digest, both phases."""
    g = golden("trained_c1")
    arch, data, x, y, eps, sd = trained_batch(g)
    for world in (True, False):
        tag = "world" if world else "joint"
        assert float(R.relu_kink_margin(arch, sd, x, y, eps, world).min()) > float(g["margin"])
        out = R.loss_and_grads(arch, sd, x, y, eps, world)
        np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
        if not world:
            # (1 + lv - mu^2 - exp(lv) at lv, mu ~ 1e-3 is a cancellation: the term is held to the ulp of the 1 it
            #  cancels against, summed over Z entries)
            assert float(out["loss_kl"]) == pytest.approx(float(g["joint_loss_kl"]), rel=1e-5, abs=arch["Z"] * 2.0 ** -24)
        for k in ("mu", "logvar", "z", "future_state"):
            digest_close(R.tensor_digest(out[k]), g["%s_%s_digest" % (tag, k)], float(g["%s_%s_max" % (tag, k)]), 2e-6)
        assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
        for k, gr in out["grads"].items():
            digest_close(R.tensor_digest(gr), g["%s_graddigest::%s" % (tag, k)], float(g["%s_gradmax::%s" % (tag, k)]), 2e-5)


@pytest.mark.parametrize("name", ["look3_tiny", "look2_c1", "l1_tiny", "l1_look2_c1", "look2_mixed_tiny"])
def test_lookahead_unroll_matches_reference(golden, name):
    """tpv:367-428 with lookahead 3 / 2: windows, ragged last batch, loss terms averaged over the
    steps, and the gradients of the back-propagation through every earlier step.  The l1_* cases
    were captured with trainer key "loss" = "L1" (nn.L1Loss for the three reconstruction terms)."""
    g = golden(name)
    loss = "L1" if name.startswith("l1_") else "MSE"
    arch = arch_from_meta(g)
    n_ep, n_steps, batch, L = [int(v) for v in g["meta"][9:13]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    assert X.shape == (n_ep * (n_steps - L), L, 2 * arch["Db"]) and len(X) == int(g["n_windows"])
    batches = list(R.make_loader(X, Y, batch))
    assert len(batches) == int(g["n_batches"]) and batches[-1][0].shape[0] == int(g["last_batch_size"])
    np.testing.assert_array_equal(R.tensor_digest(batches[-1][0]), g["loader_last_x_digest"])
    np.testing.assert_array_equal(R.tensor_digest(batches[-1][1]), g["loader_last_y_digest"])
    x, y = batches[0]
    assert list(x.shape) == g["x_shape"].tolist()
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    es = R.eps_stream(2, arch["Z"])
    eps = torch.stack([es(t, (x.shape[0], arch["Z"])) for t in range(L)])
    for world in (True, False):
        tag = "world" if world else "joint"
        out = R.loss_and_grads(arch, sd, x, y, eps if L > 1 else eps[0], world, loss=loss)
        np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
        for k in ("mu", "logvar", "z", "future_state"):           # internals of the LAST step
            if tag + "_" + k in g:
                np.testing.assert_allclose(out[k].numpy(), g[tag + "_" + k], rtol=1e-5, atol=1e-6)
            else:
                np.testing.assert_allclose(R.tensor_digest(out[k]), g["%s_%s_digest" % (tag, k)],
                                           rtol=1e-5, atol=1e-6)
        assert len(out["steps"]) == L
        assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
        for k, gr in out["grads"].items():
            np.testing.assert_allclose(R.tensor_digest(gr), g["%s_graddigest::%s" % (tag, k)],
                                       rtol=2e-4, atol=1e-9)
            full = "%s_grad::%s" % (tag, k)
            if full in g:
                np.testing.assert_allclose(gr.numpy(), g[full], rtol=1e-4, atol=1e-8)
    # world phase with L > 1 still trains only the world model, but its gradient now also
    # arrives through the (frozen) encoder/decoder of the earlier steps
    assert all(str(k).startswith("_world_model") for k in g["world_grad_keys"])


@pytest.mark.parametrize("name,arch", [("helper_tiny", R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))),
                                       ("helper_default", R.make_arch(197, 45))])
def test_motor_decoder_helper_restatement_matches_reference(golden, name, arch):
    """`motor_decoder_helper_enable` (rmt:490-498, 670-680, 833-835): layout and forward of the restated model against the
    reference's own model at the same weights, observations and draws -- bit for bit on CPU."""
    g = golden(name)
    h = R.with_helper(arch, rng=float(g["helper_range"]))
    assert [k for k, _ in R.state_dict_spec(h)] == list(g["sd_keys"])
    assert [list(s) + [0] * (2 - len(s)) for _, s in R.state_dict_spec(h)] == g["sd_shapes"].tolist()
    sd = R.perturb_biases(R.init_state_dict(h, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(h["mh"])
    sd[k_out] = sd[k_out] * 60.0
    m = R.RefModel(h)
    m.load_state_dict(sd)
    m.eval()
    obs, eps = torch.from_numpy(g["obs"]), torch.from_numpy(g["eps"])
    for noise, tag in ((False, "mean"), (True, "noise")):
        m.latent_prior_noise = noise
        m.eps_source = lambda shape: eps
        with torch.no_grad():
            logits = m(obs)
        assert torch.equal(logits, torch.from_numpy(g[tag + "_logits"]))
        assert torch.equal(m.cur_z, torch.from_numpy(g[tag + "_z"]))
        assert torch.equal(m.cur_future_state, torch.from_numpy(g[tag + "_future_state"]))
        assert torch.equal(m.cur_value, torch.from_numpy(g[tag + "_value"]))


def test_frozen_nets_get_no_grad(golden):
    g = golden("single_tiny")
    assert all(str(k).startswith("_world_model") for k in g["world_grad_keys"])
    assert all(str(k).startswith(("_task_encoder", "_motor_decoder")) for k in g["joint_grad_keys"])


def test_checkpoint_layout_matches_reference(golden):
    g = golden("single_default")
    arch = arch_from_meta(g)
    sd = R.init_state_dict(arch, 1)
    files = R.checkpoint_files(sd)
    assert sorted(files) == list(g["ckpt_files"])
    assert str(g["ckpt_return_basename"]) == "model.pth"
    assert list(g["ckpt_te_outer_keys"]) == ["task_encoder"]
    for f, obj in files.items():
        if f == "task_encoder.pt":
            obj = obj["task_encoder"]
        assert list(obj.keys()) == list(g["ckpt_keys::" + f]), f


@pytest.mark.parametrize("name", ["train_tiny", "train_c1", "train_tiny_look2", "train_tiny_elu_wd", "train_mixed_tiny"])
def test_training_run_matches_reference(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch, m_world, n_epochs, lr_step = [int(v) for v in g["meta"][9:15]]
    L = int(g["meta"][15]) if len(g["meta"]) > 15 else 1
    wd = float(g["weight_decay"]) if "weight_decay" in g.files else 0.0       # "weight_decay" edited (tpv:253)
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = R.RefTrainer(arch, sd, X, Y, batch, m_world, lr_step=lr_step,
                      eps_fn=R.eps_stream(2, arch["Z"]), weight_decay=wd)
    losses = []
    for e in range(n_epochs):
        lr = tr.opt.param_groups[0]["lr"]
        assert lr == pytest.approx(g["epoch_lrs"][e], rel=1e-12)
        assert lr == pytest.approx(R.lr_for_epoch(e + 1, step_size=lr_step), rel=1e-12)
        losses.append(tr.step()["mean_train_loss"])
        tag = "after_epoch%d" % (e + 1)
        if any(k.startswith(tag + "_digest::") for k in g.files):
            for k, v in tr.model.state_dict().items():
                np.testing.assert_allclose(R.tensor_digest(v), g["%s_digest::%s" % (tag, k)],
                                           rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(losses, g["epoch_losses"], rtol=1e-5)
    assert tr.eps_calls == int(g["eps_calls"]) == tr.global_batch * L
    # Adam bookkeeping (lazy state, WM stops at the switch, TE/MD start at t = 1, VB never)
    named = dict(tr.model.named_parameters())
    assert list(named.keys()) == list(g["adam_keys"])
    nb = len(tr.loader)
    for k, has, st in zip(g["adam_keys"], g["adam_has_state"], g["adam_steps"]):
        k = str(k)
        if k.startswith("_value_branch"):
            assert not has
        elif k.startswith("_world_model"):
            assert has and st == nb * m_world
        else:
            assert has and st == nb * (n_epochs - m_world)
        ours = tr.opt.state.get(named[k], {})
        assert (len(ours) > 0) == bool(has)
        if "adam_exp_avg::" + k in g:
            np.testing.assert_allclose(ours["exp_avg"].numpy(), g["adam_exp_avg::" + k],
                                       rtol=1e-3, atol=1e-9)
            np.testing.assert_allclose(ours["exp_avg_sq"].numpy(), g["adam_exp_avg_sq::" + k],
                                       rtol=1e-3, atol=1e-12)


def test_shuffled_training_run_matches_reference(golden):
    """The reference's trainer with `shuffle_data: True` (tm:166-175, 181) under torch.manual_seed: the restatement, whose
    loader is torch's own shuffled DataLoader, reproduces its epoch losses across the phase switch and its final weights."""
    g = golden("shuffle_tiny")
    arch = arch_from_meta(g)
    n_ep, n_steps, batch, m_world, n_epochs, seed = [int(v) for v in g["meta"][9:15]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data)
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    tr = R.RefTrainer(arch, sd, X, Y, batch, m_world, lr_step=2, eps_fn=R.eps_stream(2, arch["Z"]), shuffle=True)
    torch.manual_seed(seed)
    losses = [tr.step()["mean_train_loss"] for _ in range(n_epochs)]
    np.testing.assert_allclose(losses, g["epoch_losses"], rtol=1e-5)
    for k, v in tr.model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["final::" + k], rtol=2e-3, atol=2e-5)
    # not the sequential run: the order matters to every loss after the first step
    assert abs(losses[-1] - float(golden("train_tiny")["epoch_losses"][-1])) > 1e-4


def test_world_model_frozen_after_switch(golden):
    g = golden("train_tiny")
    m_world = int(g["meta"][12])
    n_epochs = int(g["meta"][13])
    for k in g.files:
        if k.startswith("after_epoch%d::_world_model" % m_world):
            tail = k.split("::")[1]
            np.testing.assert_array_equal(g[k], g["after_epoch%d::%s" % (n_epochs, tail)])


def test_adam_restatement_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(37, 11)
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref], lr=5e-4)
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    q = p.clone()
    for t in range(1, 6):
        gr = torch.randn(37, 11)
        ref.grad = gr.clone()
        opt.step()
        q, m, v = R.adam_reference_update(q, gr, m, v, t, 5e-4)
        np.testing.assert_allclose(q.numpy(), ref.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_known_answer_run_of_the_reference(golden):
    """SURVEY.md 8(c) anchor, captured from the reference run untouched (its own constructor under
    torch.manual_seed(0); sampler draws from torch's global generator): four epoch losses across the
    phase switch.  The oracle loop started from the same generator state reproduces them."""
    from util import make_trainer
    g = golden("anchor_c1")
    arch = arch_from_meta([197, 45, 32, 256, 2, 256, 2, 256, 2])
    data = R.survey_anchor_demo()
    torch.manual_seed(0)
    mine = make_trainer(arch, data, 64, m_world=2, device="cpu")      # consumes the init draws like the reference
    state = torch.get_rng_state()
    sd = {k: v.detach().clone() for k, v in mine.model.state_dict().items()}
    X, Y = R.build_windows(data)
    tr = R.RefTrainer(arch, sd, X, Y, 64, 2)                           # (its own nn.Linear init draws are discarded)
    torch.set_rng_state(state)
    losses = [tr.step()["mean_train_loss"] for _ in range(4)]
    np.testing.assert_allclose(losses[:2], g["world_epoch_losses"], rtol=1e-7)
    np.testing.assert_allclose(losses[2:], g["joint_epoch_losses"], rtol=1e-6)
    np.testing.assert_allclose(g["world_epoch_losses"], [1.0004073202989663, 0.9972580170175832], rtol=1e-12)


def subset_combos(g):
    return [tuple(tuple(part.split("+")) for part in str(c).split("/")) for c in g["combos"]]


def test_input_subsets_restatement_matches_reference(golden):
    """`task_encoder_inputs` / `motor_decoder_inputs` (rmt:470, 485; 607-613, 646-653, 776-783, 822-829): for every
    captured combination the restated model has the reference's state-dict layout (narrower first layers) and, at the
    same seeded weights, minibatch and sampler draws, the reference's total, code, prediction and gradients in both
    phases -- including which parameters receive a gradient at all (a decoder on ["body"] leaves the encoder with the
    KL term's gradient only)."""
    g = golden("subsets_tiny")
    base = arch_from_meta(g["meta"])
    n_ep, n_steps, batch = [int(v) for v in g["meta"][9:12]]
    data = R.synth_demo(0, n_ep, n_steps, base["Db"], base["Da"], kind="iid")
    X, Y = R.build_windows(data)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    for ci, (te_in, md_in) in enumerate(subset_combos(g)):
        arch = R.with_inputs(base, te_in, md_in)
        pre = "c%d_" % ci
        spec = R.state_dict_spec(arch)
        assert [k for k, _ in spec] == list(g[pre + "sd_keys"])
        assert [list(s) + [0] * (2 - len(s)) for _, s in spec] == g[pre + "sd_shapes"].tolist()
        sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        eps = torch.from_numpy(g[pre + "eps"])
        for world in (True, False):
            tag = pre + ("world" if world else "joint")
            out = R.loss_and_grads(arch, sd, x, y, eps, world)
            np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
            np.testing.assert_allclose(out["z"].numpy(), g[tag + "_z"], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose(out["future_state"].numpy(), g[tag + "_future_state"], rtol=1e-5, atol=1e-6)
            assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
            for k, gr in out["grads"].items():
                np.testing.assert_allclose(gr.numpy(), g["%s_grad::%s" % (tag, k)], rtol=1e-4, atol=1e-8)


@pytest.mark.parametrize("name", ["helper_train_tiny", "helper_train_look2_tiny"])
def test_helper_training_restatement_matches_reference(golden, name):
    """Supervised training of a helper model (rmt:670-680, 833-835): the reference's loss sees the helper's term in a_hat and
    its optimizer trains the helper with the decoder.  One minibatch in both phases and the reference's own five-epoch
    loop: epoch losses, final weights, Adam step counts.  lookahead 1: the helper has no gradient in the world phase and
    one in the joint phase.  lookahead 2 (tpv:367-428): the world phase reaches it too -- the state the world model
    continues from is its own prediction under the HELPED action (tpv:417-421) -- so its Adam counter runs from the first
    epoch while the decoder's starts at the switch."""
    g = golden(name)
    base = arch_from_meta(g["meta"])
    arch = R.with_helper(base, rng=float(g["helper_range"]))
    n_ep, n_steps, batch, m_world, n_epochs = [int(v) for v in g["meta"][9:14]]
    L = int(g["meta"][14]) if len(g["meta"]) > 14 else 1
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(arch["mh"])
    sd[k_out] = sd[k_out] * 60.0
    es = R.eps_stream(2, arch["Z"])
    eps = torch.from_numpy(g["eps"]) if L == 1 else torch.stack([es(t, (x.shape[0], arch["Z"])) for t in range(L)])
    for world in (True, False):
        tag = "world" if world else "joint"
        out = R.loss_and_grads(arch, sd, x, y, eps, world)
        np.testing.assert_allclose(out["total"].numpy(), g[tag + "_total"], rtol=1e-6)
        assert list(out["grads"].keys()) == list(g[tag + "_grad_keys"])
        assert any(k.startswith("_motor_decoder_helper") for k in out["grads"]) == (not world or L > 1)
        for k, gr in out["grads"].items():
            np.testing.assert_allclose(gr.numpy(), g["%s_grad::%s" % (tag, k)], rtol=1e-4, atol=1e-8)
    tr = R.RefTrainer(arch, sd, X, Y, batch, m_world, lr_step=2, eps_fn=R.eps_stream(2, arch["Z"]))
    losses = [tr.step()["mean_train_loss"] for _ in range(n_epochs)]
    np.testing.assert_allclose(losses, g["epoch_losses"], rtol=1e-5)
    for k, v in tr.model.state_dict().items():
        np.testing.assert_allclose(v.numpy(), g["final::" + k], rtol=2e-3, atol=2e-6, err_msg=k)
    nb = len(tr.loader)
    steps = dict(zip((str(k) for k in g["adam_keys"]), g["adam_steps"]))
    assert steps["_world_model._model.0._model.0.weight"] == nb * m_world
    assert steps["_motor_decoder_helper._model.0._model.0.weight"] == nb * (n_epochs if L > 1 else n_epochs - m_world)
    assert steps["_motor_decoder._model.0._model.0.weight"] == nb * (n_epochs - m_world)


def test_fixtures_regenerate_byte_for_byte():
    """The pin checks itself: the committed generator, run against the reference, writes the committed fixtures byte for
    byte (oracle/verify_golden.py).  Dev container only -- the GPU box has no reference, the test skips there.  The
    `*tiny*` fixtures (21 of 36, 20 s) always; all of them (5 min on 4 cores, profiles/r06_verify_golden.txt) with
    PVAE_VERIFY_GOLDEN_ALL=1."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import verify_golden as V
    if not os.path.isdir(V.REF):
        pytest.skip("no reference at %s" % V.REF)
    names = V.fixture_names()
    if os.environ.get("PVAE_VERIFY_GOLDEN_ALL") != "1":
        names = [n for n in names if "tiny" in n]
    assert len(names) >= 20
    bad = V.verify(names)
    assert not bad, bad
