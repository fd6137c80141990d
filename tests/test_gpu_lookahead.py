"""GPU parity of the multi-step unroll (`lookahead` > 1, tpv:367-428): the HIP path through the
C ABI against the oracle and against captures of the reference itself (look*_*.npz,
train_tiny_look2.npz).  Same tolerances as test_gpu_parity.py."""
import numpy as np
import pytest
import torch

from oracle import refpath as R
from physicsvae_amd import _lib
from physicsvae_amd.engine import make_step_params
from util import arch_from_meta, make_trainer, max_err_scaled, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _loss_of(name):
    return "L1" if name.startswith("l1_") else "MSE"


def _params(world, rows, **kw):
    c = R.phase_coeffs(world)
    return make_step_params(lr=5e-4, a_rec=c["a_rec_coeff"], kl=c["vae_kl_coeff"], s_rec=c["s_rec_coeff"],
                            cyc=c["vae_cycle_coeff"], global_rows=rows, **kw)


def _setup(golden, name):
    g = golden(name)
    arch = arch_from_meta(g)
    n_ep, n_steps, batch, L = [int(v) for v in g["meta"][9:13]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    x, y = next(iter(R.make_loader(X, Y, batch)))
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    es = R.eps_stream(2, arch["Z"])
    eps = torch.stack([es(t, (x.shape[0], arch["Z"])) for t in range(L)])
    tr = make_trainer(arch, data, batch, device=DEV, extra={"lookahead": L, "loss": _loss_of(name)})
    tr.model.load_state_dict(sd)
    assert tr.engine.lookahead == L and len(tr.train_loader.dataset) == int(g["n_windows"])
    return g, arch, data, x, y, sd, eps, tr, L


@pytest.mark.parametrize("name", ["look3_tiny", "look2_c1", "l1_tiny", "l1_look2_c1", "look2_mixed_tiny"])
@pytest.mark.parametrize("world", [True, False])
def test_unrolled_batch_matches_oracle_and_golden(golden, name, world):
    """l1_*: trainer key "loss" = "L1" (tm:100-101), lookahead 1 and 2."""
    g, arch, data, x, y, sd, eps, tr, L = _setup(golden, name)
    lk = _loss_of(name)
    eng = tr.engine
    tag = "world" if world else "joint"
    phase = _lib.PHASE_WORLD if world else _lib.PHASE_JOINT
    rows = x.shape[0]
    # (1) the full minibatch against the reference capture
    eng.set_batch(x, y)
    loss = eng.forward_backward(phase, rows, _params(world, rows, loss=lk), eps=eps, fused_adam=False).cpu()
    assert float(loss[0]) == pytest.approx(float(g[tag + "_total"]), rel=1e-5)
    gv = eng.named_views(eng.grads)
    for k in g[tag + "_grad_keys"]:
        np.testing.assert_allclose(R.tensor_digest(gv[str(k)].cpu())[:3], g["%s_graddigest::%s" % (tag, k)][:3],
                                   rtol=5e-3, atol=1e-7)
    # (2) tight against the oracle, without the samples that sit on a ReLU kink in any step
    keep = R.relu_kink_margin(arch, sd, x, y, eps, world) > 4e-6
    assert int(keep.sum()) >= rows - max(4, rows // 4)
    x, y, eps = x[keep], y[keep], eps[:, keep]
    rows = x.shape[0]
    want = R.loss_and_grads(arch, sd, x, y, eps, world, loss=lk)
    eng.set_batch(x, y)
    eng.grads.fill_(float("nan"))
    loss = eng.forward_backward(phase, rows, _params(world, rows, loss=lk), eps=eps, fused_adam=False).cpu()
    assert float(loss[0]) == pytest.approx(float(want["total"]), rel=1e-5)
    for i, k in enumerate(("loss_a", "loss_kl", "loss_s", "loss_cyc")):
        assert float(loss[1 + i]) == pytest.approx(float(want[k]), rel=1e-5, abs=1e-9), k
    # forward internals of EVERY step (the state of step t+1 is the prediction of step t)
    if L == 1 and world:                           # only the world model runs (pvae.h)
        assert max_err_scaled(eng.read("s2_hat", rows).cpu(), want["s2_from_gt_action"]) < 3e-5
    for t in range(L if not (L == 1 and world) else 0):
        st = want["steps"][t]
        for ours, theirs in (("mu", "mu"), ("logvar", "logvar"), ("z", "z"), ("a_hat", "a_hat"),
                             ("s2_hat", "future_state")):
            assert max_err_scaled(eng.read(ours, rows, step=t).cpu(), st[theirs]) < 3e-5, (t, ours)
        assert torch.equal(eng.read("eps", rows, step=t).cpu(), eps[t])
    gv = eng.named_views(eng.grads)
    assert list(want["grads"].keys()) == list(g[tag + "_grad_keys"])
    for k, gr in want["grads"].items():
        ours = gv[k].cpu()
        assert torch.isfinite(ours).all(), k
        assert max_err_scaled(ours, gr) < 1e-4, k
        assert rel_err(ours, gr) < 1e-4, k
    nets = [_lib.NET_WM] if world else [_lib.NET_TE, _lib.NET_MD]
    seg = eng.segment(eng.grads, nets)
    real = sum(gv[k].abs().double().sum().item() for k in want["grads"])
    assert seg.abs().double().sum().item() == pytest.approx(real, rel=1e-9)


def test_unrolled_gather_equals_explicit_batch_and_eval(golden):
    """Dataset-resident windows of L steps (first, middle, ragged last minibatch) == set_batch of the
    loader's [B, L, .] tensors, bit for bit; forward-only evaluation gives the same loss as the
    training call."""
    g, arch, data, x, y, sd, eps, tr, L = _setup(golden, "look3_tiny")
    eng = tr.engine
    eng.bind_dataset(*tr.train_loader.dataset.device_arrays(eng.device))
    X, Y = R.build_windows(data, lookahead=L)
    np.testing.assert_array_equal(tr.train_loader.dataset.X, X.astype(np.float32).astype(np.float64))
    batches = list(R.make_loader(X, Y, tr.train_loader.batch_size))
    spans = list(tr.train_loader.spans())
    assert len(batches) == len(spans) == int(g["n_batches"]) and spans[-1][1] == int(g["last_batch_size"])
    hb = list(tr.train_loader)[-1]
    assert torch.equal(hb[0], batches[-1][0]) and torch.equal(hb[1], batches[-1][1])
    es = R.eps_stream(5, arch["Z"])
    for b in (0, len(batches) // 2, len(batches) - 1):
        xb, yb = batches[b]
        first, rows = spans[b]
        e = torch.stack([es(b * L + t, (rows, arch["Z"])) for t in range(L)])
        for phase, world in ((_lib.PHASE_WORLD, True), (_lib.PHASE_JOINT, False)):
            sp = _params(world, rows)
            eng.gather(first, rows)
            l1 = eng.forward_backward(phase, rows, sp, eps=e, backward=False).clone()
            eng.gather(first, rows)
            l1b = eng.forward_backward(phase, rows, sp, eps=e, fused_adam=False).clone()
            g1 = eng.grads.clone()
            eng.set_batch(xb, yb)
            l2 = eng.forward_backward(phase, rows, sp, eps=e, fused_adam=False).clone()
            assert torch.equal(l1, l2) and torch.equal(l1b, l2) and torch.equal(g1, eng.grads)
            want = R.loss_and_grads(arch, sd, xb, yb, e, world)
            assert float(l1[0]) == pytest.approx(float(want["total"]), rel=1e-5)


def test_unrolled_fused_adam_and_shards(golden):
    """(a) Adam fused in the stacked weight-gradient launch == gradients + separate Adam, bit for
    bit; (b) two uneven row shards scaled by 1/global_rows sum to the full-batch gradient (what the
    N-GPU all-reduce relies on); (c) the staged backward used for data parallelism produces the same
    gradients as the one-call path."""
    g, arch, data, x, y, sd, eps, tr, L = _setup(golden, "look2_c1")
    eng = tr.engine
    rows = x.shape[0]
    for phase, world, nets in ((_lib.PHASE_WORLD, True, [_lib.NET_WM]),
                               (_lib.PHASE_JOINT, False, [_lib.NET_TE, _lib.NET_MD])):
        sp = _params(world, rows, adam_t=(3, 3, 3))
        tr.model.load_state_dict(sd)
        eng.exp_avg.normal_(0, 1e-3)
        eng.exp_avg_sq.uniform_(1e-6, 1e-4)
        p0, m0, v0 = eng.params.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone()
        eng.set_batch(x, y)
        full_loss = eng.forward_backward(phase, rows, sp, eps=eps, fused_adam=False).clone()
        full = eng.segment(eng.grads, nets).clone()
        eng.adam(nets, sp)
        p_sep, m_sep = eng.params.clone(), eng.exp_avg.clone()
        eng.params.copy_(p0)                       # whole arena, pad entries included
        eng.exp_avg.copy_(m0)
        eng.exp_avg_sq.copy_(v0)
        eng.set_batch(x, y)
        eng.forward_backward(phase, rows, sp, eps=eps, fused_adam=True)
        assert torch.equal(eng.params, p_sep) and torch.equal(eng.exp_avg, m_sep)
        # shards
        eng.params.copy_(p0)
        acc, acc_loss = torch.zeros_like(full), torch.zeros_like(full_loss)
        cut = rows // 2 + 3
        for lo, hi in ((0, cut), (cut, rows)):
            eng.set_batch(x[lo:hi], y[lo:hi])
            acc_loss += eng.forward_backward(phase, hi - lo, sp, eps=eps[:, lo:hi], fused_adam=False)
            acc += eng.segment(eng.grads, nets)
        assert max_err_scaled(acc.cpu(), full.cpu()) < 1e-5
        assert torch.allclose(acc_loss.cpu(), full_loss.cpu(), rtol=1e-5, atol=1e-8)
        # staged backward
        eng.set_batch(x, y)
        eng.grads.zero_()
        eng.forward_seed(phase, rows, sp, eps=eps)
        out = torch.zeros(5, device=DEV)
        k, n, covered = 0, 1, 0
        while k < n:
            seg, net, n = eng.backward_stage(phase, rows, sp, k, loss_out=out)
            if seg is not None:
                covered += seg[1]
                assert net in nets
            k += 1
        assert covered == sum(eng.segments[q][1] for q in nets)
        assert torch.equal(eng.segment(eng.grads, nets), full) and torch.equal(out, full_loss)


def test_unrolled_training_run_matches_reference_capture(golden):
    g = golden("train_tiny_look2")
    arch = arch_from_meta(g)
    n_ep, n_steps, batch, m_world, n_epochs, lr_step, L = [int(v) for v in g["meta"][9:16]]
    data = R.synth_demo(0, n_ep, n_steps, arch["Db"], arch["Da"], kind="dynamics")
    tr = make_trainer(arch, data, batch, m_world=m_world, device=DEV, lr_step=lr_step,
                      eps_fn=R.eps_stream(2, arch["Z"]), extra={"lookahead": L})
    tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
    losses = []
    for e in range(n_epochs):
        assert tr.optimizer.lr == pytest.approx(float(g["epoch_lrs"][e]), rel=1e-12)
        losses.append(tr.train()["mean_train_loss"])
        tag = "after_epoch%d" % (e + 1)
        for k, v in tr.model.state_dict().items():
            full = "%s::%s" % (tag, k)
            if full in g.files:
                assert max_err_scaled(v.cpu(), g[full]) < 2e-3, (tag, k)
    np.testing.assert_allclose(losses, g["epoch_losses"], rtol=1e-3)
    assert tr.global_batch * L == int(g["eps_calls"])
    nb = len(tr.train_loader)
    assert tr.optimizer.net_steps[_lib.NET_WM] == nb * m_world
    assert tr.optimizer.net_steps[_lib.NET_TE] == nb * (n_epochs - m_world)


def test_unrolled_full_size_step_tracks_oracle():
    """B=256, 4x1024, L=2 (the benchmark sizes): a few optimizer steps in each phase follow the
    oracle loop (same eps), and the on-chip Philox path runs."""
    arch = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    L = 2
    data = R.synth_demo(0, 2, 300, 197, 45, kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    es = R.eps_stream(2, 32)
    ref = R.RefTrainer(arch, sd, X, Y, 256, max_iter_world_model=1, eps_fn=es)
    tr = make_trainer(arch, data, 256, m_world=1, device=DEV, eps_fn=es, extra={"lookahead": L})
    tr.model.load_state_dict(sd)
    for epoch in range(2):
        want = ref.step()["mean_train_loss"]
        got = tr.train()["mean_train_loss"]
        assert got == pytest.approx(want, rel=1e-3), epoch
    tr.eps_fn = None                               # Philox draws, L consecutive offsets per step
    res = tr.train()
    assert np.isfinite(res["mean_train_loss"])


@pytest.mark.parametrize("act", ["tanh", "sigmoid", "elu"])
def test_unrolled_training_with_other_activations_tracks_oracle(act):
    """The trainer's "act_fn" through the multi-step unroll (every launch of the unrolled plan carries the
    activation and its derivative): a lookahead-2 run with Adam weight decay crossing the phase switch follows
    the oracle's loop epoch by epoch."""
    arch = R.make_arch(23, 7, latent=8, te=(64, 2), md=(96, 2), wm=(128, 3), act=act)
    L = 2
    data = R.synth_demo(0, 3, 80, 23, 7, kind="dynamics")
    X, Y = R.build_windows(data, lookahead=L)
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    es = R.eps_stream(2, 8)
    ref = R.RefTrainer(arch, sd, X, Y, 32, max_iter_world_model=2, eps_fn=es, weight_decay=0.01)
    tr = make_trainer(arch, data, 32, m_world=2, device=DEV, eps_fn=es, extra={"lookahead": L, "weight_decay": 0.01})
    tr.model.load_state_dict(sd)
    for epoch in range(4):
        want = ref.step()["mean_train_loss"]
        got = tr.train()["mean_train_loss"]
        assert got == pytest.approx(want, rel=1e-3), (act, epoch)
    for k, v in ref.model.state_dict().items():
        if not k.startswith("_value_branch"):
            assert rel_err(tr.model.state_dict()[k].cpu(), v) < (0.1 if k.endswith("bias") else 2e-2), k
