"""oracle/gen_golden.py -- record golden vectors from the REFERENCE ITSELF.

Dev-container only: imports /root/reference's train_physics_vae / torch_models /
rllib_model_torch unmodified through the stand-in `ray`/`gym` packages in oracle/stubs
(SURVEY.md Appendix A), drives them on seeded synthetic inputs and writes small .npz
fixtures into tests/golden/.  Only *data* (inputs' seeds and the reference's outputs)
is committed; no reference source travels.  Run:  python oracle/gen_golden.py

Inputs are regenerated from seeds by oracle/refpath.py on both sides, so fixtures hold
outputs (plus the seeds/dims that define the inputs).
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("PVAE_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "stubs"), REF, ROOT]
if not hasattr(np, "product"):
    np.product = np.prod                      # numpy 2 dropped it; rmt:600-604 uses it

import torch  # noqa: E402

import train_physics_vae as T  # noqa: E402  (the reference)
from oracle import refpath as R  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def resolve_grid(cfg):
    for k, v in list(cfg.items()):
        if isinstance(v, dict) and "grid_search" in v:
            assert len(v["grid_search"]) == 1
            cfg[k] = v["grid_search"][0]
    return cfg


def make_reference_trainer(pkl, arch, batch, m_world, num_data=None, lookahead=1, loss="MSE", prior=None,
                           weight_decay=None, shuffle=None):
    argv = ["--data_train", pkl, "--batch_size", str(batch), "--max_iter", "1000",
            "--max_iter_world_model", str(m_world), "--latent_dim", str(arch["Z"])]
    T.args = T.arg_parser().parse_args(argv)
    T.args.num_data = num_data
    cfg = resolve_grid(T.get_trainer_config(T.args))
    # Stacks with per-layer widths / activations: FC accepts any list of fc layers (rmt:234-270) but the trainer
    # always overwrites custom_model_config's *_layers with gen_layers(width, depth) (tpv:290-311), so for those the
    # helper is wrapped for the duration of the construction: a "width" that is a layer-by-layer spec yields that
    # layer list (same dict keys as gen_layers' rows).  Model, loss, optimizer and loop stay the reference's.
    for key, spec in (("TE", arch["te"]), ("MD", arch["md"]), ("world_model", arch["wm"])):
        cfg[key + "_width"], cfg[key + "_depth"] = (list(map(tuple, spec)), None) if general(spec) else spec
    cfg["lookahead"] = lookahead              # tpv:277 hard-wires 1; users edit the dict
    cfg["loss"] = loss                        # tpv:257 -> get_loss_fn (tm:97-107)
    if prior is not None:
        cfg["latent_prior_type"] = prior      # tpv:262 (a grid leaf upstream; users edit the dict)
    cfg["act_fn"] = arch.get("act", "relu")   # tpv:262 -> gen_layers(act_hidden=...) for every stack
    if weight_decay is not None:
        cfg["weight_decay"] = weight_decay    # tpv:253 -> torch.optim.Adam(weight_decay=...) (tm:119-122)
    if shuffle is not None:
        cfg["shuffle_data"] = shuffle         # tm:181 reads this key (tpv:260 sets "suffle_data": users fix the typo)
    orig = T.gen_layers

    def gen_layers(width, depth, **kw):
        return R.fc_layer_list(width) if isinstance(width, list) else orig(width=width, depth=depth, **kw)
    T.gen_layers = gen_layers
    try:
        return T.TrainModel(cfg)
    finally:
        T.gen_layers = orig


class EpsPatch:
    """Replace torch.randn_like by a deterministic per-call stream (one call per forward)."""

    def __init__(self, fn):
        self.fn, self.calls = fn, 0

    def __enter__(self):
        self.orig = torch.randn_like

        def patched(t, *a, **k):
            e = self.fn(self.calls, t.shape)
            self.calls += 1
            return e
        torch.randn_like = patched
        return self

    def __exit__(self, *exc):
        torch.randn_like = self.orig


def general(spec):
    """A stack given layer by layer -- [(width, activation), ...] -- instead of gen_layers' (width, depth)."""
    return len(spec) > 0 and isinstance(spec[0], (tuple, list))


def wd(spec):
    """(first width, depth) of a stack for the `meta` vector (general stacks travel in full as "stacks")."""
    return (spec[0][0], len(spec)) if general(spec) else tuple(spec)


def act_meta(arch):
    """Extra entries of fixtures captured with an edited "act_fn" or with stacks gen_layers cannot emit (files of the
    relu / uniform default stay as they were)."""
    act = arch.get("act", "relu")
    out = {"act_fn": np.array(act)} if act != "relu" else {}
    if any(general(arch[k]) for k in ("te", "md", "wm")):
        out["stacks"] = np.array(json.dumps({k: [list(l) for l in R.hidden_layers(arch[k], act)] for k in ("te", "md", "wm")}))
    return out


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def single_batch_capture(tr, x, y, eps, world):
    """Losses, internals and grads from the reference's own compute_loss + backward."""
    m = tr.model
    m.set_learnable_task_encoder(not world)
    m.set_learnable_motor_decoder(not world)
    m.set_learnable_world_model(world)
    tr.read_loss_fn_coeff(world=world)
    tr.model.train()
    tr.optimizer.zero_grad()
    with EpsPatch(lambda c, shape: eps if eps.dim() == 2 else eps[c]):
        loss = tr.compute_loss(y, x)
    loss.backward()
    out = {"total": loss.detach().numpy()}
    out["mu"] = m._cur_task_encoder_mu.detach().numpy()
    out["logvar"] = m._cur_task_encoder_logvar.detach().numpy()
    out["z"] = m._cur_task_encoder_variable.detach().numpy()
    out["future_state"] = m._cur_future_state.detach().numpy()
    return out, grads_of(m)


def case_single(name, arch, n_ep, n_steps, batch, full):
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"],
                        dim_action=arch["Da"], kind="iid")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=2)
        sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        # --- construction facts
        ref_sd = tr.model.state_dict()
        fix["sd_keys"] = np.array(list(ref_sd.keys()))
        fix["sd_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in ref_sd.values()])
        fix["n_params"] = np.array(sum(v.numel() for v in ref_sd.values()))
        for k, v in ref_sd.items():          # normc property of the reference's own init
            if k.endswith("weight"):
                fix.setdefault("init_row_norms_minmax", []).append(
                    [float(v.norm(dim=1).min()), float(v.norm(dim=1).max())])
        fix["init_row_norms_minmax"] = np.array(fix["init_row_norms_minmax"])
        tr.model.load_state_dict(sd)          # strict: proves our key layout drops in
        # --- loader facts
        loader = tr.train_loader
        fix["n_windows"] = np.array(len(loader.dataset))
        fix["n_batches"] = np.array(len(loader))
        fix["sampler"] = np.array(type(loader.sampler).__name__)
        batches = list(loader)
        fix["last_batch_size"] = np.array(batches[-1][0].shape[0])
        for tag, b in (("first", 0), ("mid", len(batches) // 2), ("last", len(batches) - 1)):
            xb, yb = batches[b]
            fix["loader_%s_x_digest" % tag] = R.tensor_digest(xb)
            fix["loader_%s_y_digest" % tag] = R.tensor_digest(yb)
            if full:
                fix["loader_%s_x" % tag] = xb.numpy()
                fix["loader_%s_y" % tag] = yb.numpy()
        # --- one minibatch, both phases
        x, y = batches[0]
        eps = R.eps_stream(2, arch["Z"])(0, (x.shape[0], arch["Z"]))
        for world in (True, False):
            tag = "world" if world else "joint"
            out, grads = single_batch_capture(tr, x, y, eps, world)
            for k, v in out.items():
                if full or v.ndim == 0:
                    fix["%s_%s" % (tag, k)] = v
                else:
                    fix["%s_%s_digest" % (tag, k)] = R.tensor_digest(torch.from_numpy(v))
            # per-term losses via the restated formulas are checked in tests; record terms too
            fix["%s_grad_keys" % tag] = np.array(list(grads.keys()))
            for k, g in grads.items():
                if full:
                    fix["%s_grad::%s" % (tag, k)] = g.numpy()
                fix["%s_graddigest::%s" % (tag, k)] = R.tensor_digest(g)
        # value branch / AppendLogStd do not influence the loss: perturb VB, loss unchanged
        sd2 = {k: (v + 1.0 if k.startswith("_value_branch") else v) for k, v in sd.items()}
        tr.model.load_state_dict(sd2)
        out2, _ = single_batch_capture(tr, x, y, eps, False)
        fix["joint_total_vb_perturbed"] = out2["total"]
        tr.model.load_state_dict(sd)
        # --- checkpoint layout
        ck = os.path.join(td, "ck")
        os.makedirs(ck)
        ret = tr.save_checkpoint(ck)
        fix["ckpt_return_basename"] = np.array(os.path.basename(ret))
        files = sorted(os.listdir(ck))
        fix["ckpt_files"] = np.array(files)
        for f in files:
            obj = torch.load(os.path.join(ck, f))
            if f == "task_encoder.pt":
                fix["ckpt_te_outer_keys"] = np.array(list(obj.keys()))
                obj = obj["task_encoder"]
            fix["ckpt_keys::" + f] = np.array(list(obj.keys()))
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]),
                            n_ep, n_steps, batch])
    fix.update(act_meta(arch))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "keys:", len(fix))


def case_single_clear(name, arch, n_ep, n_steps, batch, margin=1e-5):
    """One minibatch of the reference's compute_loss + backward on rows that are OFF every ReLU kink: the first
    `batch` windows of the demo whose smallest hidden pre-activation magnitude (both phases) exceeds `margin`
    (a unit within fp32 rounding of 0 flips a sample's gradient path in any two fp32 implementations, so captures
    that contain such rows can only be compared loosely).  With them left out the gradient of every tensor is
    reproducible to fp32 summation noise, and the HIP path is held to the capture at 1e-4 instead of 5e-3.
    Records the selected window indices, losses, digests of the internals and, per gradient tensor, digest + max."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=2)
        sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        tr.model.load_state_dict(sd)
        X, Y = R.build_windows(data)
        xa = torch.from_numpy(np.asarray(X)).float()
        ya = torch.from_numpy(np.asarray(Y)).float()
        n = xa.shape[0]
        eps_all = R.eps_stream(2, arch["Z"])(0, (n, arch["Z"]))
        m = torch.minimum(R.relu_kink_margin(arch, sd, xa, ya, eps_all, True),
                          R.relu_kink_margin(arch, sd, xa, ya, eps_all, False))
        idx = torch.nonzero(m > margin).reshape(-1)[:batch]
        assert idx.numel() == batch, "only %d of %d windows clear the margin" % (idx.numel(), n)
        x, y, eps = xa[idx], ya[idx], eps_all[idx]
        fix["rows_idx"] = idx.numpy().astype(np.int64)
        fix["n_windows"] = np.array(n)
        fix["margin"] = np.array(margin)
        fix["min_margin_of_batch"] = np.array(float(m[idx].min()))
        fix["rows_dropped_before_last"] = np.array(int(idx[-1]) + 1 - batch)
        for world in (True, False):
            tag = "world" if world else "joint"
            out, grads = single_batch_capture(tr, x, y, eps, world)
            for k, v in out.items():
                if v.ndim == 0:
                    fix["%s_%s" % (tag, k)] = v
                else:
                    fix["%s_%s_digest" % (tag, k)] = R.tensor_digest(torch.from_numpy(v))
                    fix["%s_%s_max" % (tag, k)] = np.array(float(np.abs(v).max()))
            fix["%s_grad_keys" % tag] = np.array(list(grads.keys()))
            for k, g in grads.items():
                fix["%s_graddigest::%s" % (tag, k)] = R.tensor_digest(g)
                fix["%s_gradmax::%s" % (tag, k)] = np.array(float(g.abs().max()))
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), n_ep, n_steps, batch])
    fix.update(act_meta(arch))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "keys:", len(fix), "| rows skipped:", int(fix["rows_dropped_before_last"]),
          "| min margin %.3g" % float(fix["min_margin_of_batch"]))


def case_trained(name, arch, n_ep, n_steps, batch, m_world, n_epochs, lr_step, margin=1e-5):
    """The reference's compute_loss + backward (tpv:361-435) at TRAINED weights: the reference trainer itself runs
    `m_world` world-model epochs and `n_epochs - m_world` joint epochs on learnable data (StepLR shortened to `lr_step`
    epochs so that the rate has decayed several times, eps keyed by forward call), then one kink-free minibatch is
    captured in both phases exactly as case_single_clear does at initialisation -- but where the trajectory lives:
    a fitted world model, a collapsed posterior (KL ~ 1e-4 and below), output layers grown from |row| = 0.01.
    Stores the trained TE / MD / WM tensors in full (fp32; the value branch never receives a gradient, tm:119-122 +
    torch's `grad is None` skip, and stays at its seeded initial value), the minibatch's window indices, losses,
    digests of the internals and per gradient tensor digest + max, plus the trainer's read-outs of the run."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"], dim_action=arch["Da"], kind="dynamics")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=m_world)
        tr.lr_scheduler = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=lr_step, gamma=0.7)
        sd0 = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        tr.model.load_state_dict(sd0)
        eps_fn = R.eps_stream(2, arch["Z"])
        losses = []
        with EpsPatch(lambda c, shape: eps_fn(c, shape)):
            for e in range(n_epochs):
                losses.append(tr.train()["mean_train_loss"])
        fix["epoch_losses"] = np.array(losses, dtype=np.float64)
        fix["final_lr"] = np.array(tr.optimizer.param_groups[0]["lr"], dtype=np.float64)
        sd = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
        for k, v in sd.items():
            if k.startswith("_value_branch"):
                assert torch.equal(v, sd0[k]), k                 # never trained: regenerated from the seed by the tests
            else:
                fix["trained::" + k] = v.numpy().copy()
        # Adam's per-parameter step counters at that point (WM stopped at the switch; TE / MD started at 1 there)
        named = dict(tr.model.named_parameters())
        fix["adam_keys"] = np.array(list(named.keys()))
        fix["adam_steps"] = np.array([float(tr.optimizer.state.get(p, {}).get("step", -1.0)) for p in named.values()])
        X, Y = R.build_windows(data)
        xa = torch.from_numpy(np.asarray(X)).float()
        ya = torch.from_numpy(np.asarray(Y)).float()
        n = xa.shape[0]
        eps_all = R.eps_stream(5, arch["Z"])(0, (n, arch["Z"]))
        m = torch.minimum(R.relu_kink_margin(arch, sd, xa, ya, eps_all, True),
                          R.relu_kink_margin(arch, sd, xa, ya, eps_all, False))
        idx = torch.nonzero(m > margin).reshape(-1)[:batch]
        assert idx.numel() == batch, "only %d of %d windows clear the margin" % (idx.numel(), n)
        x, y, eps = xa[idx], ya[idx], eps_all[idx]
        fix["rows_idx"] = idx.numpy().astype(np.int64)
        fix["n_windows"] = np.array(n)
        fix["margin"] = np.array(margin)
        fix["min_margin_of_batch"] = np.array(float(m[idx].min()))
        fix["rows_dropped_before_last"] = np.array(int(idx[-1]) + 1 - batch)
        for world in (True, False):
            tag = "world" if world else "joint"
            out, grads = single_batch_capture(tr, x, y, eps, world)
            for k, v in out.items():
                if v.ndim == 0:
                    fix["%s_%s" % (tag, k)] = v
                else:
                    fix["%s_%s_digest" % (tag, k)] = R.tensor_digest(torch.from_numpy(v))
                    fix["%s_%s_max" % (tag, k)] = np.array(float(np.abs(v).max()))
            # the KL term on its own, from the tensors the reference's model holds after compute_loss (tpv:385-389)
            m_ = tr.model
            if not world:
                mu, lv = m_._cur_task_encoder_mu.detach(), m_._cur_task_encoder_logvar.detach()
                fix["joint_loss_kl"] = torch.mean(-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp(), dim=1), dim=0).numpy()
            fix["%s_grad_keys" % tag] = np.array(list(grads.keys()))
            for k, g in grads.items():
                fix["%s_graddigest::%s" % (tag, k)] = R.tensor_digest(g)
                fix["%s_gradmax::%s" % (tag, k)] = np.array(float(g.abs().max()))
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), n_ep, n_steps, batch,
                            m_world, n_epochs, lr_step])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "| epoch losses: first %.4g, at switch %.4g, after switch %.4g, last %.4g | lr %.3g | KL %.3g"
          % (losses[0], losses[m_world - 1], losses[m_world], losses[-1], float(fix["final_lr"]), float(fix["joint_loss_kl"])),
          "| rows skipped:", int(fix["rows_dropped_before_last"]), "| min margin %.3g" % float(fix["min_margin_of_batch"]))


def case_lookahead(name, arch, n_ep, n_steps, batch, lookahead, full, loss="MSE"):
    """One minibatch through the reference's multi-step unroll (tpv:367-428), both phases."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"],
                        dim_action=arch["Da"], kind="dynamics")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=2, lookahead=lookahead, loss=loss)
        sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        tr.model.load_state_dict(sd)
        loader = tr.train_loader
        fix["n_windows"] = np.array(len(loader.dataset))
        fix["n_batches"] = np.array(len(loader))
        batches = list(loader)
        fix["last_batch_size"] = np.array(batches[-1][0].shape[0])
        fix["loader_last_x_digest"] = R.tensor_digest(batches[-1][0])
        fix["loader_last_y_digest"] = R.tensor_digest(batches[-1][1])
        x, y = batches[0]
        fix["x_shape"] = np.array(x.shape)
        es = R.eps_stream(2, arch["Z"])
        eps = torch.stack([es(t, (x.shape[0], arch["Z"])) for t in range(lookahead)])
        for world in (True, False):
            tag = "world" if world else "joint"
            out, grads = single_batch_capture(tr, x, y, eps, world)
            for k, v in out.items():
                if full or v.ndim == 0:
                    fix["%s_%s" % (tag, k)] = v
                else:
                    fix["%s_%s_digest" % (tag, k)] = R.tensor_digest(torch.from_numpy(v))
            fix["%s_grad_keys" % tag] = np.array(list(grads.keys()))
            for k, g in grads.items():
                if full:
                    fix["%s_grad::%s" % (tag, k)] = g.numpy()
                fix["%s_graddigest::%s" % (tag, k)] = R.tensor_digest(g)
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]),
                            n_ep, n_steps, batch, lookahead])
    fix.update(act_meta(arch))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "keys:", len(fix), "world", fix["world_total"], "joint", fix["joint_total"])


def case_training(name, arch, n_ep, n_steps, batch, m_world, n_epochs, full, lr_step=2, lookahead=1,
                  weight_decay=None):
    """Multi-epoch run crossing the phase switch, eps keyed by global minibatch index."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"],
                        dim_action=arch["Da"], kind="dynamics")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=m_world, lookahead=lookahead, weight_decay=weight_decay)
        # shorten StepLR so that the decay is exercised inside the captured run
        tr.lr_scheduler = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=lr_step, gamma=0.7)
        sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
        tr.model.load_state_dict(sd)
        eps_fn = R.eps_stream(2, arch["Z"])
        losses, lrs = [], []
        with EpsPatch(lambda c, shape: eps_fn(c, shape)) as ep:
            for e in range(n_epochs):
                lrs.append(tr.optimizer.param_groups[0]["lr"])
                losses.append(tr.train()["mean_train_loss"])
                if e + 1 in (1, m_world, n_epochs):
                    tag = "after_epoch%d" % (e + 1)
                    st = tr.model.state_dict()
                    for k, v in st.items():
                        if full:
                            fix["%s::%s" % (tag, k)] = v.detach().numpy().copy()
                        fix["%s_digest::%s" % (tag, k)] = R.tensor_digest(v)
            fix["eps_calls"] = np.array(ep.calls)
        fix["epoch_losses"] = np.array(losses, dtype=np.float64)
        fix["epoch_lrs"] = np.array(lrs, dtype=np.float64)
        # Adam bookkeeping: which params have state, and their step counts
        named = dict(tr.model.named_parameters())
        steps, have = [], []
        for k, p in named.items():
            st = tr.optimizer.state.get(p, {})
            have.append(len(st) > 0)
            steps.append(float(st["step"]) if len(st) else -1.0)
            if len(st) and (full or k.endswith("2._model.0.bias")):
                fix["adam_exp_avg::" + k] = st["exp_avg"].numpy().copy()
                fix["adam_exp_avg_sq::" + k] = st["exp_avg_sq"].numpy().copy()
        fix["adam_keys"] = np.array(list(named.keys()))
        fix["adam_has_state"] = np.array(have)
        fix["adam_steps"] = np.array(steps)
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]),
                            n_ep, n_steps, batch, m_world, n_epochs, lr_step, lookahead])
    fix.update(act_meta(arch))
    if weight_decay:
        fix["weight_decay"] = np.array(float(weight_decay))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "losses", losses, "lrs", lrs)


def case_shuffle(name, arch, n_ep, n_steps, batch, m_world, n_epochs, seed=1234):
    """`shuffle_data: True` (tm:166-175, 181): the reference's own DataLoader shuffles every pass with a RandomSampler
    seeded from torch's default generator.  Captured under torch.manual_seed(seed) set right before the first epoch:
    every epoch's sample order (the indices its dataset is asked for), the epoch losses across the phase switch and the
    final weights.  eps is the usual patched stream, which does not touch the default generator, so the loader's two
    draws per pass are the only ones."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"],
                        dim_action=arch["Da"], kind="dynamics")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=m_world, shuffle=True)
        assert type(tr.train_loader.sampler).__name__ == "RandomSampler"
        tr.lr_scheduler = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=2, gamma=0.7)
        tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3))
        eps_fn = R.eps_stream(2, arch["Z"])
        ds_cls = type(tr.train_loader.dataset)
        orig_get, asked = ds_cls.__getitem__, []

        def getitem(self, i):
            asked.append(int(i))
            return orig_get(self, i)
        ds_cls.__getitem__ = getitem
        losses = []
        try:
            torch.manual_seed(seed)
            with EpsPatch(lambda c, shape: eps_fn(c, shape)):
                for e in range(n_epochs):
                    del asked[:]
                    losses.append(tr.train()["mean_train_loss"])
                    fix["order_epoch%d" % e] = np.array(asked, dtype=np.int64)
        finally:
            ds_cls.__getitem__ = orig_get
        for k, v in tr.model.state_dict().items():
            fix["final::" + k] = v.detach().numpy().copy()
    fix["epoch_losses"] = np.array(losses, dtype=np.float64)
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]),
                            n_ep, n_steps, batch, m_world, n_epochs, seed])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "losses", losses, "first order", fix["order_epoch0"][:8])


def case_anchor(name):
    """The reference exactly as a user runs it: its OWN constructor under torch.manual_seed(0) (normc
    init through the torch RNG), B=64, 2x256 stacks, max_iter_world_model=2.  Records the initial
    state dict (digests) and the two world-phase epoch losses (which do not depend on the sampler's
    draws: 1.0004073202989663, 0.9972580170175832 -- the known-answer anchor of SURVEY.md 8c)."""
    arch = R.make_arch(197, 45, latent=32, te=(256, 2), md=(256, 2), wm=(256, 2))
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, R.survey_anchor_demo())
        torch.manual_seed(0)
        tr = make_reference_trainer(pkl, arch, 64, m_world=2)
        sd0 = tr.model.state_dict()
        fix["sd_keys"] = np.array(list(sd0.keys()))
        for k, v in sd0.items():
            fix["init_digest::" + k] = R.tensor_digest(v)
        fix["n_batches"] = np.array(len(tr.train_loader))
        fix["world_epoch_losses"] = np.array([tr.train()["mean_train_loss"] for _ in range(2)], dtype=np.float64)
        for k, v in tr.model.state_dict().items():
            if k.startswith("_world_model"):
                fix["after_world_digest::" + k] = R.tensor_digest(v)
        # two joint epochs with the sampler drawing from torch's global CPU generator, untouched:
        # per epoch one int64 draw by the DataLoader iterator (its base seed), then one
        # randn_like([B, Z]) per minibatch -- in the world epochs above as well (tpv:378)
        fix["joint_epoch_losses"] = np.array([tr.train()["mean_train_loss"] for _ in range(2)], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, fix["world_epoch_losses"], fix["joint_epoch_losses"])


def case_ingest(name, arch):
    """Dataset ingest variants of the reference (tpv:94-164): two --data_train files merged, and the
    --num_data cap; window counts, batch counts and digests of the loader's first / last batches."""
    d1 = R.synth_demo(seed=0, n_episodes=2, n_steps=14, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    d2 = R.synth_demo(seed=5, n_episodes=3, n_steps=11, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        p1, p2 = os.path.join(td, "a.pkl"), os.path.join(td, "b.pkl")
        R.write_demo(p1, d1)
        R.write_demo(p2, d2)
        for tag, num in (("all", None), ("cap", 37)):
            argv = ["--data_train", p1, "--data_train", p2, "--batch_size", "8", "--max_iter", "10",
                    "--max_iter_world_model", "2", "--latent_dim", str(arch["Z"])]
            T.args = T.arg_parser().parse_args(argv)
            T.args.num_data = num
            cfg = resolve_grid(T.get_trainer_config(T.args))
            cfg["TE_width"], cfg["TE_depth"] = arch["te"]
            cfg["MD_width"], cfg["MD_depth"] = arch["md"]
            cfg["world_model_width"], cfg["world_model_depth"] = arch["wm"]
            tr = T.TrainModel(cfg)
            batches = list(tr.train_loader)
            fix[tag + "_n_windows"] = np.array(len(tr.train_loader.dataset))
            fix[tag + "_n_batches"] = np.array(len(batches))
            fix[tag + "_last_batch_size"] = np.array(batches[-1][0].shape[0])
            for b, nm in ((0, "first"), (len(batches) - 1, "last")):
                fix["%s_%s_x_digest" % (tag, nm)] = R.tensor_digest(batches[b][0])
                fix["%s_%s_y_digest" % (tag, nm)] = R.tensor_digest(batches[b][1])
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"])])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, {k: int(v) for k, v in fix.items() if k.endswith(("n_windows", "n_batches", "last_batch_size"))})


def case_noprior(name, arch, n_ep, n_steps, batch):
    """`latent_prior_type = False` (rmt:622-623, 815-816): a mode the reference runs -- the encoder's Z outputs
    are the decoder's code, nothing is sampled, no KL term.  One minibatch, both phases: total, internals, every
    gradient, plus the state-dict layout (the encoder's last layer has Z rows, not 2Z)."""
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        tr = make_reference_trainer(pkl, arch, batch, m_world=2, prior=False)
        ref_sd = tr.model.state_dict()
        fix["sd_keys"] = np.array(list(ref_sd.keys()))
        fix["sd_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in ref_sd.values()])
        sd = R.perturb_biases(R.init_state_dict(dict(arch, prior=False), seed=1), seed=3)
        tr.model.load_state_dict(sd)
        x, y = next(iter(tr.train_loader))
        for world in (True, False):
            tag = "world" if world else "joint"
            m = tr.model
            m.set_learnable_task_encoder(not world)
            m.set_learnable_motor_decoder(not world)
            m.set_learnable_world_model(world)
            tr.read_loss_fn_coeff(world=world)
            m.train()
            tr.optimizer.zero_grad()
            loss = tr.compute_loss(y, x)
            loss.backward()
            fix[tag + "_total"] = loss.detach().numpy()
            fix[tag + "_z"] = m._cur_task_encoder_variable.detach().numpy()
            fix[tag + "_future_state"] = m._cur_future_state.detach().numpy()
            grads = grads_of(m)
            fix[tag + "_grad_keys"] = np.array(list(grads.keys()))
            for k, g in grads.items():
                fix["%s_grad::%s" % (tag, k)] = g.numpy()
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), n_ep, n_steps, batch])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "world", fix["world_total"], "joint", fix["joint_total"])


def case_ingest_rel(name, arch):
    """`load_dataset_for_PhysicsVAE(cond="rel")` of the reference itself (tpv:149-150: the second half of
    x is s_{t+1} - s_t; the trainer never asks for it, so it is called directly): digests of the whole X / Y
    and of the first and last windows as the reference's DatasetBase yields them, lookahead 1 and 2."""
    data = R.synth_demo(seed=3, n_episodes=3, n_steps=12, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid",
                        quantum=0.0)                      # un-rounded values: the float64 subtraction matters
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        p1 = os.path.join(td, "a.pkl")
        R.write_demo(p1, data)
        for L in (1, 2):
            ds = T.load_dataset_for_PhysicsVAE([p1], lookahead=L, cond="rel")
            fix["L%d_n_windows" % L] = np.array(len(ds))
            fix["L%d_X_digest" % L] = R.tensor_digest(torch.from_numpy(np.asarray(ds.X)))
            fix["L%d_Y_digest" % L] = R.tensor_digest(torch.from_numpy(np.asarray(ds.Y)))
            for i, nm in ((0, "first"), (len(ds) - 1, "last")):
                x, y = ds[i]
                fix["L%d_%s_x" % (L, nm)] = x.numpy()
                fix["L%d_%s_y" % (L, nm)] = y.numpy()
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"])])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, {k: int(v) for k, v in fix.items() if k.endswith("n_windows")})


def case_checkpoint_interop(name, arch):
    """Both directions of the checkpoint drop-in, with the REFERENCE's own classes:
    (a) the five files written by the reference's save_checkpoint are stored byte for byte (data
        files), for our loaders to read in tests;
    (b) the five files written by OUR trainer (CPU device) are loaded through the reference's
        load_checkpoint / load_weights / load_weights_{task_encoder,motor_decoder,world_model}; the
        resulting state dicts must equal the weights we saved, and the reference's forward on a
        fixed observation after loading them is recorded as the expected output."""
    from physicsvae_amd import train_physics_vae as OT
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import make_trainer
    data = R.synth_demo(seed=0, n_episodes=2, n_steps=14, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    sd = R.perturb_biases(R.init_state_dict(arch, seed=1), seed=3)
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        ref = make_reference_trainer(pkl, arch, 8, m_world=2)
        ref.model.load_state_dict(sd)
        d_ref = os.path.join(td, "ref_ck")
        os.makedirs(d_ref)
        ref.save_checkpoint(d_ref)
        for f in sorted(os.listdir(d_ref)):
            fix["ref_file::" + f] = np.frombuffer(open(os.path.join(d_ref, f), "rb").read(), dtype=np.uint8)
        # (b) our files -> the reference's loaders
        ours = make_trainer(arch, data, 8, device="cpu")
        sd2 = R.perturb_biases(R.init_state_dict(arch, seed=7), seed=9)      # different weights than `sd`
        ours.model.load_state_dict(sd2)
        d_our = os.path.join(td, "our_ck")
        os.makedirs(d_our)
        ret = ours.save_checkpoint(d_our)
        ok = {}
        ref.load_checkpoint(ret)                                              # tm:215-216, model.pth
        ok["load_checkpoint"] = all(torch.equal(v, sd2[k]) for k, v in ref.model.state_dict().items())
        ref.model.load_state_dict(sd)
        ref.model.load_weights(os.path.join(d_our, "model.pt"))               # rmt:873-875
        ok["load_weights"] = all(torch.equal(v, sd2[k]) for k, v in ref.model.state_dict().items())
        ref.model.load_state_dict(sd)
        ref.model.load_weights_task_encoder(os.path.join(d_our, "task_encoder.pt"))
        ref.model.load_weights_motor_decoder(os.path.join(d_our, "motor_decoder.pt"))
        ref.model.load_weights_world_model(os.path.join(d_our, "world_model.pt"))
        got = ref.model.state_dict()
        ok["per_net_loaders"] = all(torch.equal(v, (sd if k.startswith("_value_branch") else sd2)[k])
                                    for k, v in got.items())
        for k, v in ok.items():
            assert v, k
            fix["reference_accepts::" + k] = np.array(True)
        obs = torch.from_numpy(np.random.default_rng(5).standard_normal((4, 2 * arch["Db"])).astype(np.float32))
        ref.model.latent_prior_noise = False if hasattr(ref.model, "latent_prior_noise") else None
        with EpsPatch(lambda c, shape: torch.zeros(shape)):
            logits, _ = ref.model(input_dict={"obs": obs, "obs_flat": obs}, state=None, seq_lens=None)
        fix["obs"] = obs.numpy()
        fix["reference_logits_after_loading_our_files"] = logits.detach().numpy()
        fix["reference_future_state"] = ref.model._cur_future_state.detach().numpy()
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"])])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "reference accepts our files:", ok)


def case_helper(name, arch, batch=6):
    """`motor_decoder_helper_enable` (rmt:490-498, 670-680, 833-835): the reference's model built by its trainer with the
    helper switched on in custom_model_config (the trainer passes the dict through, tpv:290-311).  Recorded: the
    state-dict layout (the helper registers between the motor decoder and the world model), and the reference's forward
    on fixed observations at seeded weights -- logits, z, the world model's prediction (which sees the helped action,
    rmt:758) and the value -- with the sampler's noise off and on (draws supplied), plus `forward_decoder` alone."""
    harch = R.with_helper(arch)
    data = R.synth_demo(seed=0, n_episodes=2, n_steps=14, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        orig = T.update_model_config

        def update_model_config(trainer_config):
            orig(trainer_config)
            cmc = trainer_config["model"]["custom_model_config"]
            cmc["motor_decoder_helper_enable"] = True                  # rmt:490 (layers / range: the defaults, rmt:491-498)
        T.update_model_config = update_model_config
        try:
            tr = make_reference_trainer(pkl, arch, 8, m_world=2)
        finally:
            T.update_model_config = orig
        m = tr.model
        ref_sd = m.state_dict()
        fix["sd_keys"] = np.array(list(ref_sd.keys()))
        fix["sd_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in ref_sd.values()])
        fix["helper_range"] = np.array(m._motor_decoder_helper_range)
        sd = R.perturb_biases(R.init_state_dict(harch, seed=1), seed=3)
        # (an output layer of norm 0.01 would make the helper's term 1e-3 of the action: give it weight)
        k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(harch["mh"])
        sd[k_out] = sd[k_out] * 60.0
        m.load_state_dict(sd)
        m.eval()
        obs = torch.from_numpy(np.random.default_rng(5).standard_normal((batch, 2 * arch["Db"])).astype(np.float32))
        eps = R.eps_stream(2, arch["Z"])(0, (batch, arch["Z"]))
        fix["obs"], fix["eps"] = obs.numpy(), eps.numpy()
        for noise in (False, True):
            tag = "noise" if noise else "mean"
            m.latent_prior_noise = noise
            with EpsPatch(lambda c, shape: eps):
                with torch.no_grad():
                    logits, _ = m(input_dict={"obs": obs, "obs_flat": obs}, state=None, seq_lens=None)
            fix[tag + "_logits"] = logits.numpy()
            fix[tag + "_z"] = m._cur_task_encoder_variable.numpy()
            fix[tag + "_future_state"] = m._cur_future_state.numpy()
            fix[tag + "_value"] = m.value_function().detach().numpy()
        with torch.no_grad():
            zb, zt = obs[:, : arch["Db"]], eps
            fix["decoder_logits"] = m.forward_decoder(zb, zt, None, None, 0)[0].numpy()
        # the five-plus-one files of the reference for a helper model: which loaders / savers exist
        f = os.path.join(td, "helper.pt")
        m.save_weights_motor_decoder_helper(f)
        fix["helper_file_keys"] = np.array(list(torch.load(f).keys()))
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), batch])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "keys", len(fix["sd_keys"]), "max |logits| mean", float(np.abs(fix["mean_logits"]).max()))


def case_subsets(name, arch, batch=8):
    """`task_encoder_inputs` / `motor_decoder_inputs` (rmt:470, 485; 607-613, 646-653, 776-783, 822-829): the
    reference's model built by its trainer with subsets of ["body", "task"] switched on in custom_model_config
    (the trainer passes the dict through).  Per combination: the state-dict layout (the first layers are narrower),
    and one minibatch of the reference's own compute_loss in both phases at seeded weights -- total, z, the world
    model's prediction, every gradient."""
    combos = [(("task",), ("body", "task")), (("body",), ("task",)), (("body", "task"), ("body",)), (("task",), ("task",))]
    data = R.synth_demo(seed=0, n_episodes=2, n_steps=14, dim_body=arch["Db"], dim_action=arch["Da"], kind="iid")
    fix = {"combos": np.array(["%s/%s" % ("+".join(t), "+".join(d)) for t, d in combos])}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        for ci, (te_in, md_in) in enumerate(combos):
            orig = T.update_model_config

            def update_model_config(trainer_config):
                orig(trainer_config)
                cmc = trainer_config["model"]["custom_model_config"]
                cmc["task_encoder_inputs"] = list(te_in)
                cmc["motor_decoder_inputs"] = list(md_in)
            T.update_model_config = update_model_config
            try:
                tr = make_reference_trainer(pkl, arch, batch, m_world=2)
            finally:
                T.update_model_config = orig
            ref_sd = tr.model.state_dict()
            pre = "c%d_" % ci
            fix[pre + "sd_keys"] = np.array(list(ref_sd.keys()))
            fix[pre + "sd_shapes"] = np.array([list(v.shape) + [0] * (2 - v.dim()) for v in ref_sd.values()])
            sarch = R.with_inputs(arch, te_in, md_in)
            tr.model.load_state_dict(R.perturb_biases(R.init_state_dict(sarch, seed=1), seed=3))
            x, y = next(iter(tr.train_loader))
            eps = R.eps_stream(2, arch["Z"])(0, (x.shape[0], arch["Z"]))
            fix[pre + "eps"] = eps.numpy()
            for world in (True, False):
                tag = pre + ("world" if world else "joint")
                m = tr.model
                m.set_learnable_task_encoder(not world)
                m.set_learnable_motor_decoder(not world)
                m.set_learnable_world_model(world)
                tr.read_loss_fn_coeff(world=world)
                m.train()
                tr.optimizer.zero_grad()
                with EpsPatch(lambda c, shape: eps):
                    loss = tr.compute_loss(y, x)
                loss.backward()
                fix[tag + "_total"] = loss.detach().numpy()
                fix[tag + "_z"] = m._cur_task_encoder_variable.detach().numpy()
                fix[tag + "_future_state"] = m._cur_future_state.detach().numpy()
                grads = grads_of(m)
                fix[tag + "_grad_keys"] = np.array(list(grads.keys()))
                for k, g in grads.items():
                    fix["%s_grad::%s" % (tag, k)] = g.numpy()
            print("  ", fix["combos"][ci], "world", fix[pre + "world_total"], "joint", fix[pre + "joint_total"])
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), 2, 14, batch])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name)


def case_helper_train(name, arch, n_ep=3, n_steps=21, batch=8, m_world=2, n_epochs=5, lookahead=1):
    """Supervised training of a model with `motor_decoder_helper_enable` (rmt:490-498, 670-680, 833-835): the helper's
    term sits inside a_hat, so the reference's own compute_loss / optimizer train the helper together with the decoder
    (nothing ever freezes it: tpv:326-329, 347-350 switch encoder, decoder and world model only).  Recorded: one
    minibatch in both phases at seeded weights (total, every gradient -- the helper has none in the world phase and one
    in the joint phase), and the reference's own training loop over `n_epochs` epochs (epoch losses, final weights)."""
    harch = R.with_helper(arch)
    data = R.synth_demo(seed=0, n_episodes=n_ep, n_steps=n_steps, dim_body=arch["Db"], dim_action=arch["Da"], kind="dynamics")
    fix = {}
    with tempfile.TemporaryDirectory() as td:
        pkl = os.path.join(td, "demo.pkl")
        R.write_demo(pkl, data)
        orig = T.update_model_config

        def update_model_config(trainer_config):
            orig(trainer_config)
            trainer_config["model"]["custom_model_config"]["motor_decoder_helper_enable"] = True
        T.update_model_config = update_model_config
        try:
            tr = make_reference_trainer(pkl, arch, batch, m_world=m_world, lookahead=lookahead)
        finally:
            T.update_model_config = orig
        m = tr.model
        fix["helper_range"] = np.array(m._motor_decoder_helper_range)
        sd = R.perturb_biases(R.init_state_dict(harch, seed=1), seed=3)
        k_out = "_motor_decoder_helper._model.%d._model.0.weight" % len(harch["mh"])
        sd[k_out] = sd[k_out] * 60.0               # (norm 0.01 rows would make the helper's term 1e-3 of the action)
        m.load_state_dict(sd)
        x, y = next(iter(tr.train_loader))
        eps = R.eps_stream(2, arch["Z"])(0, (x.shape[0], arch["Z"]))
        fix["eps"] = eps.numpy()
        # (lookahead > 1: one draw per unrolled step, eps(t); the world phase then reaches the helper too -- the state the
        #  world model is asked to continue from is its own prediction under the HELPED action, tpv:417-421)
        eps_t = [R.eps_stream(2, arch["Z"])(t, (x.shape[0], arch["Z"])) for t in range(lookahead)]
        for world in (True, False):
            tag = "world" if world else "joint"
            m.set_learnable_task_encoder(not world)
            m.set_learnable_motor_decoder(not world)
            m.set_learnable_world_model(world)
            tr.read_loss_fn_coeff(world=world)
            m.train()
            tr.optimizer.zero_grad()
            with EpsPatch((lambda c, shape: eps) if lookahead == 1 else (lambda c, shape: eps_t[c % lookahead])):
                loss = tr.compute_loss(y, x)
            loss.backward()
            fix[tag + "_total"] = loss.detach().numpy()
            grads = grads_of(m)
            fix[tag + "_grad_keys"] = np.array(list(grads.keys()))
            for k, g in grads.items():
                fix["%s_grad::%s" % (tag, k)] = g.numpy()
        # the reference's own loop from the same weights, in a fresh trainer (phase switch after m_world epochs)
        T.update_model_config = update_model_config
        try:
            tr = make_reference_trainer(pkl, arch, batch, m_world=m_world, lookahead=lookahead)
        finally:
            T.update_model_config = orig
        tr.lr_scheduler = torch.optim.lr_scheduler.StepLR(tr.optimizer, step_size=2, gamma=0.7)
        tr.model.load_state_dict(sd)
        es = R.eps_stream(2, arch["Z"])
        losses = []
        with EpsPatch(lambda c, shape: es(c, shape)):
            for e in range(n_epochs):
                losses.append(tr.train()["mean_train_loss"])
        fix["epoch_losses"] = np.array(losses, dtype=np.float64)
        for k, v in tr.model.state_dict().items():
            fix["final::" + k] = v.detach().numpy().copy()
        named = dict(tr.model.named_parameters())
        fix["adam_keys"] = np.array(list(named.keys()))
        fix["adam_steps"] = np.array([float(tr.optimizer.state.get(p, {}).get("step", -1.0)) for p in named.values()])
    fix["meta"] = np.array([arch["Db"], arch["Da"], arch["Z"], *wd(arch["te"]), *wd(arch["md"]), *wd(arch["wm"]), n_ep, n_steps, batch,
                            m_world, n_epochs] + ([lookahead] if lookahead > 1 else []))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fix)
    print("wrote", name, "world", fix["world_total"], "joint", fix["joint_total"], "epochs", losses)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    tiny = R.make_arch(7, 3, latent=4, te=(16, 2), md=(24, 2), wm=(32, 2))
    c1 = R.make_arch(197, 45, latent=32, te=(256, 2), md=(256, 2), wm=(256, 2))
    c2 = R.make_arch(197, 45, latent=32, te=(1024, 4), md=(1024, 4), wm=(1024, 4))
    dflt = R.make_arch(197, 45)              # DEFAULT_CONFIG arch: 26-tensor layout
    # stacks gen_layers cannot emit but FC accepts (rmt:234-270): per-layer widths and activations
    mixed_tiny = dict(tiny, te=[(16, "relu"), (24, "tanh")], md=[(32, "elu"), (16, "relu"), (24, "sigmoid")],
                      wm=[(40, "relu"), (24, "linear"), (32, "relu")])
    mixed_c1 = dict(c1, te=[(256, "relu"), (128, "tanh")], md=[(512, "relu"), (256, "elu"), (128, "relu")],
                    wm=[(1024, "relu"), (512, "sigmoid"), (192, "relu")])
    pyramid_c1 = dict(c1, te=[(512, "relu"), (256, "relu"), (96, "relu")], md=[(160, "relu"), (320, "relu")],
                      wm=[(1024, "relu"), (256, "relu")])
    jobs = {
        "single_mixed_tiny": lambda: case_single("single_mixed_tiny", mixed_tiny, 2, 14, 8, full=True),
        "single_mixed_c1": lambda: case_single("single_mixed_c1", mixed_c1, 2, 200, 64, full=False),
        "single_pyramid_c1_clear": lambda: case_single_clear("single_pyramid_c1_clear", pyramid_c1, 2, 200, 64),
        "train_mixed_tiny": lambda: case_training("train_mixed_tiny", mixed_tiny, 3, 21, 8, m_world=2, n_epochs=5,
                                                  full=True),
        "look2_mixed_tiny": lambda: case_lookahead("look2_mixed_tiny", mixed_tiny, 2, 15, 8, lookahead=2, full=True),
        "single_tiny": lambda: case_single("single_tiny", tiny, 2, 14, 8, full=True),
        "single_c1": lambda: case_single("single_c1", c1, 2, 200, 64, full=False),
        "single_c2": lambda: case_single("single_c2", c2, 2, 300, 256, full=False),
        "single_default": lambda: case_single("single_default", dflt, 2, 100, 32, full=False),
        # minibatches off every ReLU kink (margin 1e-5): the captures the HIP gradients are held to at 1e-4
        "single_c1_clear": lambda: case_single_clear("single_c1_clear", c1, 2, 200, 64),
        "single_c2_clear": lambda: case_single_clear("single_c2_clear", c2, 2, 300, 256),
        "single_default_clear": lambda: case_single_clear("single_default_clear", dflt, 2, 100, 32),
        "single_tiny_clear": lambda: case_single_clear("single_tiny_clear", tiny, 2, 14, 8),
        # the reference's compute_loss + backward at TRAINED weights (30 world + 40 joint epochs, six LR decays)
        "trained_c1": lambda: case_trained("trained_c1", c1, 4, 200, 64, m_world=30, n_epochs=70, lr_step=10),
        "train_tiny": lambda: case_training("train_tiny", tiny, 3, 21, 8, m_world=2, n_epochs=5,
                                            full=True),
        "train_c1": lambda: case_training("train_c1", c1, 4, 200, 64, m_world=2, n_epochs=4,
                                          full=False),
        "anchor_c1": lambda: case_anchor("anchor_c1"),
        "ingest_tiny": lambda: case_ingest("ingest_tiny", tiny),
        "ingest_rel_tiny": lambda: case_ingest_rel("ingest_rel_tiny", tiny),
        "noprior_tiny": lambda: case_noprior("noprior_tiny", tiny, 2, 14, 8),
        "helper_tiny": lambda: case_helper("helper_tiny", tiny),
        "subsets_tiny": lambda: case_subsets("subsets_tiny", tiny),
        "helper_train_tiny": lambda: case_helper_train("helper_train_tiny", tiny),
        "helper_train_look2_tiny": lambda: case_helper_train("helper_train_look2_tiny", tiny, n_steps=22, lookahead=2),
        "helper_default": lambda: case_helper("helper_default", dflt),
        # the trainer's "act_fn" (hidden activation of every stack) and Adam's weight_decay: config keys a user edits
        "single_tiny_tanh": lambda: case_single("single_tiny_tanh", dict(tiny, act="tanh"), 2, 14, 8, full=True),
        "single_tiny_sigmoid": lambda: case_single("single_tiny_sigmoid", dict(tiny, act="sigmoid"), 2, 14, 8, full=True),
        "single_tiny_elu": lambda: case_single("single_tiny_elu", dict(tiny, act="elu"), 2, 14, 8, full=True),
        "single_c1_tanh": lambda: case_single("single_c1_tanh", dict(c1, act="tanh"), 2, 200, 64, full=False),
        "train_tiny_elu_wd": lambda: case_training("train_tiny_elu_wd", dict(tiny, act="elu"), 3, 21, 8, m_world=2,
                                                   n_epochs=5, full=True, weight_decay=0.01),
        "ckpt_interop_tiny": lambda: case_checkpoint_interop("ckpt_interop_tiny", tiny),
        "look3_tiny": lambda: case_lookahead("look3_tiny", tiny, 2, 15, 8, lookahead=3, full=True),
        "look2_c1": lambda: case_lookahead("look2_c1", c1, 2, 200, 64, lookahead=2, full=False),
        "l1_tiny": lambda: case_lookahead("l1_tiny", tiny, 2, 15, 8, lookahead=1, full=True, loss="L1"),
        "l1_look2_c1": lambda: case_lookahead("l1_look2_c1", c1, 2, 200, 64, lookahead=2, full=False, loss="L1"),
        "shuffle_tiny": lambda: case_shuffle("shuffle_tiny", tiny, 3, 21, 8, m_world=2, n_epochs=5),
        "train_tiny_look2": lambda: case_training("train_tiny_look2", tiny, 3, 22, 8, m_world=2, n_epochs=5,
                                                  full=True, lookahead=2),
    }
    for k, fn in jobs.items():
        if a.only is None or a.only == k:
            torch.manual_seed(0)              # every fixture from the same generator state, whatever ran before it:
            fn()                              # `--only NAME` and a full run write the same bytes (oracle/verify_golden.py)


if __name__ == "__main__":
    main()
