"""oracle/refpath.py -- CPU restatement of the PhysicsVAE supervised-training hot path.

TEST INFRASTRUCTURE, NOT PRODUCT.  Only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import this module.  The product package
(`physicsvae_amd/`) never imports it and fails loudly when its HIP library is missing.

What it restates (stock torch CPU fp32 ops, in the reference's op order), with the
reference lines each piece follows (`tpv` = train_physics_vae.py, `tm` = torch_models.py,
`rmt` = rllib_model_torch.py under /root/reference):

  * synthetic demo dict in the on-disk schema                     tpv:57-92
  * sliding (s_t, s_t+1, a_t) windows, float64                    tpv:117-164
  * per-sample Dataset + sequential DataLoader, partial last batch tm:39-95,166-193
  * FC stacks / normc init / state_dict key layout                rmt:234-283 + ray 1.11.0
    SlimFC/normc_initializer (third-party, not in tree: semantics per SURVEY.md App. B)
  * PhysicsVAE forward: TE -> reparameterise -> MD -> WM -> VB     rmt:742-853
  * losses (a-rec MSE, KL to N(0,I), s-rec MSE, cycle MSE)        tpv:361-435
  * two-phase schedule, freezing, Adam, StepLR-per-epoch          tpv:314-351, tm:110-161
  * five-file checkpoint layout                                   tpv:440-467, rmt:870-928

Pinning: `tests/golden/*.npz` hold outputs of the *reference itself* (imported in the dev
container through `oracle/stubs`, script `oracle/gen_golden.py`); `tests/test_oracle_golden.py`
checks this restatement against them.  The reference ships no tests or golden vectors of
its own, so those captures are the pin.
"""
import math
import pickle
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

NETS = ("_task_encoder", "_motor_decoder", "_world_model", "_value_branch")

# --------------------------------------------------------------------------------------
# Latent priors other than N(0, I): OUR SPECIFICATION, not a capture.
#
# The reference sketches two more `latent_prior_type`s (rmt:614-635, 795-819; tpv:390-409) and both
# crash before the first loss value exists, so no golden vector can be recorded ("parity unpinned"
# for these two options only; everything they share with the default prior is pinned):
#   * "normal_state_mean_one_std": the constructor passes `layers=latent_prior_type` (a string) to
#     create_layer (rmt:632), and the loss reads `self.model._cur_vae_prior_mu`, an attribute that is
#     never set (the model stores `_cur_latent_prior_mu`, rmt:808; tpv:395-396);
#   * "hypersphere_uniform": the same missing attribute (tpv:406).
# What we build is the evident intent of that code, with the two slips fixed and nothing else changed:
#
#   normal_state_mean_one_std -- a learned prior mean.  `_latent_prior` = MLP(Db -> Z) over the body
#     state, layers = `latent_prior_layers` (ours: the task encoder's width/depth when None), registered
#     BEFORE the task encoder (rmt:627-635), so its keys lead the state_dict.  The encoder still emits
#     (mu, logvar) and z = mu + eps*exp(logvar/2).  KL term = KL(N(mu, sigma^2) || N(mu_p, 1)) summed over
#     the latent dims and averaged over the minibatch:
#         loss_kl = mean_i sum_j 0.5 * (exp(logvar) + (mu - mu_p)^2 - 1 - logvar)
#     (the closed form of the torch.distributions call of tpv:397-402 with logvar_p = 0, rmt:809).  The
#     sketch sums over the rows and averages over the dims (`kl_div += ...; kl_div.mean()`), i.e. B/Z
#     times this; we keep the normalisation of the default prior so that mu_p = 0 reproduces
#     "normal_zero_mean_one_std" exactly and vae_kl_coeff keeps its meaning.  The prior net receives
#     gradient only through this term and trains with the encoder/decoder in the joint phase (it is never
#     frozen upstream either: rmt:930-950 has no switch for it).
#
#   hypersphere_uniform -- a deterministic encoder on the unit sphere.  The encoder emits Z values
#     (rmt:620-621), mu = e/|e| (F.normalize, eps 1e-12; rmt:811); the decoder receives z = mu (the
#     sketch passes the un-normalised e on, which makes "one radius" meaningless -- fixed).  The prior
#     sample of that forward is u = n/|n| with n = randn_like(mu) (rmt:813-814) and
#         loss_kl = mean_i <mu_i, u_i>                                     (tpv:404-407)
#     `latent_prior_noise` only gates n (False: u = 0, the term vanishes).
# --------------------------------------------------------------------------------------
PRIORS = ("normal_zero_mean_one_std", "normal_state_mean_one_std", "hypersphere_uniform")
# `latent_prior_type = False` (rmt:622-623, 815-816; tpv:384 `if self.latent_prior_type and ...`) is a mode the
# reference DOES run: the encoder emits Z values that go to the decoder as they are, no sampling, no KL term.
# It is restated below and pinned by a capture of the reference (tests/golden/noprior_tiny.npz).


# --------------------------------------------------------------------------------------
# architecture description
# --------------------------------------------------------------------------------------
def make_arch(dim_body, dim_action, latent=32, te=(256, 2), md=(512, 3), wm=(1024, 2),
              vb=(256, 2), prior="normal_zero_mean_one_std", pr=None, act="relu"):
    """Widths/depths as `gen_layers(width, depth)` expands them (tpv:180-192, 290-311);
    defaults are PhysicsVAE.DEFAULT_CONFIG (rmt:462-510).  `prior` / `pr`: see PRIORS above
    (`pr` = (width, depth) of the learned prior, default = the task encoder's).  `act`: the trainer's
    "act_fn" (tpv:262 -> gen_layers' act_hidden -> get_activation_fn rmt:30-46), one for every stack."""
    assert prior in PRIORS or prior is False, prior
    assert act in ACTIVATIONS, act
    return dict(Db=int(dim_body), Da=int(dim_action), Z=int(latent),
                te=tuple(te), md=tuple(md), wm=tuple(wm), vb=tuple(vb), prior=prior,
                pr=tuple(pr) if pr is not None else tuple(te), act=act)


BOTH = ("body", "task")


def with_inputs(arch, te_inputs=BOTH, md_inputs=BOTH):
    """`arch` with `task_encoder_inputs` / `motor_decoder_inputs` (rmt:470, 485): which of the two halves of the
    observation the task encoder reads (rmt:609-612, 776-783) and which of (s_body, z) the motor decoder reads
    (rmt:646-652, 822-829).  Any non-empty subset of ("body", "task"), in that order."""
    for v in (te_inputs, md_inputs):
        assert tuple(v) in (BOTH, ("body",), ("task",)), v
    return dict(arch, te_inputs=tuple(te_inputs), md_inputs=tuple(md_inputs))


def input_widths(arch):
    """(task encoder's, motor decoder's) input width: rmt:607-612, 646-653 (dim_state_task = dim_body in training)."""
    Db, Z = arch["Db"], arch["Z"]
    te, md = arch.get("te_inputs", BOTH), arch.get("md_inputs", BOTH)
    return (Db * ("body" in te) + Db * ("task" in te), Db * ("body" in md) + Z * ("task" in md))


HELPER_DEFAULT = ((128, "relu"), (128, "relu"))        # motor_decoder_helper_layers' hidden part (rmt:491-495); output: tanh


def with_helper(arch, hidden=HELPER_DEFAULT, rng=0.5):
    """`arch` with the motor decoder's helper switched on (rmt:490-498, 670-680): a second stack on the decoder's input
    [s1 | z] with a tanh output layer whose `range`-scaled output is added to the action half of the logits
    (rmt:833-835).  Registered between the motor decoder and the world model."""
    assert rng > 0                                         # rmt:673
    return dict(arch, mh=[(int(w), a) for w, a in hidden], mh_range=float(rng))


ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "elu": nn.ELU}      # rmt:37-44


def hidden_layers(spec, default_act="relu"):
    """Hidden layers of one stack as [(width, activation)].  `spec` is (width, depth) -- what gen_layers expands
    (tpv:180-192), every layer with `default_act` -- or a list of (width, activation) pairs: the general layer
    list FC.__init__ accepts (rmt:234-270; "linear" / None = no activation module, rmt:32-33)."""
    if len(spec) and isinstance(spec[0], (tuple, list)):
        out = [(int(w), "linear" if a is None else a) for w, a in spec]
        assert all(a in ACTIVATIONS or a == "linear" for _, a in out), out
        return out
    w, d = spec
    return [(int(w), default_act)] * int(d)


def with_stacks(arch, stacks_json):
    """`arch` with the per-layer stacks a fixture recorded as its "stacks" entry (json of {"te" / "md" / "wm":
    [[width, activation], ...]}, written by oracle/gen_golden.py's act_meta)."""
    import json
    out = dict(arch)
    out.update({k: [(int(w), a) for w, a in v] for k, v in json.loads(str(stacks_json)).items()})
    out["pr"] = out["te"]
    return out


def stack_acts(arch, key):
    """Activation names of the hidden layers of arch[key] ("te" / "md" / "wm" / "pr" / "vb")."""
    default = "relu" if key == "vb" else arch.get("act", "relu")    # value_fn_layers keep their own relu (rmt:500)
    return [a for _, a in hidden_layers(arch.get(key, arch["te"]), default)]


def fc_layer_list(spec, default_act="relu"):
    """The `*_layers` list of dicts (rmt:462-510 keys) describing a stack: gen_layers' output for (width, depth)."""
    hid = hidden_layers(spec, default_act)
    return [{"type": "fc", "hidden_size": w, "activation": a, "init_weight": {"name": "normc", "std": 1.0}} for w, a in hid] + [
        {"type": "fc", "hidden_size": "output", "activation": "linear", "init_weight": {"name": "normc", "std": 0.01}}]


def net_layer_dims(arch):
    """(in, out) of every Linear, per net, in registration order (rmt:638-699)."""
    Db, Da, Z = arch["Db"], arch["Da"], arch["Z"]

    def chain(n_in, spec, n_out):
        dims, prev = [], n_in
        for w, _ in hidden_layers(spec):
            dims.append((prev, w))
            prev = w
        dims.append((prev, n_out))
        return dims

    prior = arch.get("prior", PRIORS[0])
    te_out = Z if (prior == "hypersphere_uniform" or prior is False) else 2 * Z   # rmt:618-623
    learned = [("_latent_prior", chain(Db, arch.get("pr", arch["te"]), Z))] if prior == PRIORS[1] else []
    te_in, md_in = input_widths(arch)
    return OrderedDict(learned + [                                             # rmt:627-635 comes first
        ("_task_encoder", chain(te_in, arch["te"], te_out)),       # rmt:638-644, 607-613
        ("_motor_decoder", chain(md_in, arch["md"], Da)),          # rmt:646-668
    ] + ([("_motor_decoder_helper", chain(md_in, arch["mh"], Da))] if arch.get("mh") else []) + [   # rmt:670-680
        ("_world_model", chain(Db + Da, arch["wm"], Db)),          # rmt:682-689
        ("_value_branch", chain(2 * Db, arch["vb"], 1)),           # rmt:693-699
    ])


def state_dict_spec(arch):
    """Ordered [(key, shape)] exactly as `PhysicsVAE.state_dict()` lists them:
    `<net>._model.<i>._model.0.{weight,bias}` (FC -> SlimFC -> Sequential(Linear[,ReLU]))."""
    spec = []
    for net, dims in net_layer_dims(arch).items():
        for i, (n_in, n_out) in enumerate(dims):
            spec.append(("%s._model.%d._model.0.weight" % (net, i), (n_out, n_in)))
            spec.append(("%s._model.%d._model.0.bias" % (net, i), (n_out,)))
    return spec


def init_state_dict(arch, seed):
    """normc init (rows L2-normalised to 1.0 for hidden layers, 0.01 for the output
    layer; bias 0 -- tpv:184-189, ray normc_initializer) drawn from numpy so that both
    sides of a parity test regenerate bit-identical weights from a seed."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for net, dims in net_layer_dims(arch).items():
        for i, (n_in, n_out) in enumerate(dims):
            std = 0.01 if i == len(dims) - 1 else 1.0
            w = rng.standard_normal((n_out, n_in))
            w *= std / np.sqrt((w * w).sum(axis=1, keepdims=True))
            sd["%s._model.%d._model.0.weight" % (net, i)] = torch.from_numpy(w.astype(np.float32))
            sd["%s._model.%d._model.0.bias" % (net, i)] = torch.zeros(n_out, dtype=torch.float32)
    return sd


def perturb_biases(sd, seed, scale=0.05):
    """Parity inputs should not have all-zero biases (a wrong bias path would go unseen)."""
    rng = np.random.default_rng(seed)
    out = OrderedDict()
    for k, v in sd.items():
        if k.endswith(".bias"):
            out[k] = torch.from_numpy((rng.standard_normal(v.shape) * scale).astype(np.float32))
        else:
            out[k] = v.clone()
    return out


# --------------------------------------------------------------------------------------
# synthetic demonstrations in the reference's pickle schema (tpv:57-92; writer
# envs/rllib_env_imitation.py:63-87,140-144)
# --------------------------------------------------------------------------------------
def synth_demo(seed, n_episodes, n_steps, dim_body, dim_action, kind="iid", quantum=1024.0):
    """kind="iid": N(0,1) states, N(0,1) actions clipped to +-3 (SURVEY.md 8c anchor).
    kind="dynamics": s_{t+1} = tanh(A s_t + B a_t) + noise, so the world model has
    something to learn.  Values are rounded to multiples of 1/quantum so the float64 ->
    float32 conversion (tm:67) is exact on every platform."""
    rng = np.random.default_rng(seed)
    episodes = []
    if kind == "dynamics":
        A = rng.standard_normal((dim_body, dim_body)) / math.sqrt(dim_body)
        Bm = rng.standard_normal((dim_action, dim_body)) / math.sqrt(dim_action)
    for _ in range(n_episodes):
        act = np.clip(rng.standard_normal((n_steps, dim_action)), -3.0, 3.0)
        if kind == "iid":
            sb = rng.standard_normal((n_steps, dim_body))
        elif kind == "dynamics":
            sb = np.empty((n_steps, dim_body))
            sb[0] = rng.standard_normal(dim_body)
            for t in range(n_steps - 1):
                sb[t + 1] = np.tanh(sb[t] @ A + act[t] @ Bm) + 0.05 * rng.standard_normal(dim_body)
        else:
            raise ValueError(kind)
        if quantum:
            sb = np.round(sb * quantum) / quantum
            act = np.round(act * quantum) / quantum
        episodes.append({
            "time": [float(t) / 30.0 for t in range(n_steps)],
            "state": [np.concatenate([sb[t], sb[min(t + 1, n_steps - 1)]]) for t in range(n_steps)],
            "state_body": [sb[t].copy() for t in range(n_steps)],
            "state_task": [sb[min(t + 1, n_steps - 1)].copy() for t in range(n_steps)],
            "action": [act[t].copy() for t in range(n_steps)],
            "reward": [0.0] * n_steps,
        })
    return {
        "dim_action": dim_action, "dim_state": 2 * dim_body, "dim_state_body": dim_body,
        "dim_state_task": dim_body, "exp_std": 0.05, "iter_per_episode": 1,
        "episodes": episodes,
    }


def survey_anchor_demo():
    """SURVEY.md 8(c) anchor inputs: numpy default_rng(0), per episode standard_normal((1000,197))
    then standard_normal((1000,45)).clip(-3,3), 10 episodes (values NOT rounded)."""
    rng = np.random.default_rng(0)
    eps = []
    for _ in range(10):
        sb = rng.standard_normal((1000, 197))
        a = rng.standard_normal((1000, 45)).clip(-3, 3)
        eps.append({"time": [t / 30.0 for t in range(1000)],
                    "state": [np.concatenate([sb[t], sb[min(t + 1, 999)]]) for t in range(1000)],
                    "state_body": [sb[t] for t in range(1000)], "state_task": [sb[min(t + 1, 999)] for t in range(1000)],
                    "action": [a[t] for t in range(1000)], "reward": [0.0] * 1000})
    return {"dim_action": 45, "dim_state": 394, "dim_state_body": 197, "dim_state_task": 197, "exp_std": 0.05,
            "iter_per_episode": 1, "episodes": eps}


def write_demo(path, data):
    with open(path, "wb") as f:
        pickle.dump(data, f)


def merge_demo_files(files):
    """tpv:94-114: first file is the base, later files must agree on the meta fields
    and contribute their episodes."""
    merged = None
    for n, path in enumerate(files):
        with open(path, "rb") as f:
            d = pickle.load(f)
        if n == 0:
            merged = d
            continue
        for key in ("iter_per_episode", "dim_state", "dim_state_body", "dim_state_task",
                    "dim_action", "exp_std"):
            assert merged[key] == d[key], key
        merged["episodes"] = merged["episodes"] + d["episodes"]
    return merged


def build_windows(data, lookahead=1, num_samples=None, cond="abs"):
    """tpv:117-164 with use_a_gt=False.  Returns float64
    X[N, L, 2*Db] = [sb[i+j] | sb[i+j+1]] (cond "abs") or [sb[i+j] | sb[i+j+1] - sb[i+j]] (cond "rel",
    tpv:149-150), Y[N, L, Da] = a[i+j]; episode order, i ascending; the `num_samples` cap stops at
    exactly that many windows (tpv:137-138)."""
    assert lookahead >= 1 and cond in ("abs", "rel")
    X, Y = [], []
    for ep in data["episodes"]:
        T = len(ep["time"])
        assert T >= lookahead
        for i in range(T - lookahead):
            if num_samples is not None and len(X) >= num_samples:
                break
            X.append(np.vstack([np.hstack([ep["state_body"][i + j],
                                           ep["state_body"][i + j + 1] - (ep["state_body"][i + j] if cond == "rel" else 0.0)])
                                for j in range(lookahead)]))
            Y.append(np.vstack([ep["action"][i + j] for j in range(lookahead)]))
    return np.array(X), np.array(Y)


class WindowDataset(torch.utils.data.Dataset):
    """tm:39-95 with normalize_x = normalize_y = False (tpv:163-164): per-sample
    float64 -> float32 `torch.Tensor(...)` copies."""

    def __init__(self, X, Y):
        self.X, self.Y = X, Y

    def __len__(self):
        return len(self.X)

    def __getitem__(self, i):
        return torch.Tensor(self.X[i]), torch.Tensor(self.Y[i])


def make_loader(X, Y, batch_size, shuffle=None):
    """tm:166-193: shuffle is None in practice (the "suffle_data" typo, tpv:260 vs tm:181)
    -> SequentialSampler, drop_last False.  With `shuffle_data` spelt right (tm:181) it is torch's own shuffled
    loader: a RandomSampler seeded from the default generator on every pass."""
    return torch.utils.data.DataLoader(WindowDataset(X, Y), batch_size=batch_size, shuffle=shuffle)


# --------------------------------------------------------------------------------------
# model (own nn.Module reproducing the state_dict layout)
# --------------------------------------------------------------------------------------
class _Slim(nn.Module):
    def __init__(self, n_in, n_out, relu, act="relu"):
        super().__init__()
        mods = [nn.Linear(n_in, n_out)]
        if relu and act != "linear":
            mods.append(ACTIVATIONS[act]())
        self._model = nn.Sequential(*mods)

    def forward(self, x):
        return self._model(x)


class _Stack(nn.Module):
    def __init__(self, dims, act="relu", out_act=None):
        super().__init__()
        acts = [act] * (len(dims) - 1) if isinstance(act, str) else list(act)
        last = len(dims) - 1
        self._model = nn.Sequential(*[_Slim(i, o, relu=(n < last or out_act is not None), act=acts[n] if n < last else (out_act or "relu"))
                                      for n, (i, o) in enumerate(dims)])

    def forward(self, x):
        return self._model(x)


class RefModel(nn.Module):
    """PhysicsVAE restated (rmt:461-950): the latent priors of PRIORS (and False), any `task_encoder_inputs` /
    `motor_decoder_inputs` (with_inputs), the helper (with_helper), constant log_std (sample_std 0.1)."""

    def __init__(self, arch):
        super().__init__()
        self.arch = arch
        dims = net_layer_dims(arch)
        self.prior = arch.get("prior", PRIORS[0])
        if "_latent_prior" in dims:                                   # rmt:627-635: registered first
            self._latent_prior = _Stack(dims["_latent_prior"], stack_acts(arch, "pr"))
        self._task_encoder = _Stack(dims["_task_encoder"], stack_acts(arch, "te"))
        self._motor_decoder = _Stack(dims["_motor_decoder"], stack_acts(arch, "md"))
        if arch.get("mh"):                                            # rmt:670-680 (its output layer ends in tanh, rmt:672)
            self._motor_decoder_helper = _Stack(dims["_motor_decoder_helper"], stack_acts(arch, "mh"), out_act="tanh")
        self._world_model = _Stack(dims["_world_model"], stack_acts(arch, "wm"))
        self._value_branch = _Stack(dims["_value_branch"], stack_acts(arch, "vb"))
        self.log_std = math.log(0.1)            # AppendLogStd constant (rmt:160-206, 466)
        self.latent_prior_noise = True          # rmt:705
        self.eps_source = None                  # callable(shape) -> eps, else torch.randn
        self.trace = None                       # list -> one dict of internals per forward call

    # rmt:734-740
    def reparameterize(self, mu, logvar):
        if not self.latent_prior_noise:
            return mu
        std = torch.exp(0.5 * logvar)
        eps = self.eps_source(std.shape) if self.eps_source is not None else torch.randn_like(std)
        self.cur_eps = eps
        return mu + eps * std

    # rmt:742-771 (+773-853)
    def forward(self, obs):
        Db, Da, Z = self.arch["Db"], self.arch["Da"], self.arch["Z"]
        obs = obs.float()
        te_in = self.arch.get("te_inputs", BOTH)                      # rmt:776-783
        obs_task = obs if te_in == BOTH else (obs[..., :Db] if te_in == ("body",) else obs[..., Db:])
        h = self._task_encoder(obs_task)                              # rmt:788-793
        if self.prior is False:                                       # rmt:815-816: the code is the encoder output
            self.cur_mu, self.cur_logvar, self.cur_prior_mu = h, None, None
            z = h
        elif self.prior == "hypersphere_uniform":                       # rmt:810-814, fixed as specified above
            self.cur_mu, self.cur_logvar = nn.functional.normalize(h), None
            z = self.cur_mu
            if self.latent_prior_noise:
                n = self.eps_source(h.shape) if self.eps_source is not None else torch.randn_like(h)
                self.cur_eps = n
                self.cur_prior_mu = nn.functional.normalize(n)
            else:
                self.cur_prior_mu = torch.zeros_like(h)
        else:
            self.cur_mu, self.cur_logvar = h[..., :Z], h[..., Z:]     # rmt:795-800
            z = self.reparameterize(self.cur_mu, self.cur_logvar)
            if self.prior == "normal_state_mean_one_std":             # rmt:801-809
                self.cur_prior_mu = self._latent_prior(obs[..., :Db])
        self.cur_z = z
        md_in = self.arch.get("md_inputs", BOTH)                      # rmt:822-829
        zin = torch.cat(([obs[..., :Db]] if "body" in md_in else []) + ([z] if "task" in md_in else []), dim=-1)
        a_hat = self._motor_decoder(zin)                              # rmt:822-831
        if self.arch.get("mh"):                                       # rmt:833-835
            a_hat = a_hat + self.arch["mh_range"] * self._motor_decoder_helper(zin)
        logits = torch.cat([a_hat, torch.full_like(a_hat, self.log_std)], dim=-1)
        self.cur_future_state = self.forward_world(obs, logits)       # rmt:758
        self.cur_value = self._value_branch(obs).squeeze(1)           # rmt:760-769
        if self.trace is not None:
            lv = self.cur_logvar if self.cur_logvar is not None else torch.zeros_like(self.cur_mu)
            self.trace.append(dict(mu=self.cur_mu.detach(), logvar=lv.detach(), z=z.detach(),
                                   a_hat=a_hat.detach(), future_state=self.cur_future_state.detach()))
        return logits

    # rmt:839-844
    def forward_world(self, obs, logits):
        Db, Da = self.arch["Db"], self.arch["Da"]
        return self._world_model(torch.cat([obs[..., :Db], logits[..., :Da]], dim=-1))

    def set_learnable(self, net, flag):                               # rmt:930-950
        for p in getattr(self, net).parameters():
            p.requires_grad = flag


DEFAULT_COEFFS = dict(vae_kl_coeff=1.0, a_rec_coeff=1.0, s_rec_coeff=0.0, vae_cycle_coeff=1e-3)


def phase_coeffs(world, cfg=None):
    """tpv:331-335: world phase = (0, 0, 1, 0); joint = config values (tpv:282-285)."""
    cfg = dict(DEFAULT_COEFFS, **(cfg or {}))
    if world:
        return dict(vae_kl_coeff=0.0, a_rec_coeff=0.0, s_rec_coeff=1.0, vae_cycle_coeff=0.0)
    return cfg


def compute_loss(model, x, y, coeffs, loss="MSE"):
    """tpv:361-435; `loss` selects self.loss_fn (tm:97-107, 126: "MSE" | "L1" | "MAE").  x [B,L,2Db], y [B,L,Da], L = lookahead.  Returns (total, terms).
    The full forward always runs (tpv:378), including in the world phase.  For L > 1 the
    state fed to step t+1 is the model's own prediction `_cur_future_state` of step t
    (tpv:421, not detached: the gradient flows back through the world model, the motor
    decoder and the task encoder of every earlier step), and each term is the mean over
    the L steps (tpv:423-428)."""
    Db = model.arch["Db"]
    Da = model.arch["Da"]
    L = x.shape[1]
    mse = nn.MSELoss() if loss == "MSE" else nn.L1Loss()
    zero = torch.zeros((), dtype=torch.float32)
    loss_a = loss_kl = loss_s = loss_cyc = zero
    s1 = x[:, 0, :Db]                                                 # tpv:365
    for t in range(L):
        x_t, y_t = x[:, t, :], y[:, t, :]                             # tpv:369-370 (B > 1)
        s2 = x_t[:, Db:]                                              # tpv:375
        logits = model(torch.cat([s1, s2], dim=-1))                   # tpv:377-378
        a_hat = logits[:, :Da]                                        # tpv:356-359
        if coeffs["a_rec_coeff"] > 0.0:
            loss_a = loss_a + mse(y_t, a_hat)                         # tpv:381-382
            if coeffs["vae_kl_coeff"] > 0.0:
                mu, lv = model.cur_mu, model.cur_logvar
                prior = getattr(model, "prior", PRIORS[0])
                if prior is False:                                    # tpv:384: `if self.latent_prior_type and ...`
                    pass
                elif prior == PRIORS[0]:                              # tpv:385-389
                    loss_kl = loss_kl + torch.mean(-0.5 * torch.sum(1 + lv - mu.pow(2) - lv.exp(), dim=1), dim=0)
                elif prior == PRIORS[1]:                              # tpv:390-403 as specified above
                    d = mu - model.cur_prior_mu
                    loss_kl = loss_kl + torch.mean(0.5 * torch.sum(lv.exp() + d.pow(2) - 1 - lv, dim=1), dim=0)
                else:                                                 # tpv:404-407
                    loss_kl = loss_kl + (mu * model.cur_prior_mu).sum(-1).mean()
        if coeffs["s_rec_coeff"] > 0:
            s2_gt_act = model.forward_world(s1, y_t)                  # tpv:411-414
            loss_s = loss_s + mse(s2, s2_gt_act)
        if coeffs["vae_cycle_coeff"] > 0:
            loss_cyc = loss_cyc + mse(s2, model.cur_future_state)     # tpv:417-419
        s1 = model.cur_future_state                                   # tpv:421
    if L > 1:                                                         # tpv:423-428
        n = float(L)
        loss_a, loss_kl, loss_s, loss_cyc = loss_a / n, loss_kl / n, loss_s / n, loss_cyc / n
    total = (coeffs["a_rec_coeff"] * loss_a + coeffs["vae_kl_coeff"] * loss_kl +
             coeffs["s_rec_coeff"] * loss_s + coeffs["vae_cycle_coeff"] * loss_cyc)
    return total, dict(loss_a=loss_a, loss_kl=loss_kl, loss_s=loss_s, loss_cyc=loss_cyc)


def _eps_feeder(eps):
    """eps [B,Z] (one forward) or [L,B,Z] (one slice per forward call, in call order)."""
    if eps is None:
        return None
    if eps.dim() == 2:
        return lambda shape: eps
    it = iter(eps)
    return lambda shape: next(it)


def loss_and_grads(arch, sd, x, y, eps, world, coeff_cfg=None, loss="MSE"):
    """One minibatch through the restated graph; returns forward internals, loss terms and
    every trainable gradient (frozen nets get no gradient, tpv:326-329, 347-350)."""
    model = RefModel(arch)
    model.load_state_dict(sd)
    model.train()
    model.eps_source = _eps_feeder(eps)
    model.set_learnable("_task_encoder", not world)
    model.set_learnable("_motor_decoder", not world)
    model.set_learnable("_world_model", world)
    if hasattr(model, "_latent_prior"):
        model.set_learnable("_latent_prior", not world)
    coeffs = phase_coeffs(world, coeff_cfg)
    model.trace = []
    total, terms = compute_loss(model, x, y, coeffs, loss)
    total.backward()
    grads = OrderedDict((k, p.grad.detach().clone()) for k, p in model.named_parameters()
                        if p.grad is not None)
    Da = arch["Da"]
    lv = model.cur_logvar if model.cur_logvar is not None else torch.zeros_like(model.cur_mu)
    out = dict(total=total.detach(), mu=model.cur_mu.detach(), logvar=lv.detach(),
               z=model.cur_z.detach(), future_state=model.cur_future_state.detach(),
               grads=grads, steps=model.trace)
    if getattr(model, "cur_prior_mu", None) is not None:
        out["prior_mu"] = model.cur_prior_mu.detach()
    model.trace = None
    out.update({k: v.detach() for k, v in terms.items()})
    with torch.no_grad():
        Db = arch["Db"]
        x0 = x[:, 0, :]
        out["a_hat"] = out["steps"][-1]["a_hat"]
        out["s2_from_gt_action"] = model.forward_world(x0, torch.cat([y[:, 0, :], y[:, 0, :]], -1))
    return out


def relu_kink_margin(arch, sd, x, y, eps, world):
    """Per-sample min |pre-activation| over every hidden ReLU that the phase's backward pass
    goes through.  ReLU is not differentiable at 0: a unit whose pre-activation is within
    fp32 rounding of zero lands on either side depending on summation order, which flips one
    sample's gradient path in ANY two fp32 implementations (MKL vs MFMA, CPU vs GPU).  Parity
    tests drop such samples (margin below a few ulps of the activation scale) before comparing
    gradients tightly."""
    keys = {"_task_encoder": "te", "_motor_decoder": "md", "_world_model": "wm"}
    if not any(a == "relu" for k in keys.values() for a in stack_acts(arch, k)):
        return torch.full((x.shape[0],), float("inf"))     # tanh / sigmoid / elu are differentiable everywhere
    model = RefModel(arch)
    model.load_state_dict(sd)
    margins = []

    def hook(mod, inp, out):
        margins.append(out.detach().abs().min(dim=1).values)

    L = x.shape[1]
    model.eps_source = _eps_feeder(eps)
    nets = (["_world_model"] if world and L == 1 else
            ["_task_encoder", "_motor_decoder", "_world_model"])
    hs = []
    for net in nets:
        for slim, a in zip(list(getattr(model, net)._model)[:-1], stack_acts(arch, keys[net])):   # hidden ReLU layers only
            if a == "relu":
                hs.append(slim._model[0].register_forward_hook(hook))
    with torch.no_grad():
        x0 = x[:, 0, :]
        if world and L == 1:
            model.forward_world(x0, torch.cat([y[:, 0, :], y[:, 0, :]], -1))
        elif L == 1:
            model(x0)
        else:                                   # every step's every stack carries gradient
            compute_loss(model, x, y, phase_coeffs(world))
    for h in hs:
        h.remove()
    if not margins:                                # no ReLU on the path of this phase
        return torch.full((x.shape[0],), float("inf"))
    return torch.stack(margins).min(dim=0).values


def adam_reference_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor update, amsgrad False, weight_decay 0 (tm:119-122):
    m <- lerp(m, g, 1-b1); v <- b2 v + (1-b2) g^2;
    p <- p - (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps).  `step` is the 1-based count."""
    m = m + (g - m) * (1.0 - beta1)
    v = v * beta2 + g * g * (1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
    p = p - (lr / bc1) * (m / (v.sqrt() / bc2_sqrt + eps))
    return p, m, v


def lr_for_epoch(epoch_1based, lr0=5e-4, step_size=50, gamma=0.7):
    """StepLR ticking once per epoch (tm:158-159): lr during epoch e = lr0 * gamma^floor((e-1)/step)."""
    return lr0 * gamma ** ((epoch_1based - 1) // step_size)


# --------------------------------------------------------------------------------------
# the whole loop, as the reference runs it (this is also the timed CPU baseline)
# --------------------------------------------------------------------------------------
class RefTrainer:
    """tm:109-161 + tpv:313-351 restated: Adam over *all* parameters (frozen ones are
    skipped because their .grad stays None), StepLR per epoch, phase
    flip when `iter == max_iter_world_model` is seen *before* the increment."""

    def __init__(self, arch, sd, X, Y, batch_size, max_iter_world_model, lr=5e-4,
                 lr_step=50, lr_gamma=0.7, coeff_cfg=None, eps_fn=None, loss="MSE", weight_decay=0.0, shuffle=None):
        self.arch = arch
        self.loss = loss
        self.model = RefModel(arch)
        self.model.load_state_dict(sd)
        self.loader = make_loader(X, Y, batch_size, shuffle)
        self.opt = torch.optim.Adam(self.model.parameters(), lr=lr, weight_decay=weight_decay)   # tm:119-122
        self.sched = torch.optim.lr_scheduler.StepLR(self.opt, step_size=lr_step, gamma=lr_gamma)
        self.max_iter_world_model = max_iter_world_model
        self.coeff_cfg = coeff_cfg
        self.iter = 0
        self.global_batch = 0
        self.eps_calls = 0                      # one draw per model forward = lookahead per minibatch
        self.eps_fn = eps_fn
        self.world = True
        self._apply_phase()

    def _apply_phase(self):
        self.model.set_learnable("_task_encoder", not self.world)
        self.model.set_learnable("_motor_decoder", not self.world)
        self.model.set_learnable("_world_model", self.world)
        if hasattr(self.model, "_latent_prior"):                  # trains with the encoder (spec above)
            self.model.set_learnable("_latent_prior", not self.world)
        self.coeffs = phase_coeffs(self.world, self.coeff_cfg)

    def _next_eps(self, shape):
        e = self.eps_fn(self.eps_calls, shape)
        self.eps_calls += 1
        return e

    def step(self, max_batches=None):
        if self.iter == self.max_iter_world_model:
            self.world = False
            self._apply_phase()
        self.iter += 1
        self.model.train()
        acc, n = 0.0, 0
        tacc = dict(loss_a=0.0, loss_kl=0.0, loss_s=0.0, loss_cyc=0.0)
        for x, y in self.loader:
            if self.eps_fn is not None:
                self.model.eps_source = self._next_eps
            self.opt.zero_grad()
            loss, terms = compute_loss(self.model, x, y, self.coeffs, self.loss)
            loss.backward()
            self.opt.step()
            acc += loss.item()
            for k in tacc:                        # (test bookkeeping: per-term epoch means)
                tacc[k] += float(terms[k].detach()) if torch.is_tensor(terms[k]) else float(terms[k])
            n += 1
            self.global_batch += 1
            if max_batches is not None and n >= max_batches:
                break
        self.sched.step()
        self.last_terms = {k: v / max(n, 1) for k, v in tacc.items()}
        return {"mean_train_loss": acc / (len(self.loader) if max_batches is None else n),
                "mean_test_loss": 0.0}


def eps_stream(seed, latent):
    """Deterministic epsilon per *forward call index* (one randn_like per model forward,
    SURVEY.md 3.2; call = global minibatch * lookahead + t): both the reference capture and
    the HIP path consume eps(call)."""
    def fn(global_batch, shape):
        rng = np.random.default_rng([seed, int(global_batch)])
        e = rng.standard_normal((int(shape[0]), latent)).astype(np.float32)
        return torch.from_numpy(e)
    return fn


def time_cpu_baseline(arch, sd, X, Y, batch_size, world, n_batches, warmup=2, threads=None):
    """Timed leg for bench.py's `cpu_baseline`: the reference's op sequence (per-sample
    Dataset, collate, full forward incl. value branch and the world-phase extra forward,
    autograd, torch.optim.Adam, per-batch .item()) on the host cores."""
    if threads:
        torch.set_num_threads(threads)
    tr = RefTrainer(arch, sd, X, Y, batch_size, max_iter_world_model=(10 ** 9 if world else 0))
    tr.step(max_batches=warmup)
    t0 = time.perf_counter()
    tr.step(max_batches=n_batches)
    dt = time.perf_counter() - t0
    done = min(n_batches, len(tr.loader))
    samples = min(done * batch_size, len(X))
    return dict(samples_per_s=samples / dt, seconds=dt, batches=done, samples=samples,
                threads=torch.get_num_threads())


# --------------------------------------------------------------------------------------
# checkpoint layout (tpv:440-467, tm:209-213, rmt:870-928)
# --------------------------------------------------------------------------------------
def checkpoint_files(model_sd):
    """Returns {filename: object} exactly as the reference writes them."""
    def sub(net):
        pre = net + "."
        return OrderedDict((k[len(pre):], v) for k, v in model_sd.items() if k.startswith(pre))
    return OrderedDict([
        ("model.pth", OrderedDict(model_sd)),
        ("model.pt", OrderedDict(model_sd)),
        ("task_encoder.pt", {"task_encoder": sub("_task_encoder")}),
        ("motor_decoder.pt", sub("_motor_decoder")),
        ("world_model.pt", sub("_world_model")),
    ])


def tensor_digest(t, n_samples=16, seed=12345):
    """Small fingerprint of a big tensor for fixtures: sum, abs-sum, L2, fixed samples."""
    a = t.detach().double().reshape(-1).numpy()
    idx = np.random.default_rng(seed).integers(0, a.size, size=n_samples)
    return np.concatenate([[a.sum(), np.abs(a).sum(), math.sqrt(float((a * a).sum()))], a[idx]])
