"""PhysicsVAE module surface (reference: rllib_model_torch.py `PhysicsVAE`, rmt:461-950) on
top of the HIP engine.

What is kept from the reference so that checkpoints and callers drop in unchanged:
  * constructor signature (obs_space, action_space, num_outputs, model_config, name)
  * sub-module names -> state_dict keys `<net>._model.<i>._model.0.{weight,bias}` with
    shapes [n_out, n_in] / [n_out] (rmt:234-283 + ray SlimFC)
  * forward(input_dict, state, seq_lens) -> (logits[B, 2*Da], state), value_function(),
    forward_encoder/decoder/world/value_branch, task_encoder_variable(),
    set_exploration_std, save_/load_weights*, set_learnable_*

What is different by design: the task-encoder, motor-decoder and world-model parameters
are strided views into ONE flat device arena owned by `HipEngine` (include/pvae.h), the
layer arithmetic runs in hand-written gfx950 kernels, and no ray/gym import is needed.
The value branch (rmt:693-699) never enters the training loss (tpv:356-435) and is not
part of the hot path: it is kept as ordinary torch parameters so the 26-tensor checkpoint
layout is complete, and evaluated with torch ops when a caller asks for value_function().
"""
import copy
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from ._lib import NET_MD, NET_MH, NET_NAMES, NET_PR, NET_TE, NET_WM, PRIOR_KINDS
from .engine import Arch, HipEngine, Stack


def fc_spec(width, depth, out_size="output", act_hidden="relu", act_out="linear"):
    """Layer-spec list in the reference's dict format (what gen_layers emits, tpv:180-192)."""
    assert depth > 0 and width > 0
    hidden = {"type": "fc", "hidden_size": width, "activation": act_hidden,
              "init_weight": {"name": "normc", "std": 1.0}}
    last = {"type": "fc", "hidden_size": out_size, "activation": act_out,
            "init_weight": {"name": "normc", "std": 0.01}}
    return [dict(hidden) for _ in range(depth)] + [last]


DEFAULT_FC_256X2 = fc_spec(256, 2)
DEFAULT_FC_512X3 = fc_spec(512, 3)
DEFAULT_FC_1024X2 = fc_spec(1024, 2)


def normc_(tensor, std=1.0):
    """ray 1.11.0 `normc_initializer` semantics: N(0,1) then every output
    row rescaled to L2 norm `std`.  Works in place on (strided) views."""
    with torch.no_grad():
        w = torch.randn(tensor.shape, dtype=torch.float32)
        w *= std / torch.sqrt(w.pow(2).sum(1, keepdim=True))
        tensor.copy_(w)
    return tensor


ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "elu": nn.ELU}     # rmt:30-46 minus swish


def get_initializer(info):
    """rmt:220-232: "normc" (std), "xavier_normal" / "xavier_uniform" (gain); in place on a (strided) view."""
    if info["name"] == "normc":
        return lambda t: normc_(t, info["std"])
    if info["name"] in ("xavier_normal", "xavier_uniform"):
        fn = getattr(nn.init, info["name"] + "_")

        def init(t):
            with torch.no_grad():
                t.copy_(fn(torch.empty(t.shape, dtype=torch.float32), gain=info["gain"]))
            return t
        return init
    raise NotImplementedError(info["name"])


def _fc_stack(layers, what):
    """A `*_layers` list of the model config (rmt:462-510) as the HIP path takes it: fc layers only -- each with
    its own integer width, its own activation out of relu / tanh / sigmoid / elu / linear and its own init_weight
    (FC.__init__, rmt:234-270) -- then a linear fc output layer.  `gen_layers` (tpv:180-192) emits the special case
    of one width and one activation.  Anything else (bn / softmax / hardmax layers, swish, a non-linear output) is
    refused loudly.  Returns (Stack of the hidden layers, [init_weight dict per Linear])."""
    if not layers or any(l.get("type") != "fc" for l in layers):
        raise NotImplementedError("%s: only 'fc' layers are supported on the HIP path" % what)
    *hidden, last = layers
    if not hidden:
        raise NotImplementedError("%s: at least one hidden layer is required" % what)
    if any(not isinstance(l["hidden_size"], int) or isinstance(l["hidden_size"], bool) for l in hidden):
        raise NotImplementedError("%s: hidden layers need integer widths, got %s" % (what, [l["hidden_size"] for l in hidden]))
    acts = ["linear" if l.get("activation") is None else l.get("activation") for l in hidden]
    if any(a not in ACTIVATIONS and a != "linear" for a in acts) or last.get("activation") not in ("linear", None):
        raise NotImplementedError("%s: hidden activations out of %s and a linear output layer, got %s -> %s"
                                  % (what, sorted(ACTIVATIONS) + ["linear"], acts, last.get("activation")))
    if last["hidden_size"] != "output":
        raise NotImplementedError("%s: last layer must have hidden_size 'output'" % what)
    default_init = [{"name": "normc", "std": 1.0}] * len(hidden) + [{"name": "normc", "std": 0.01}]
    inits = [l.get("init_weight") or d for l, d in zip(layers, default_init)]
    for info in inits:
        get_initializer(info)                      # (unknown names fail here, not half-way through construction)
    return Stack([l["hidden_size"] for l in hidden], acts), inits


def _helper_stack(layers, rng):
    """`motor_decoder_helper_layers` (rmt:491-495): fc layers, the output layer ending in tanh (asserted upstream,
    rmt:672-673, as is a positive range).  Returns (widths, hidden activations, [init_weight dict per Linear])."""
    assert layers[-1]["activation"] == "tanh"
    assert rng > 0
    if any(l.get("type") != "fc" for l in layers) or layers[-1]["hidden_size"] != "output" or len(layers) < 2:
        raise NotImplementedError("motor_decoder_helper_layers: fc hidden layers and an fc output layer")
    hidden = layers[:-1]
    acts = ["linear" if l.get("activation") is None else l["activation"] for l in hidden]
    if any(a not in ACTIVATIONS and a != "linear" for a in acts):
        raise NotImplementedError("motor_decoder_helper_layers: hidden activations out of %s" % (sorted(ACTIVATIONS) + ["linear"]))
    inits = [l.get("init_weight") or {"name": "normc", "std": 0.01 if l is layers[-1] else 1.0} for l in layers]
    for info in inits:
        get_initializer(info)
    return [int(l["hidden_size"]) for l in hidden], acts, inits


class SlimFC(nn.Module):
    """Linear (+activation) held as `self._model = nn.Sequential(...)` -- the ray SlimFC shape that
    gives the `._model.0.weight` key suffix."""

    def __init__(self, in_size, out_size, init, weight=None, bias=None, act=None):
        super().__init__()
        lin = nn.Linear(in_size, out_size)
        if weight is not None:                 # alias the engine arena instead of own storage
            lin.weight = nn.Parameter(weight)
            lin.bias = nn.Parameter(bias)
        get_initializer(init)(lin.weight.data)
        with torch.no_grad():
            lin.bias.zero_()
        self._model = nn.Sequential(*([lin, ACTIVATIONS[act]()] if act not in (None, "linear") else [lin]))

    def forward(self, x):
        return self._model(x)


class AppendLogStd(nn.Module):
    """rmt:160-206.  "constant": log_std is a plain tensor (NOT a parameter or buffer, so it is absent
    from state_dict) appended to the decoder output; "state_independent": an nn.Parameter (key
    `..._model.<n>.log_std`).  The supervised loss reads the action half of the logits only
    (tpv:356-359), so the parameter never receives a gradient there -- upstream or here."""

    def __init__(self, init_val, dim, type="constant", device=None):
        super().__init__()
        self.type = type
        val = torch.full((dim,), float(init_val), dtype=torch.float32)
        if type == "constant":
            self.log_std = val
        elif type == "state_independent":
            self.log_std = nn.Parameter(val.to(device) if device is not None else val)
        else:
            raise NotImplementedError(type)

    def set_val(self, val):
        assert self.type == "constant", "Change value is only allowed in constant logstd"
        assert np.isscalar(val), "Only scalar is currently supported"
        self.log_std[:] = float(val)
        self._dev = None
        self._ver = getattr(self, "_ver", 0) + 1

    def on_device(self, device):
        """The vector as a device tensor for the inference launches (a constant log_std lives on the host, as
        upstream; its device copy is refreshed whenever set_val changes it)."""
        if self.type != "constant":
            return self.log_std.detach()
        if getattr(self, "_dev", None) is None or self._dev.device != device:
            self._dev = self.log_std.to(device)
        return self._dev

    def forward(self, x):
        ls = self.log_std.to(x.device).reshape([1] * (x.dim() - 1) + [-1])
        return torch.cat([x, ls.expand(*x.shape[:-1], -1)], dim=-1)


class FC(nn.Module):
    """rmt:234-283: `self._model = nn.Sequential(SlimFC..., [AppendLogStd])`."""

    def __init__(self, dims, views=None, append_log_std=False, sample_std=1.0, log_std_type="constant",
                 device=None, act="relu", inits=None, out_act=None):
        """`act`: one name for every hidden layer, or one per hidden layer; `inits`: init_weight dict per Linear
        (default: gen_layers' normc 1.0 / 0.01 for the output layer, tpv:184-189); `out_act`: activation of the output
        layer (the motor decoder's helper ends in tanh, rmt:491-495)."""
        super().__init__()
        mods = []
        acts = [act] * (len(dims) - 1) if isinstance(act, str) else list(act)
        for i, (n_in, n_out) in enumerate(dims):
            last = i == len(dims) - 1
            w, b = (views[i] if views is not None else (None, None))
            init = inits[i] if inits is not None else {"name": "normc", "std": 0.01 if last else 1.0}
            mods.append(SlimFC(n_in, n_out, init, weight=w, bias=b, act=out_act if last else acts[i]))
        if append_log_std:
            mods.append(AppendLogStd(math.log(sample_std), dims[-1][1], type=log_std_type, device=device))
        self._model = nn.Sequential(*mods)

    def forward(self, x):
        return self._model(x)

    def save_weights(self, file):
        torch.save(_portable(self.state_dict()), file)

    def load_weights(self, file):
        self.load_state_dict(torch.load(file, map_location="cpu"))
        self.eval()


def _portable(sd):
    """Contiguous CPU copies: checkpoints must not carry the arena's strides or device
    (the reference saves whatever device the model is on, rmt:870-871)."""
    return OrderedDict((k, v.detach().to("cpu").contiguous().clone()) for k, v in sd.items())


class _Cur:
    """What the last forward left behind (`_cur_*` of rmt:742-771 and the bookkeeping of the lazy read-backs).
    Kept in a plain object: every attribute write on an nn.Module goes through Module.__setattr__, microseconds
    each, and the rollout forward sets eight of them per call."""
    __slots__ = ("_cur_value", "_lazy", "_mu", "_logvar", "_cur_task_encoder_variable", "_cur_body_encoder_variable",
                 "_cur_latent_prior_mu", "_cur_latent_prior_logvar", "_cur_future_state", "_rng_calls")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)
        self._rng_calls = 0


def _cur_property(name):
    return property(lambda self: getattr(self._st, name), lambda self, v: setattr(self._st, name, v))


class PhysicsVAE(nn.Module):
    DEFAULT_CONFIG = {
        "project_dir": None,
        "log_std_type": "constant",
        "sample_std": 0.1,
        "load_weights": None,
        "task_encoder_inputs": ["body", "task"],
        "task_encoder_layers": DEFAULT_FC_256X2,
        "task_encoder_load_weights": None,
        "task_encoder_learnable": True,
        "task_encoder_output_dim": 32,
        "latent_prior_type": "normal_zero_mean_one_std",
        "latent_prior_layers": None,
        "motor_decoder_inputs": ["body", "task"],
        "motor_decoder_layers": DEFAULT_FC_512X3,
        "motor_decoder_load_weights": None,
        "motor_decoder_learnable": True,
        "motor_decoder_helper_enable": False,                              # rmt:490-498
        "motor_decoder_helper_layers": fc_spec(128, 2, act_out="tanh"),
        "motor_decoder_helper_load_weights": None,
        "motor_decoder_helper_learnable": True,
        "motor_decoder_helper_range": 0.5,
        "value_fn_layers": DEFAULT_FC_256X2,
        "world_model_layers": DEFAULT_FC_1024X2,
        "world_model_load_weights": None,
        "world_model_learnable": True,
        "observation_space": None,
        "observation_space_body": None,
        "observation_space_task": None,
        "action_space": None,
        # ours: where the arena lives and how many rows one call may carry
        "device": None,
        "max_batch": 256,
        "lookahead": 1,
    }

    def __init__(self, obs_space, action_space, num_outputs, model_config, name, **kwargs):
        super().__init__()
        self.obs_space, self.action_space = obs_space, action_space
        self.model_config, self.name = model_config, name
        assert num_outputs % 2 == 0, ("num_outputs must be divisible by two", num_outputs)
        self.num_outputs = num_outputs
        cfg = copy.deepcopy(PhysicsVAE.DEFAULT_CONFIG)
        cfg.update(model_config.get("custom_model_config") or {})
        if cfg["log_std_type"] not in ("constant", "state_independent"):
            raise NotImplementedError(cfg["log_std_type"])                      # rmt:182-183
        if cfg["latent_prior_type"] not in PRIOR_KINDS:
            raise NotImplementedError("Unknown latent_prior_type:%s" % (cfg["latent_prior_type"],))    # rmt:624-625
        # rmt:470, 485: any non-empty subset of ["body", "task"] -- a column window of the full-width first layer (engine.Arch)
        self._task_encoder_inputs = list(cfg["task_encoder_inputs"])
        self._motor_decoder_inputs = list(cfg["motor_decoder_inputs"])

        self.dim_state_body = int(np.prod(cfg["observation_space_body"].shape))
        self.dim_state_task = int(np.prod(cfg["observation_space_task"].shape))
        self.dim_state = int(np.prod(obs_space.shape))
        self.dim_action = int(np.prod(action_space.shape))
        assert self.dim_state == self.dim_state_body + self.dim_state_task
        assert self.dim_state_task == self.dim_state_body, "PhysicsVAE training uses s_task = s_body(t+1)"
        assert num_outputs // 2 == self.dim_action
        Z = int(cfg["task_encoder_output_dim"])
        self._task_encoder_output_dim = Z
        # "normal_state_mean_one_std" / "hypersphere_uniform": the reference sketches them and crashes
        # (rmt:632, tpv:395-396, 406); built here to the specification in oracle/refpath.py (PRIORS)
        self._latent_prior_type = cfg["latent_prior_type"]

        te, te_init = _fc_stack(cfg["task_encoder_layers"], "task_encoder_layers")
        md, md_init = _fc_stack(cfg["motor_decoder_layers"], "motor_decoder_layers")
        wm, wm_init = _fc_stack(cfg["world_model_layers"], "world_model_layers")
        vb, vb_init = _fc_stack(cfg["value_fn_layers"], "value_fn_layers")
        learned_prior = self._latent_prior_type == "normal_state_mean_one_std"
        pr, pr_init = _fc_stack(cfg.get("latent_prior_layers") or cfg["task_encoder_layers"], "latent_prior_layers")
        # the default activation of the ctx: the trainer's one "act_fn" when every layer agrees (tpv:290-311);
        # stacks that differ from it travel layer by layer (pvae_config.layer_width / layer_act)
        used = {a for st in ((te, md, wm, pr) if learned_prior else (te, md, wm)) for a in st.acts}
        act = next(iter(used)) if len(used) == 1 and next(iter(used)) != "linear" else "relu"
        # The helper (rmt:670-680, 833-835): a residual policy on the decoder's input whose tanh output, scaled by
        # `motor_decoder_helper_range`, is added to the action.  A fifth stack of the arena (PVAE_NET_MH): upstream's
        # supervised loss sees its term inside a_hat, so the joint phase trains it with the decoder.
        self._motor_decoder_helper_range = cfg.get("motor_decoder_helper_range")
        mh = mh_init = None
        if cfg.get("motor_decoder_helper_enable"):
            widths, acts, mh_init = _helper_stack(cfg["motor_decoder_helper_layers"], self._motor_decoder_helper_range)
            mh = Stack(widths, acts)
        self.arch = Arch(self.dim_state_body, self.dim_action, Z, te, md, wm, prior=self._latent_prior_type,
                         pr=pr, act=act, te_inputs=self._task_encoder_inputs, md_inputs=self._motor_decoder_inputs,
                         mh=mh, mh_range=self._motor_decoder_helper_range if mh is not None else 0.5)
        device = cfg["device"] or ("cuda" if torch.cuda.is_available() else "cpu")
        self.engine = HipEngine(self.arch, int(cfg["max_batch"]), device=device,
                                lookahead=int(cfg.get("lookahead", 1) or 1))

        views = self.engine.named_views()
        per_net = {n: [] for n in (NET_TE, NET_MD, NET_WM, NET_PR, NET_MH)}
        for info in self.engine.layers:
            base = "%s._model.%d._model.0." % (NET_NAMES[info["net"]], info["index"])
            per_net[info["net"]].append(((info["n_in"], info["n_out"]),
                                         (views[base + "weight"], views[base + "bias"])))

        stacks = {NET_TE: (te, te_init), NET_MD: (md, md_init), NET_WM: (wm, wm_init), NET_PR: (pr, pr_init), NET_MH: (mh, mh_init)}

        def build(net, **kw):
            dims = [d for d, _ in per_net[net]]
            vws = [v for _, v in per_net[net]]
            return FC(dims, views=vws, act=stacks[net][0].acts, inits=stacks[net][1], **kw)

        # registration order fixes the state_dict order: [prior,] TE, MD, WM, VB (rmt:627-699)
        self._latent_prior = build(NET_PR) if learned_prior else None
        self._task_encoder = build(NET_TE)
        self._motor_decoder = build(NET_MD, append_log_std=True, sample_std=cfg["sample_std"],
                                    log_std_type=cfg["log_std_type"], device=self.engine.device)
        # (registered between the motor decoder and the world model, as upstream: the state-dict order)
        self._motor_decoder_helper = None
        if mh is not None:
            self._motor_decoder_helper = build(NET_MH, out_act="tanh")
            self.__dict__["_mh_acts"] = list(mh.acts)
        self._world_model = build(NET_WM)
        self.__dict__["_als"] = self._motor_decoder._model[-1]      # (a plain reference: module lookups cost microseconds per forward)
        vb_dims, prev = [], self.dim_state
        for width in vb.widths:
            vb_dims.append((prev, width))
            prev = width
        vb_dims.append((prev, 1))
        self._value_branch = FC(vb_dims, act=vb.acts, inits=vb_init).to(self.engine.device)
        self._vb_act = vb.acts

        self._st = _Cur()
        self._cur_value = None
        self._lazy, self._mu, self._logvar = None, None, None
        self._cur_task_encoder_variable = None
        self._cur_body_encoder_variable = None
        self._cur_task_encoder_mu = None
        self._cur_task_encoder_logvar = None
        self._cur_latent_prior_mu = None
        self._cur_latent_prior_logvar = None
        self._cur_future_state = None
        self.latent_prior_noise = True
        self._rng_seed, self._rng_calls = 0, 0

        def rooted(path):
            import os
            return os.path.join(cfg["project_dir"], path) if cfg.get("project_dir") else path

        if cfg.get("load_weights"):
            self.load_weights(rooted(cfg["load_weights"]))
        if cfg.get("task_encoder_load_weights"):
            self.load_weights_task_encoder(rooted(cfg["task_encoder_load_weights"]))
            self.set_learnable_task_encoder(cfg["task_encoder_learnable"])
        if cfg.get("motor_decoder_load_weights"):
            self.load_weights_motor_decoder(rooted(cfg["motor_decoder_load_weights"]))
            self.set_learnable_motor_decoder(cfg["motor_decoder_learnable"])
        if cfg.get("motor_decoder_helper_load_weights"):                                    # rmt:721-723
            self.load_weights_motor_decoder_helper(rooted(cfg["motor_decoder_helper_load_weights"]))
            self.set_learnable_motor_decoder_helper(cfg["motor_decoder_helper_learnable"])
        if cfg.get("world_model_load_weights"):
            self.load_weights_world_model(rooted(cfg["world_model_load_weights"]))
            self.set_learnable_world_model(cfg["world_model_learnable"])

    # -- nn.Module plumbing ---------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        """`.to()/.cuda()/.float()` would re-allocate the arena-backed parameters and break
        the aliasing; the device is chosen at construction (custom_model_config['device'])."""
        want = self.engine.device
        probe = fn(torch.zeros(1, device=want))

        def same_device(a, b):                     # 'cuda' == 'cuda:<current>'
            if a.type != b.type:
                return False
            if a.type != "cuda":
                return True
            cur = torch.cuda.current_device() if torch.cuda.is_available() else 0
            return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)
        if not same_device(probe.device, want) or probe.dtype != torch.float32:
            raise RuntimeError("PhysicsVAE lives on %s/float32 (chosen at construction); "
                               "rebuild it with custom_model_config['device'] instead of .to()"
                               % self.engine.device)
        return self

    def get_initial_state(self):
        return []

    def seed(self, seed):
        """Key of the on-chip Philox stream used when no eps is supplied."""
        self._rng_seed, self._rng_calls = int(seed), 0

    # -- forward (rmt:742-853), all contractions on the HIP kernels ---------------------------
    def __call__(self, input_dict, state=None, seq_lens=None):
        # ModelV2.__call__ semantics: obs_flat = obs, then forward
        d = dict(input_dict)
        d["obs_flat"] = d["obs"] if "obs" in d else d["obs_flat"]
        out, st = self.forward(d, state or [], seq_lens)
        return out, st

    # The 30 Hz control loop calls forward at B = 1 (envs/rllib_env_imitation.py:215-266) and reads the
    # action only.  "lazy" (default): the world model's prediction `_cur_future_state` (rmt:758) is computed
    # when somebody reads it -- 3 launches less per forward; True: with every forward, as upstream; False: never.
    rollout_predicts_state = "lazy"

    for _name in _Cur.__slots__:                      # `model._cur_future_state` etc. keep working, reads and writes
        locals()[_name] = _cur_property(_name)
    del _name

    def _get_future_state(self):
        st = self._st
        if st._cur_future_state is None and st._lazy is not None and len(st._lazy) > 2 and self._world_model is not None:
            obs, rows, eps, noise, offset = st._lazy        # the same call again, now with the prediction: same
            st._cur_future_state = self.engine.infer(obs, eps=eps, noise=noise, seed=self._rng_seed, offset=offset,
                                                     want_s2=True)[1]           # kernels, same draws -> same action
        return st._cur_future_state

    _cur_future_state = property(_get_future_state, lambda self, v: setattr(self._st, "_cur_future_state", v))

    def forward(self, input_dict, state, seq_lens, eps=None):
        """rmt:742-771.  One library call for the whole chain (TE -> sampler -> MD -> WM, `pvae_infer`).
        The encoder's mu / logvar and the value estimate are produced on demand
        (`_cur_task_encoder_mu`, `value_function()`): the rollout loop reads neither."""
        obs = input_dict["obs_flat"].float()
        eng = self.engine
        if obs.dim() != 2 or obs.shape[0] > eng.max_batch or self._latent_prior is not None or eng.lookahead != 1:
            return self._forward_staged(obs, state, seq_lens, eps)
        rows = obs.shape[0]
        noise = bool(self.latent_prior_noise)
        st = self._st
        if (self.__dict__.get("_srv_on") and rows <= 4 and obs.device.type == "cpu" and (eps is None or not noise)
                and self._latent_prior_type != "hypersphere_uniform"):
            if rows == 1:
                return self._forward_served(obs, state, noise)
            if not self.__dict__.get("_srv_rows_off"):
                try:
                    return self._forward_served_rows(obs, state, noise)
                except RuntimeError as e:
                    # the multi-row instance was refused (its four input vectors do not fit the LDS plan: -24): the single-row
                    # instance stays resident (include/pvae.h) and 2-4-row observations take the launch path from now on
                    if "rollout server" not in str(e):
                        raise
                    self.__dict__["_srv_rows_off"] = True
                    st._rng_calls -= 1                 # (the launch path below draws at the offset the request would have)
        obs = obs.to(eng.device)
        st._rng_calls += 1
        # (eager on purpose: with the input assembly, the sampler and the output copies inside the layer
        #  launches the call is 10 launches, and issue -> result at B = 1 measures 37 us eager against 44 us
        #  for a replay of the same launches as a HIP graph, whose fixed cost is higher; `graphed_infer`
        #  stays available for callers that replay many forwards back to back)
        logits, s2, z = eng.infer_logits(obs, self.__dict__["_als"].on_device(eng.device),
                                         eps=eps if noise else None, noise=noise, seed=self._rng_seed,
                                         offset=st._rng_calls, want_s2=self.rollout_predicts_state is True)
        st._cur_future_state = s2
        st._cur_body_encoder_variable = obs[..., : self.dim_state_body]
        st._cur_task_encoder_variable = z
        # mu / logvar / value (and the prediction, when lazy): computed when somebody asks
        # (supplied draws are copied: the deferred call must see what this one saw even if the caller reuses its buffer;
        #  the rollout loop supplies none -- Philox draws are a function of (seed, offset))
        # The observation a deferred read re-uses is the library's own copy of it (<= 4 rows: written by the first
        # encoder launch, `kept_obs`), or a clone: the caller may recycle its buffer right after this call.
        keep = eng.kept_obs(rows) if rows <= 4 and eng.fused_rollout else obs.clone()
        st._lazy = ((keep, rows, eps.clone() if (noise and eps is not None) else None, noise, st._rng_calls)
                    if self.rollout_predicts_state == "lazy" else (keep, rows))
        st._mu = st._logvar = st._cur_value = None
        st._cur_latent_prior_mu = (eng.read("eps", rows) if self._latent_prior_type == "hypersphere_uniform"
                                   else None)                  # rmt:813-814: the unit prior sample of this forward
        return logits, state

    # -- the call-persistent rollout server (opt-in; include/pvae.h pvae_rollout_server_*) -------------------
    def start_rollout_server(self, idle_ms=100.0, lifetime_s=600.0, scope="auto"):
        """Serve `forward` at B = 1 from the resident rollout kernel (one XCD, encoder + decoder [+ helper] weights in LDS, mailbox
        in pinned host memory): a forward whose observation arrives as a CPU tensor of ONE row then costs no launch and
        no device copy, and returns CPU tensors (the 30 Hz control loop of envs/rllib_env_imitation.py:215-266 hands the
        action to a CPU simulator anyway).  Same action as the launch path, bit for bit; mu / logvar / z come with it;
        the world model's prediction and the value estimate stay lazy (launches, on first read).  The resident copy of
        the weights follows `load_state_dict` / `load_weights*` and the HIP trainer's optimizer steps (`reload_rollout_server()`
        after writes the library cannot see).
        `scope`: "xcd" (one XCD), "chip" (all CUs: stacks too big for one XCD, e.g. 4x1024), "auto".  Raises RuntimeError when
        nothing fits: the launch path stays in use."""
        self.engine.rollout_server_start(idle_ms=idle_ms, lifetime_s=lifetime_s, scope=scope)
        self.__dict__["_srv_on"] = True

    def stop_rollout_server(self):
        self.__dict__["_srv_on"] = False
        self.engine.rollout_server_stop()

    def reload_rollout_server(self):
        """The parameters were written outside the library (a torch optimizer, a direct write into a parameter): the next
        served request re-reads them into LDS (`pvae_params_changed`).  Optimizer steps through the HIP trainer and
        `load_state_dict` / `load_weights*` are noticed without this call."""
        eng = self.__dict__.get("engine")
        if eng is not None and eng.ctx is not None:
            eng.params_changed(torch.cuda.current_stream(eng.device).cuda_stream)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.reload_rollout_server()                 # (submodule loads -- load_weights_task_encoder etc. -- call it as well)
        return out

    def _forward_served(self, obs, state, noise):
        """One served forward.  The library writes straight into persistent CPU tensors (no per-call allocation; the
        `_cur_*` tensors of this path are valid until the next forward, the returned logits are a fresh copy)."""
        st, d = self._st, self.__dict__
        st._rng_calls += 1
        io = d.get("_srv_t")
        Da, Z = self.dim_action, self._task_encoder_output_dim
        if io is None:
            import ctypes as _C
            t_obs, t_log, t_ml, t_z = torch.zeros(1, 2 * self.dim_state_body), torch.zeros(1, 2 * Da), torch.zeros(1, 2 * Z), torch.zeros(1, Z)
            io = d["_srv_t"] = (t_obs, t_log, t_ml, t_z, t_ml[:, :Z], t_ml[:, Z:], t_obs[:, : self.dim_state_body],
                                tuple(_C.c_void_p(t.data_ptr()) for t in (t_obs, t_log, t_ml, t_z)), self.engine.lib.pvae_rollout_server_infer)
            d["_srv_ls"] = None
        t_obs, t_log, t_ml, t_z, v_mu, v_lv, v_s1, ptrs, fn = io
        als = d["_als"]
        ls = als.log_std
        if d["_srv_ls"] is not ls or als.type != "constant" or d.get("_srv_ls_ver") != getattr(als, "_ver", 0):
            t_log[0, Da:] = ls.detach().cpu().reshape(-1)          # [a_hat | log_std] (AppendLogStd, rmt:160-206)
            d["_srv_ls"], d["_srv_ls_ver"] = ls, getattr(als, "_ver", 0)
        t_obs.copy_(obs)
        rc = fn(self.engine.ctx, ptrs[0], 1 if noise else 0, self._rng_seed, st._rng_calls, 0, ptrs[1], ptrs[2], ptrs[3], 1000.0)
        if rc:
            from . import _lib
            _lib.check(rc, "pvae_rollout_server_infer")
        st._cur_future_state = None
        st._cur_body_encoder_variable = v_s1
        st._cur_task_encoder_variable = t_z
        # the prediction is a launch-path forward on the same observation and the same (seed, offset) draws: deferred to the
        # first read ("lazy"), made with this forward (True: as upstream, rmt:758) or never (False)
        st._lazy = (t_obs, 1, None, noise, st._rng_calls) if self.rollout_predicts_state in ("lazy", True) else (t_obs, 1)
        if self.rollout_predicts_state is True:
            self._get_future_state()
        st._cur_value = None
        if self._latent_prior_type is False:
            st._mu, st._logvar = t_z, None
        else:
            st._mu, st._logvar = v_mu, v_lv
        st._cur_latent_prior_mu = None
        return t_log.clone(), state

    def _forward_served_rows(self, obs, state, noise):
        """2-4 rows in ONE request to the resident kernel (pvae_rollout_server_infer_rows); same contract as `_forward_served`,
        fresh tensors per call."""
        st = self._st
        st._rng_calls += 1
        rows, Z = obs.shape[0], self._task_encoder_output_dim
        a, ml, z = self.engine.rollout_server_infer_rows(obs, noise=noise, seed=self._rng_seed, offset=st._rng_calls)
        ls = self.__dict__["_als"].log_std.detach().cpu().reshape(1, -1).expand(rows, -1)
        logits = torch.cat([torch.from_numpy(a), ls], dim=1)                  # [a_hat | log_std] (AppendLogStd, rmt:160-206)
        t_obs, t_ml, t_z = obs.clone(), torch.from_numpy(ml), torch.from_numpy(z)
        st._cur_future_state = None
        st._cur_body_encoder_variable = t_obs[:, : self.dim_state_body]
        st._cur_task_encoder_variable = t_z
        st._lazy = (t_obs, rows, None, noise, st._rng_calls) if self.rollout_predicts_state in ("lazy", True) else (t_obs, rows)
        if self.rollout_predicts_state is True:
            self._get_future_state()
        st._cur_value = None
        if self._latent_prior_type is False:
            st._mu, st._logvar = t_z, None
        else:
            st._mu, st._logvar = t_ml[:, :Z], t_ml[:, Z:]
        st._cur_latent_prior_mu = None
        return logits, state

    def _forward_staged(self, obs, state, seq_lens, eps=None):
        """The same forward stage by stage (forward_encoder / forward_decoder / forward_world): batches
        beyond the engine's panel size, the learned-prior configuration, lookahead engines."""
        self._lazy = None
        z_body, z_task, _ = self.forward_encoder(obs, state, seq_lens, 0, eps=eps)
        logits, _ = self.forward_decoder(z_body, z_task, state, seq_lens, 0)
        self._cur_future_state = self.forward_world(obs, logits)
        val, _ = self.forward_value_branch(obs, state, seq_lens, 0)
        self._cur_body_encoder_variable = z_body
        self._cur_task_encoder_variable = z_task
        self._cur_value = val.squeeze(1)
        return logits, state

    def _encoder_stat(self, name):
        """mu / logvar of the last fused forward, read back from the engine's panels on first use."""
        if self._lazy is not None and self._mu is None:
            rows = self._lazy[1]
            no_logvar = self._latent_prior_type in ("hypersphere_uniform", False)
            # (no-logvar encoders: mu IS the code this forward returned -- the decoder-input panel the engine's
            #  "z" reads is not written by the <= 4-row rollout path)
            self._mu = self._st._cur_task_encoder_variable if no_logvar else self.engine.read("mu", rows)
            self._logvar = None if no_logvar else self.engine.read("logvar", rows)
        return self._mu if name == "mu" else self._logvar

    @property
    def _cur_task_encoder_mu(self):
        return self._encoder_stat("mu")

    @_cur_task_encoder_mu.setter
    def _cur_task_encoder_mu(self, v):
        self._mu = v

    @property
    def _cur_task_encoder_logvar(self):
        return self._encoder_stat("logvar")

    @_cur_task_encoder_logvar.setter
    def _cur_task_encoder_logvar(self, v):
        self._logvar = v

    def forward_encoder(self, obs, state=None, seq_lens=None, state_cnt=0, eps=None):
        Z = self._task_encoder_output_dim
        obs = obs.to(self.engine.device)
        h = self.engine.net_forward(NET_TE, obs)
        if self._latent_prior_type is False:                        # rmt:815-816: the encoder output is the code
            z_task = self._reparameterize(h, None)                  # (a copy through the sampler kernel, not normalised)
            return obs[..., : self.dim_state_body], z_task, state_cnt
        if self._latent_prior_type == "hypersphere_uniform":        # rmt:810-814 (z = mu: oracle PRIORS)
            z_task = self._reparameterize(h, eps)                   # e / |e|; the unit prior sample lands in "eps"
            self._cur_task_encoder_mu, self._cur_task_encoder_logvar = z_task, None
            self._cur_latent_prior_mu = self.engine.read("eps", h.shape[0])
            return obs[..., : self.dim_state_body], z_task, state_cnt
        self._cur_task_encoder_mu, self._cur_task_encoder_logvar = h[:, :Z], h[:, Z:]
        z_task = self._reparameterize(h, eps)
        if self._latent_prior is not None:                          # rmt:801-809
            self._cur_latent_prior_mu = self.engine.net_forward(NET_PR, obs[..., : self.dim_state_body].contiguous())
            self._cur_latent_prior_logvar = torch.zeros_like(self._cur_latent_prior_mu)
        return obs[..., : self.dim_state_body], z_task, state_cnt

    def _reparameterize(self, mu_logvar, eps=None):
        self._rng_calls += 1
        return self.engine.reparam(mu_logvar, eps=eps, noise=self.latent_prior_noise,
                                   seed=self._rng_seed, offset=self._rng_calls)

    def forward_decoder(self, z_body, z_task, state=None, seq_lens=None, state_cnt=0):
        if (self.__dict__.get("_srv_on") and z_body.device.type == "cpu" and z_task.device.type == "cpu" and z_body.dim() == 2
                and z_body.shape[0] == 1):
            # the "pass_through" rollout (envs/rllib_env_imitation.py:233-258: z drawn by the caller) served by the resident kernel
            a = self.engine.rollout_server_decode(torch.cat([z_body.float(), z_task.float()], dim=-1).numpy())
            return self._motor_decoder._model[-1](torch.from_numpy(a.copy())[None]), state_cnt
        z = torch.cat([z_body.to(self.engine.device), z_task.to(self.engine.device)], dim=-1)
        a_hat = self.engine.net_forward(NET_MD, z)
        mh = self._motor_decoder_helper
        if mh is not None:                                            # rmt:833-835
            if self._motor_decoder_inputs != ["body", "task"]:        # rmt:822-829: the helper's own weights are compact
                z = z[..., : self.dim_state_body] if self._motor_decoder_inputs == ["body"] else z[..., self.dim_state_body:]
                z = z.contiguous()
            if torch.is_grad_enabled():
                add = mh(z.float())
            else:
                layers = self.__dict__.get("_mh_layers")
                if layers is None:
                    layers = self.__dict__["_mh_layers"] = [(m._model[0].weight, m._model[0].bias) for m in mh._model]
                add = self.engine.mlp_forward(z, layers, act=self.__dict__["_mh_acts"], out_act="tanh")
            a_hat = a_hat + self._motor_decoder_helper_range * add
        return self._motor_decoder._model[-1](a_hat), state_cnt

    def forward_world(self, obs, logits):
        x = torch.cat([obs[..., : self.dim_state_body].to(self.engine.device),
                       logits[..., : self.dim_action].to(self.engine.device)], dim=-1)
        return self.engine.net_forward(NET_WM, x)

    def forward_value_branch(self, obs, state=None, seq_lens=None, state_cnt=0):
        """rmt:846-853.  Sampling (no autograd: RLlib's rollout workers) runs the three small Linear layers as one
        chain of GEMV launches on the parameters where they are (`pvae_mlp_forward`); with autograd on it is the
        plain torch module, so that a policy-gradient learner can train the branch as upstream."""
        obs = obs.to(self.engine.device).float()
        vb = self._value_branch
        if torch.is_grad_enabled() or self.engine.ctx is None or obs.dim() != 2:
            return vb(obs), state_cnt
        layers = self.__dict__.get("_vb_layers")          # (Parameter objects, looked up once: module attribute reads
        if layers is None:                                #  cost microseconds each)
            layers = self.__dict__["_vb_layers"] = [(m._model[0].weight, m._model[0].bias) for m in vb._model]
        return self.engine.mlp_forward(obs, layers, act=self._vb_act), state_cnt

    def value_function(self):
        if self._cur_value is None and self._lazy is not None:
            val, _ = self.forward_value_branch(self._lazy[0])      # deferred by the fused forward
            self._cur_value = val.squeeze(1)
        assert self._cur_value is not None, "must call forward() first"
        return self._cur_value

    def set_exploration_std(self, std):
        self._motor_decoder._model[-1].set_val(float(np.log(std)))

    def task_encoder_variable(self):
        return self._cur_task_encoder_variable

    def body_encoder_variable(self):
        return self._cur_body_encoder_variable

    # -- weights I/O (rmt:870-928) ----------------------------------------------------------
    def portable_state_dict(self):
        return _portable(self.state_dict())

    def save_weights(self, file):
        torch.save(self.portable_state_dict(), file)

    def load_weights(self, file):
        self.load_state_dict(torch.load(file, map_location="cpu"))
        self.eval()

    def save_weights_task_encoder(self, file):
        torch.save({"task_encoder": _portable(self._task_encoder.state_dict())}, file)

    def load_weights_task_encoder(self, file):
        self._task_encoder.load_state_dict(torch.load(file, map_location="cpu")["task_encoder"])
        self._task_encoder.eval()
        self.reload_rollout_server()

    def save_weights_motor_decoder(self, file):
        torch.save(_portable(self._motor_decoder.state_dict()), file)

    def load_weights_motor_decoder(self, file):
        loaded = torch.load(file, map_location="cpu")
        current = self._motor_decoder.state_dict()
        for key in list(loaded.keys()):          # keep our log_std (rmt:895-905)
            if "log_std" in key:
                loaded[key] = current[key]
        self._motor_decoder.load_state_dict(loaded)
        self._motor_decoder.eval()
        self.reload_rollout_server()

    def save_weights_motor_decoder_helper(self, file):                # rmt:891-893
        if self._motor_decoder_helper is not None:
            torch.save(_portable(self._motor_decoder_helper.state_dict()), file)

    def load_weights_motor_decoder_helper(self, file):                # rmt:907-910
        if self._motor_decoder_helper is not None:
            self._motor_decoder_helper.load_state_dict(torch.load(file, map_location="cpu"))
            self._motor_decoder_helper.eval()

    def save_weights_world_model(self, file):
        torch.save(_portable(self._world_model.state_dict()), file)

    def load_weights_world_model(self, file):
        self._world_model.load_state_dict(torch.load(file, map_location="cpu"))
        self._world_model.eval()

    def save_weights_latent_prior(self, file):                       # rmt:921-923
        if self._latent_prior is not None:
            torch.save(_portable(self._latent_prior.state_dict()), file)

    def load_weights_latent_prior(self, file):                       # rmt:925-928
        if self._latent_prior is not None:
            self._latent_prior.load_state_dict(torch.load(file, map_location="cpu"))
            self._latent_prior.eval()

    # -- freezing (rmt:930-950) -----------------------------------------------------------
    def set_learnable_task_encoder(self, learnable):
        for p in self._task_encoder.parameters():
            p.requires_grad = learnable

    def set_learnable_motor_decoder(self, learnable, free_log_std=True):       # rmt:936-941
        for name, p in self._motor_decoder.named_parameters():
            p.requires_grad = free_log_std if "log_std" in name else learnable

    def set_learnable_motor_decoder_helper(self, learnable):           # rmt:942-946
        if self._motor_decoder_helper is not None:
            for p in self._motor_decoder_helper.parameters():
                p.requires_grad = learnable

    def set_learnable_world_model(self, learnable):
        for p in self._world_model.parameters():
            p.requires_grad = learnable

    def set_learnable_latent_prior(self, learnable):
        """Ours (rmt:930-950 has no switch for it): the learned prior mean only ever receives gradient
        through the KL term, i.e. together with the task encoder; the trainer flips both together."""
        if self._latent_prior is not None:
            for p in self._latent_prior.parameters():
                p.requires_grad = learnable

    def learnable_nets(self):
        def on(m):      # (a state_independent log_std follows `free_log_std`, not the stack: rmt:939-941)
            return m is not None and all(p.requires_grad for n, p in m.named_parameters() if "log_std" not in n)
        return [n for n, m in ((NET_TE, self._task_encoder), (NET_MD, self._motor_decoder),
                               (NET_WM, self._world_model), (NET_PR, self._latent_prior),
                               (NET_MH, self._motor_decoder_helper)) if on(m)]
