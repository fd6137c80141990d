"""HipEngine -- owns the device buffers the C-ABI library works on and wraps its calls.

PyTorch is used for what it is good at here: device memory, streams, `torch.save`.  All
arithmetic of the hot path happens in libpvae_gfx950.so (physicsvae_amd/csrc).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ACT_KINDS, FLAG_FUSED_ADAM, FLAG_NO_BACKWARD, NET_MD, NET_MH, NET_NAMES, NET_PR, NET_TE, NET_WM, NUM_NETS, PRIOR_KINDS

TENSOR_IDS = {"mu": 0, "logvar": 1, "z": 2, "a_hat": 3, "s2_hat": 4, "eps": 5, "prior_mu": 6}


class Stack:
    """Hidden layers of one FC stack as `FC.__init__` accepts them (rmt:234-270): a width and an activation per
    layer ("linear" / None: none).  `gen_layers(width, depth)` (tpv:180-192) is the special case one width, one
    activation; `Stack.of((width, depth), act)` builds that."""

    def __init__(self, widths, acts="relu"):
        self.widths = tuple(int(w) for w in widths)
        if isinstance(acts, str) or acts is None:
            acts = [acts] * len(self.widths)
        self.acts = tuple("linear" if a is None else a for a in acts)
        if not self.widths or len(self.widths) > _lib.MAX_HIDDEN or min(self.widths) < 1:
            raise NotImplementedError("a stack needs 1..%d hidden layers of positive width, got %s" % (_lib.MAX_HIDDEN, self.widths))
        if len(self.acts) != len(self.widths):
            raise ValueError("one activation per hidden layer: %s vs %s" % (self.acts, self.widths))
        for a in self.acts:                      # "swish"/"silu": ray's Swish module (learnable beta), not offered
            if a not in _lib.LAYER_ACTS:
                raise NotImplementedError("hidden activation %r: the HIP path offers %s" % (a, sorted(_lib.LAYER_ACTS)))

    @classmethod
    def of(cls, spec, act="relu"):
        if isinstance(spec, Stack):
            return spec
        width, depth = spec
        return cls([width] * int(depth), act)

    def uniform(self, act):
        return len(set(self.widths)) == 1 and set(self.acts) == {act}

    def key(self):
        return (self.widths, self.acts)

    # (width, depth) of the first layer: what the uniform fields of pvae_config carry
    def __getitem__(self, i):
        return (self.widths[0], len(self.widths))[i]

    def __iter__(self):
        return iter((self.widths[0], len(self.widths)))


class Arch:
    """Dims of the trainable stacks (tpv:247-286 keys, gen_layers tpv:180-192).  `te` / `md` / `wm` / `pr`:
    (width, depth) with the one hidden activation `act` (everything the trainer can generate), or a `Stack`
    (per-layer widths and activations, what custom_model_config's *_layers can describe, rmt:462-510)."""

    def __init__(self, dim_body, dim_action, latent, te, md, wm, prior="normal_zero_mean_one_std", pr=None,
                 act="relu", te_inputs=("body", "task"), md_inputs=("body", "task"), mh=None, mh_range=0.5):
        self.Db, self.Da, self.Z = int(dim_body), int(dim_action), int(latent)
        # task_encoder_inputs / motor_decoder_inputs (rmt:470, 485): column windows of the full-width first layers
        self.te_inputs, self.md_inputs = tuple(te_inputs), tuple(md_inputs)
        for name, v in (("task_encoder_inputs", self.te_inputs), ("motor_decoder_inputs", self.md_inputs)):
            if v not in _lib.INPUT_BITS:
                raise NotImplementedError("%s = %r: a non-empty subset of ['body', 'task'], in that order" % (name, list(v)))
        if prior not in PRIOR_KINDS:
            raise NotImplementedError("Unknown latent_prior_type:%s" % (prior,))      # rmt:624-625
        self.prior = prior                       # latent_prior_type (rmt:614-635; oracle/refpath.py PRIORS)
        if act not in ACT_KINDS:                 # "swish"/"silu": ray's Swish module (learnable beta), not offered
            raise NotImplementedError("hidden activation %r: the HIP path offers %s" % (act, sorted(ACT_KINDS)))
        self.act = act                           # the trainer's "act_fn" (tpv:262): the default of every stack
        self.te, self.md, self.wm = Stack.of(te, act), Stack.of(md, act), Stack.of(wm, act)
        self.pr = Stack.of(pr, act) if pr is not None else self.te      # learned prior stack
        # the motor decoder's helper (rmt:490-498, 670-680): hidden layers `mh` (None: no helper), tanh output x mh_range
        self.mh = Stack.of(mh, act) if mh is not None else None
        self.mh_range = float(mh_range)
        if self.mh is not None and not self.mh_range > 0:
            raise AssertionError("motor_decoder_helper_range must be positive (rmt:673)")

    @property
    def te_out(self):
        return self.Z if (self.prior == "hypersphere_uniform" or self.prior is False) else 2 * self.Z   # rmt:618-623

    def config(self, max_batch, lookahead=1):
        cfg = _lib.Config(self.Db, self.Da, self.Z, self.te[0], self.te[1], self.md[0],
                          self.md[1], self.wm[0], self.wm[1], int(max_batch), int(lookahead),
                          PRIOR_KINDS[self.prior], self.pr[0], self.pr[1], ACT_KINDS[self.act])
        if self.mh is not None:
            cfg.mh_width, cfg.mh_depth, cfg.mh_range = self.mh[0], self.mh[1], self.mh_range
        for net, st in ((NET_TE, self.te), (NET_MD, self.md), (NET_WM, self.wm), (NET_PR, self.pr), (NET_MH, self.mh)):
            if st is None or st.uniform(self.act):             # (a zeroed row: the uniform stack the scalar fields describe)
                continue
            for i, (w, a) in enumerate(zip(st.widths, st.acts)):
                cfg.layer_width[net][i] = w
                cfg.layer_act[net][i] = 1 + _lib.LAYER_ACTS[a]
        cfg.te_inputs, cfg.md_inputs = _lib.INPUT_BITS[self.te_inputs], _lib.INPUT_BITS[self.md_inputs]
        return cfg

    def key(self):
        return (self.Db, self.Da, self.Z, self.te.key(), self.md.key(), self.wm.key(), self.prior, self.pr.key(), self.act,
                self.te_inputs, self.md_inputs, self.mh.key() if self.mh is not None else None, self.mh_range)


class GraphedInfer:
    """`HipEngine.infer` captured once into a HIP graph.  The rollout forward at B = 1 is 8-10 tiny
    launches and host-bound (~33 us of Python + launch calls for ~15 us of GPU work); replaying the
    captured sequence is one host call.  Inputs and outputs live in static tensors: `__call__` copies
    the observation (and the draws, when the sampler is on) in and returns the static outputs
    (a_hat, s2_hat|None, z) -- valid until the next call.  The graph reads the parameter arena in
    place, so optimizer steps or load_state_dict between calls are seen by the next replay."""

    def __init__(self, engine, rows, want_s2=True, noise=False):
        engine._need_gpu()
        self.engine, self.rows, self.noise = engine, int(rows), bool(noise)
        dev = engine.device
        self.obs = torch.zeros(self.rows, 2 * engine.arch.Db, dtype=torch.float32, device=dev)
        self.eps = torch.zeros(self.rows, engine.arch.Z, dtype=torch.float32, device=dev) if noise else None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):                      # warm-up outside the capture (allocations, lazy init)
            self.out = engine.infer(self.obs, eps=self.eps, noise=self.noise, want_s2=want_s2)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side):
            self.out = engine.infer(self.obs, eps=self.eps, noise=self.noise, want_s2=want_s2, out=self.out)

    def __call__(self, obs, eps=None):
        # The captured launches write the staging panels that were current at capture time.  A
        # prefetched train step swaps the current and alternate panels, so after an odd number of them
        # those panels hold the NEXT training minibatch: tell the ctx that whatever is staged or
        # prefetched is gone (the next train step gathers again; pvae_infer does the same on the eager
        # path).  Host-side bookkeeping only, nothing is launched.
        self.engine.invalidate_staging()
        self.obs.copy_(obs.reshape(self.rows, -1), non_blocking=True)
        if self.noise:
            if eps is not None:
                self.eps.copy_(eps, non_blocking=True)
            else:
                self.eps.normal_()
        self.graph.replay()
        return self.out


def make_step_params(lr, adam_t=(1, 1, 1), a_rec=1.0, kl=1.0, s_rec=0.0, cyc=1e-3,
                     global_rows=0, seed=0, offset=0, beta1=0.9, beta2=0.999, eps=1e-8, loss="MSE",
                     weight_decay=0.0):
    sp = _lib.StepParams()
    sp.loss_kind = {"MSE": _lib.LOSS_MSE, "L1": _lib.LOSS_L1, "MAE": _lib.LOSS_L1}[loss]
    sp.a_rec_coeff, sp.kl_coeff, sp.s_rec_coeff, sp.cycle_coeff = a_rec, kl, s_rec, cyc
    sp.lr, sp.beta1, sp.beta2, sp.adam_eps = lr, beta1, beta2, eps
    for i in range(NUM_NETS):                      # (TE, MD, WM[, PR]); a missing entry counts as step 1
        sp.adam_t[i] = int(adam_t[i]) if i < len(adam_t) else 1
    sp.global_rows = int(global_rows)
    sp.rng_seed, sp.rng_offset = int(seed), int(offset)
    sp.weight_decay = float(weight_decay)
    return sp


class HipEngine:
    def __init__(self, arch, max_batch, device="cuda", lookahead=1):
        self.lib = _lib.load()
        self.arch = arch
        self.max_batch = int(max_batch)
        self.lookahead = int(lookahead)          # steps unrolled per sample (tpv:277, 367-428)
        # the library's <= 4-row rollout path: the LIBRARY's effective setting (it reads PVAE_ROLLOUT_FUSED once per
        # process), asked -- not re-read from the environment here, where the two could disagree
        self.fused_rollout = bool(self.lib.pvae_rollout_is_fused()) and arch.mh is None    # (helper models: staged path, pvae.hip infer_impl)
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            # an indexed device, so that comparisons with tensor.device (always indexed) are exact
            self.device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        self.cfg = arch.config(max_batch, self.lookahead)
        n = self.lib.pvae_arena_floats(C.byref(self.cfg))
        _lib.check(int(n), "pvae_arena_floats")
        self.arena_floats = int(n)
        self.layers = []
        info = _lib.LayerInfo()
        for i in range(_lib.check(self.lib.pvae_num_layers(C.byref(self.cfg)))):
            _lib.check(self.lib.pvae_layer(C.byref(self.cfg), i, C.byref(info)))
            self.layers.append({f: getattr(info, f) for f, _ in _lib.LayerInfo._fields_})
        self.segments = {}
        for net in (NET_TE, NET_MD, NET_WM, NET_PR, NET_MH):
            off, cnt = C.c_int64(), C.c_int64()
            _lib.check(self.lib.pvae_net_segment(C.byref(self.cfg), net, C.byref(off), C.byref(cnt)))
            self.segments[net] = (off.value, cnt.value)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.arena_floats, **f32)
        self.grads = torch.zeros(self.arena_floats, **f32)
        self.exp_avg = torch.zeros(self.arena_floats, **f32)
        self.exp_avg_sq = torch.zeros(self.arena_floats, **f32)
        self.ctx = None
        self.has_comm = False                    # RCCL communicator owned by the ctx (comm_init)
        self.has_p2p = False                     # peers' arenas mapped into this process (p2p_open)
        self._srv_io = None                      # buffers of the rollout server's calls (rollout_server_start)
        self.exchange_code = _lib.EXCHANGE_ALLREDUCE   # what comm_mode last selected (p2p_close falls back to all-reduce)
        self.workspace = None
        self.dataset = None
        self._loss_scratch = None
        if self.device.type == "cuda":
            self._create_ctx()

    # -- lifetime -------------------------------------------------------------------
    def _create_ctx(self):
        ctx = C.c_void_p()
        _lib.check(self.lib.pvae_create(C.byref(self.cfg), C.byref(ctx)), "pvae_create")
        self.ctx = ctx
        nbytes = self.lib.pvae_workspace_bytes(C.byref(self.cfg))
        self.workspace = torch.zeros(nbytes // 4 + 64, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pvae_bind_arenas(ctx, self.params.data_ptr(), self.grads.data_ptr(),
                                             self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()))
        _lib.check(self.lib.pvae_bind_workspace(ctx, self.workspace.data_ptr(), nbytes))
        _lib.apply_context_options(self.lib, ctx)          # (PVAE_PAIR, PVAE_DIRECT, ...: the library reads no environment)
        self._loss_scratch = torch.zeros(5, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.pvae_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def _need_gpu(self):
        if self.ctx is None:
            raise RuntimeError("HipEngine was created on %s: the HIP hot path needs a GPU "
                               "(there is no CPU fallback)" % self.device)

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- parameter views (checkpoint layout, rmt:234-283) -------------------------------
    def _view(self, arena, info, kind):
        if kind == "weight":
            blk = arena[info["w_offset"]: info["w_offset"] + info["n_out_pad"] * info["ld"]]
            return blk.view(info["n_out_pad"], info["ld"])[: info["n_out"], info["col0"]: info["col0"] + info["n_in"]]
        return arena[info["b_offset"]: info["b_offset"] + info["n_out"]]

    def named_views(self, arena=None):
        """{'<net>._model.<i>._model.0.weight|bias': strided view into the arena}."""
        arena = self.params if arena is None else arena
        out = {}
        for info in self.layers:
            base = "%s._model.%d._model.0." % (NET_NAMES[info["net"]], info["index"])
            out[base + "weight"] = self._view(arena, info, "weight")
            out[base + "bias"] = self._view(arena, info, "bias")
        return out

    def segment(self, arena, nets):
        """Contiguous slice of `arena` covering the given nets (arena order TE | MD | PR | WM: the stacks
        trained together in the joint phase are adjacent)."""
        offs = sorted(self.segments[n] for n in nets if self.segments[n][1] > 0)
        lo = offs[0][0]
        hi = offs[-1][0] + offs[-1][1]
        assert sum(c for _, c in offs) == hi - lo, "nets are not adjacent in the arena"
        return arena[lo:hi]

    # -- data ---------------------------------------------------------------------------
    def bind_dataset(self, states, actions, window_row, next_states=None, check=True):
        """`next_states` (optional, row-aligned with `states`): what the second half of x / the target s2
        is read from instead of the next state row (cond "rel", tpv:149-150).  `check=False`: `window_row` is a
        permutation of a table that was checked when it was bound (a shuffled epoch): no read-back, no host sync."""
        self._need_gpu()
        if self.dataset is not None and self.dataset[0] is states and self.dataset[1] is actions \
                and self.dataset[2] is window_row and self.dataset[3] is next_states:
            return                                # the very same device tensors are already bound
        states = states.to(self.device, torch.float32).contiguous()
        actions = actions.to(self.device, torch.float32).contiguous()
        window_row = window_row.to(self.device, torch.int32).contiguous()
        assert states.shape[1] == self.arch.Db and actions.shape[1] == self.arch.Da
        if check:
            assert int(window_row.max()) + self.lookahead < states.shape[0] + (1 if next_states is not None else 0)
        if next_states is not None:
            next_states = next_states.to(self.device, torch.float32).contiguous()
            assert next_states.shape == states.shape
        self.dataset = (states, actions, window_row, next_states)
        _lib.check(self.lib.pvae_bind_dataset(self.ctx, states.data_ptr(), actions.data_ptr(),
                                              window_row.data_ptr(), states.shape[0],
                                              window_row.shape[0]), "pvae_bind_dataset")
        if next_states is not None:
            _lib.check(self.lib.pvae_bind_dataset_next(self.ctx, next_states.data_ptr()), "pvae_bind_dataset_next")

    def set_direct(self, on=True):
        """Training steps read the demonstration set where it lies (no staging launch; include/pvae.h pvae_set_direct).
        Opt-in (default off: every step stages its input panels first -- same bits, and faster at 256 rows per GPU)."""
        self._need_gpu()
        _lib.check(self.lib.pvae_set_direct(self.ctx, 1 if on else 0), "pvae_set_direct")

    def direct_active(self, phase, rows, sp, fused=True):
        """Would a training step with these arguments take the direct path?"""
        self._need_gpu()
        return bool(_lib.check(int(self.lib.pvae_direct_active(self.ctx, phase, int(rows), C.byref(sp), 1 if fused else 0))))

    def invalidate_staging(self):
        """Forget the staged / prefetched minibatch (something else is about to overwrite the panels)."""
        if self.ctx is not None:
            _lib.check(self.lib.pvae_invalidate_staging(self.ctx), "pvae_invalidate_staging")

    def gather(self, first_window, rows):
        self._need_gpu()
        _lib.check(self.lib.pvae_gather(self.ctx, int(first_window), int(rows), self._stream()),
                   "pvae_gather")

    def set_batch(self, x, y):
        """x [B, L, 2Db] (L == 1 may be squeezed), y [B, L, Da] -- tpv:365-376."""
        self._need_gpu()
        x = x.reshape(x.shape[0], -1).to(self.device, torch.float32).contiguous()
        assert x.shape[1] == self.lookahead * 2 * self.arch.Db, "x must be [B, lookahead, 2*Db]"
        yp = None
        if y is not None:
            y = y.reshape(y.shape[0], -1).to(self.device, torch.float32).contiguous()
            assert y.shape[1] == self.lookahead * self.arch.Da, "y must be [B, lookahead, Da]"
            yp = y.data_ptr()
        self._keep = (x, y)
        _lib.check(self.lib.pvae_set_batch(self.ctx, x.data_ptr(), yp, x.shape[0], self._stream()),
                   "pvae_set_batch")
        return x.shape[0]

    # -- compute ------------------------------------------------------------------------
    def forward_backward(self, phase, rows, sp, eps=None, fused_adam=False, backward=True,
                         loss_out=None):
        self._need_gpu()
        flags = (FLAG_FUSED_ADAM if fused_adam else 0) | (0 if backward else FLAG_NO_BACKWARD)
        if eps is not None:
            eps = self._eps(eps, rows)
        out = self._loss_scratch if loss_out is None else loss_out
        _lib.check(self.lib.pvae_forward_backward(
            self.ctx, phase, int(rows), C.byref(sp), eps.data_ptr() if eps is not None else None,
            out.data_ptr(), flags, self._stream()), "pvae_forward_backward")
        return out

    def _eps(self, eps, rows):
        """[rows, Z] (lookahead 1) or [lookahead, rows, Z]: one slice per unrolled step."""
        eps = eps.to(self.device, torch.float32).contiguous()
        want = (self.lookahead, rows, self.arch.Z)
        assert tuple(eps.shape) == want or (self.lookahead == 1 and tuple(eps.shape) == want[1:]), \
            "eps must be [lookahead, rows, Z]"
        self._keep_eps = eps
        return eps

    def forward_seed(self, phase, rows, sp, eps=None):
        self._need_gpu()
        if eps is not None:
            eps = self._eps(eps, rows)
        _lib.check(self.lib.pvae_forward_seed(self.ctx, phase, int(rows), C.byref(sp),
                                              eps.data_ptr() if eps is not None else None, self._stream()),
                   "pvae_forward_seed")

    def backward_stage(self, phase, rows, sp, stage, loss_out=None):
        """Launch `stage` of the backward pass.  Returns (grad_slice | None, net, n_stages): the
        slice of the gradient arena that is final after this stage."""
        self._need_gpu()
        off, cnt, net, n = C.c_int64(), C.c_int64(), C.c_int(), C.c_int()
        _lib.check(self.lib.pvae_backward_stage(
            self.ctx, phase, int(rows), C.byref(sp), int(stage),
            loss_out.data_ptr() if loss_out is not None else None, self._stream(),
            C.byref(off), C.byref(cnt), C.byref(net), C.byref(n)), "pvae_backward_stage")
        seg = (off.value, cnt.value) if cnt.value else None
        return seg, net.value, n.value

    def backward_plan(self, phase, sp):
        """[(net, offset, count)] per stage of the backward pass, in stage order (count 0: the stage finishes no slice);
        nothing is launched.  Independent of the minibatch's rows (`pvae_backward_plan`)."""
        self._need_gpu()
        n = C.c_int()
        cap = 128
        off, cnt, net = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int * cap)()
        _lib.check(self.lib.pvae_backward_plan(self.ctx, phase, C.byref(sp), off, cnt, net, cap, C.byref(n)), "pvae_backward_plan")
        assert n.value <= cap
        return [(net[k], off[k], cnt[k]) for k in range(n.value)]

    def adam_segment(self, net, off, cnt, sp):
        self._need_gpu()
        _lib.check(self.lib.pvae_adam_segment(self.ctx, int(net), int(off), int(cnt), C.byref(sp), self._stream()),
                   "pvae_adam_segment")

    def adam(self, nets, sp):
        self._need_gpu()
        mask = 0
        for n in nets:
            mask |= 1 << n
        _lib.check(self.lib.pvae_adam(self.ctx, mask, C.byref(sp), self._stream()), "pvae_adam")

    def train_step(self, phase, first_window, rows, sp, eps=None, loss_out=None, next_span=None):
        """One fused optimizer step; `next_span` = (first_window, rows) of the minibatch that follows
        lets the library gather it inside this step's last launch (pvae_train_step_prefetch)."""
        self._need_gpu()
        if eps is not None:
            eps = self._eps(eps, rows)
        out = self._loss_scratch if loss_out is None else loss_out
        if next_span is not None:
            _lib.check(self.lib.pvae_train_step_prefetch(
                self.ctx, phase, int(first_window), int(rows), C.byref(sp),
                eps.data_ptr() if eps is not None else None, out.data_ptr(), int(next_span[0]), int(next_span[1]),
                self._stream()), "pvae_train_step_prefetch")
            return out
        _lib.check(self.lib.pvae_train_step(
            self.ctx, phase, int(first_window), int(rows), C.byref(sp),
            eps.data_ptr() if eps is not None else None, out.data_ptr(), self._stream()),
            "pvae_train_step")
        return out

    @property
    def in_library_exchange(self):
        """`dp_train_step` can run: the ctx has an RCCL communicator or its peers' arenas are mapped."""
        return self.has_comm or self.has_p2p

    # -- data-parallel exchange inside the library (RCCL) ----------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.pvae_comm_unique_id(buf), "pvae_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, world, unique_id):
        self._need_gpu()
        assert len(unique_id) == 128
        _lib.check(self.lib.pvae_comm_init(self.ctx, int(rank), int(world), C.c_char_p(unique_id)), "pvae_comm_init")
        self.has_comm = True
        self._dp_env()

    def _dp_env(self):
        """PVAE_DP_BUCKET_MB / PVAE_DP_SHARDED / PVAE_P2P_TIMEOUT_MS of the environment, applied through the ABI (the
        library reads no environment variable)."""
        if "PVAE_DP_BUCKET_MB" in os.environ:
            self.comm_config(float(os.environ["PVAE_DP_BUCKET_MB"]))
        if self.has_comm and not self.has_p2p and "PVAE_DP_SHARDED" in os.environ:
            self.comm_mode(os.environ["PVAE_DP_SHARDED"][:1] == "1")
        if "PVAE_P2P_TIMEOUT_MS" in os.environ:
            _lib.check(self.lib.pvae_set_option(self.ctx, b"p2p_timeout_ms", int(os.environ["PVAE_P2P_TIMEOUT_MS"])))

    def comm_destroy(self):
        if self.ctx is not None and self.has_comm:
            _lib.check(self.lib.pvae_comm_destroy(self.ctx), "pvae_comm_destroy")
            self.has_comm = False

    def comm_info(self):
        """(rank, nranks) as the ctx's RCCL communicator reports them; (0, 0) without one."""
        r, n = C.c_int(), C.c_int()
        if self.ctx is None:
            return 0, 0
        _lib.check(self.lib.pvae_comm_info(self.ctx, C.byref(r), C.byref(n)), "pvae_comm_info")
        return r.value, n.value

    def comm_config(self, bucket_mb=0.0, test_delay_us=0):
        """Exchange settings of dp_train_step: bucket size in MiB (0, the default: one in-line
        reduction per stack; > 0: bucketed and overlapped on the library's exchange stream, see
        include/pvae.h) and, for ordering tests, a delay in front of every reduction."""
        _lib.check(self.lib.pvae_comm_config(self.ctx, int(bucket_mb * (1 << 20)), int(test_delay_us)), "pvae_comm_config")

    def comm_mode(self, mode):
        """Exchange form of dp_train_step (include/pvae.h): "allreduce" (False) = all-reduce + replicated Adam,
        "sharded" (True) = reduce-scatter -> Adam on the owned 1/N slice -> all-gather of the parameters, "p2p" /
        "p2p_push" = the same shape as ONE launch over peer-mapped arenas, no RCCL (needs p2p_open; pull: owners read
        their slice from the peers' arenas, push: every rank writes its contributions into the owners' staging)."""
        code = {False: _lib.EXCHANGE_ALLREDUCE, True: _lib.EXCHANGE_SHARDED, "allreduce": _lib.EXCHANGE_ALLREDUCE,
                "sharded": _lib.EXCHANGE_SHARDED, "p2p": _lib.EXCHANGE_P2P, "p2p_push": _lib.EXCHANGE_P2P_PUSH,
                "local": _lib.EXCHANGE_LOCAL}[mode]
        _lib.check(self.lib.pvae_comm_mode(self.ctx, code), "pvae_comm_mode")
        self.exchange_code = code

    @property
    def p2p_active(self):
        """The peer-mapped exchange is what dp_train_step runs (its peers are mapped AND one of its forms is selected)."""
        return self.has_p2p and self.exchange_code in (_lib.EXCHANGE_P2P, _lib.EXCHANGE_P2P_PUSH)

    # -- peer-mapped exchange (PVAE_EXCHANGE_P2P) -------------------------------------------------
    def p2p_export(self):
        """This rank's 256-byte description of its gradient / parameter arenas and flag block (IPC handles)."""
        self._need_gpu()
        buf = C.create_string_buffer(_lib.P2P_BLOB_BYTES)
        _lib.check(self.lib.pvae_p2p_export(self.ctx, buf), "pvae_p2p_export")
        return buf.raw

    def p2p_open(self, rank, world, blobs):
        """Map every peer's buffers (`blobs`: the exports of ranks 0 .. world-1 in rank order)."""
        self._need_gpu()
        assert len(blobs) == world and all(len(b) == _lib.P2P_BLOB_BYTES for b in blobs)
        if "PVAE_P2P_SELFTEST_FLAGS_ONLY" in os.environ:
            _lib.check(self.lib.pvae_set_option(self.ctx, b"p2p_selftest_flags_only", 1))
        _lib.check(self.lib.pvae_p2p_open(self.ctx, int(rank), int(world), C.c_char_p(b"".join(blobs))), "pvae_p2p_open")
        self.has_p2p = True
        self._dp_env()

    def p2p_exchange(self, net, off, cnt, sp):
        """One slice of the gradient arena through the peer-mapped exchange (sum over the ranks by the slice owners,
        Adam, updated parameters to every rank); every rank issues the same calls."""
        self._need_gpu()
        _lib.check(self.lib.pvae_p2p_exchange(self.ctx, int(net), int(off), int(cnt), C.byref(sp), self._stream()),
                   "pvae_p2p_exchange")

    def p2p_selftest(self):
        """Prove remote write, remote read and flag delivery between all mapped peers (every rank calls it)."""
        self._need_gpu()
        _lib.check(self.lib.pvae_p2p_selftest(self.ctx, self._stream()), "pvae_p2p_selftest")

    def p2p_clear_errors(self):
        """Zero the time-out count `p2p_status` reports (the caller has dealt with them: candidate dropped, snapshot restored)."""
        if self.ctx is not None:
            _lib.check(self.lib.pvae_p2p_clear_errors(self.ctx, self._stream()), "pvae_p2p_clear_errors")

    def p2p_close(self):
        if self.ctx is not None and self.has_p2p:
            _lib.check(self.lib.pvae_p2p_close(self.ctx), "pvae_p2p_close")
            self.has_p2p = False
            if self.exchange_code in (_lib.EXCHANGE_P2P, _lib.EXCHANGE_P2P_PUSH):
                self.exchange_code = _lib.EXCHANGE_ALLREDUCE          # (what the library falls back to)

    def p2p_status(self, sync=True):
        """(rank, world, waits that gave up).  world = 0: not open.  `sync`: read the time-out word (synchronises)."""
        r, n, t = C.c_int(), C.c_int(), C.c_uint32()
        if self.ctx is None:
            return 0, 0, 0
        _lib.check(self.lib.pvae_p2p_status(self.ctx, C.byref(r), C.byref(n), C.byref(t) if sync else None,
                                            self._stream()), "pvae_p2p_status")
        return r.value, n.value, t.value

    def owned_slices(self, phase, net):
        """[(offset, count, replicated)]: the arena ranges of stack `net` (trained in `phase`) for which this rank holds
        valid Adam moments under the current exchange mode (include/pvae.h pvae_owned_slices)."""
        n, cap = C.c_int32(), 64
        off, cnt, rep = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_int32 * cap)()
        _lib.check(self.lib.pvae_owned_slices(self.ctx, int(phase), int(net), off, cnt, rep, cap, C.byref(n)), "pvae_owned_slices")
        return [(off[i], cnt[i], bool(rep[i])) for i in range(n.value)]

    def allreduce_grads(self, off, cnt):
        self._need_gpu()
        _lib.check(self.lib.pvae_allreduce_grads(self.ctx, int(off), int(cnt), self._stream()), "pvae_allreduce_grads")

    def dp_train_step(self, phase, first_window, rows, sp, eps=None, loss_out=None, next_span=None):
        self._need_gpu()
        nf, nr = (int(next_span[0]), int(next_span[1])) if next_span is not None else (0, 0)
        if eps is not None and rows:
            eps = self._eps(eps, rows)
        out = self._loss_scratch if loss_out is None else loss_out
        _lib.check(self.lib.pvae_dp_train_step(
            self.ctx, phase, int(first_window), int(rows), C.byref(sp),
            eps.data_ptr() if (eps is not None and rows) else None, out.data_ptr(), nf, nr, self._stream()),
            "pvae_dp_train_step")
        return out

    def read(self, name, rows, step=0):
        """Forward intermediate of time step `step` of the last batch."""
        self._need_gpu()
        width = {"mu": self.arch.Z, "logvar": self.arch.Z, "z": self.arch.Z, "eps": self.arch.Z,
                 "prior_mu": self.arch.Z, "a_hat": self.arch.Da, "s2_hat": self.arch.Db}[name]
        dst = torch.empty(rows, width, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pvae_read_tensor(self.ctx, TENSOR_IDS[name] + 8 * int(step), dst.data_ptr(), rows,
                                             self._stream()), "pvae_read_tensor")
        return dst

    def infer(self, obs, eps=None, noise=True, seed=0, offset=0, want_s2=True, out=None):
        """rmt:742-771 forward at rollout batch sizes: returns (a_hat, s2_hat|None, z).
        `out` = a tuple returned by an earlier call with the same row count: its tensors are
        overwritten instead of allocating new ones (the 30 Hz control loop calls this every step; at
        B = 1 the host side of the call costs as much as the GPU side)."""
        self._need_gpu()
        if not (obs.dtype == torch.float32 and obs.device == self.device and obs.dim() == 2 and obs.is_contiguous()):
            obs = obs.reshape(obs.shape[0], -1).to(self.device, torch.float32).contiguous()
        rows = obs.shape[0]
        if out is not None and out[0].shape[0] == rows and (out[1] is not None) == bool(want_s2):
            a_hat, s2, z = out
        else:
            a_hat = torch.empty(rows, self.arch.Da, dtype=torch.float32, device=self.device)
            s2 = torch.empty(rows, self.arch.Db, dtype=torch.float32, device=self.device) if want_s2 else None
            z = torch.empty(rows, self.arch.Z, dtype=torch.float32, device=self.device)
        if eps is not None:
            eps = eps.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.pvae_infer(
            self.ctx, obs.data_ptr(), rows, eps.data_ptr() if eps is not None else None,
            1 if noise else 0, int(seed), int(offset), a_hat.data_ptr(),
            s2.data_ptr() if s2 is not None else None, z.data_ptr(), self._stream()), "pvae_infer")
        return a_hat, s2, z

    def infer_host(self, obs, noise=True, seed=0, offset=0, log_std=None, timeout_s=0.05):
        """The control loop's call (rmt:742-771 at 1-4 rows; callers envs/rllib_env_imitation.py:215-266) with the
        observation taken from, and the action delivered to, HOST memory by the kernels themselves: `obs` is a CPU
        tensor / array [rows, 2 Db] (it is placed in a pinned staging buffer that the first encoder launch reads over
        PCIe), and the decoder's last launch stores the action straight into a pinned result buffer.  No copy
        launches either way and no stream synchronisation: the result buffer is pre-filled with NaN and the host
        waits until every entry has been overwritten (each 4-byte store arrives whole; Philox draws only -- supplied
        draws would need a copy of their own).  Returns a CPU tensor [rows, Da] ([rows, 2 Da] = [a_hat | log_std]
        with `log_std`, a device tensor [Da]) that stays valid until the next call.  If an entry is still missing
        after `timeout_s` (e.g. the model itself produced a NaN) the call falls back to a stream synchronisation."""
        import time as _time
        import numpy as _np
        self._need_gpu()
        Da, Db = self.arch.Da, self.arch.Db
        if isinstance(obs, torch.Tensor):
            obs = obs.detach().cpu().numpy()
        obs = _np.asarray(obs, dtype=_np.float32).reshape(-1, 2 * Db)
        rows = obs.shape[0]
        if not (1 <= rows <= 4 and self.fused_rollout):
            raise ValueError("infer_host serves the control loop's 1-4 rows (the library's rollout path)")
        h = getattr(self, "_host_io", None)
        if h is None:
            h_obs = torch.empty(4, 2 * Db, dtype=torch.float32).pin_memory()
            h_out = torch.empty(4, 2 * Da, dtype=torch.float32).pin_memory()
            h = self._host_io = (h_obs, h_out, h_obs.numpy(), h_out.numpy())
        h_obs, h_out, n_obs, n_out = h
        n_obs[:rows] = obs
        width = 2 * Da if log_std is not None else Da
        res = n_out[:rows, :width]
        # "not written yet" = one particular quiet-NaN bit pattern no arithmetic produces (a NaN the MODEL computes is the
        # canonical 0x7fc00000 / 0xffc00000 and counts as delivered): compared as integers, never with isnan
        bits = res.view(_np.uint32)
        bits.fill(0x7FC0DEAD)
        if log_std is not None:
            _lib.check(self.lib.pvae_infer_logits(self.ctx, h_obs.data_ptr(), rows, None, 1 if noise else 0, int(seed),
                                                  int(offset), h_out.data_ptr(), 2 * Da, log_std.data_ptr(), None, None,
                                                  self._stream()), "pvae_infer_logits")
        else:
            _lib.check(self.lib.pvae_infer_logits(self.ctx, h_obs.data_ptr(), rows, None, 1 if noise else 0, int(seed),
                                                  int(offset), h_out.data_ptr(), 2 * Da, None, None, None,
                                                  self._stream()), "pvae_infer_logits")
        t0 = _time.perf_counter()
        while (bits == 0x7FC0DEAD).any():
            if _time.perf_counter() - t0 > timeout_s:
                torch.cuda.current_stream(self.device).synchronize()
                break
        return h_out[:rows, :width]

    # -- call-persistent rollout server (include/pvae.h pvae_rollout_server_*) ---------------------------------
    def rollout_server_start(self, idle_ms=100.0, lifetime_s=600.0, scope="auto"):
        """Launch the resident rollout kernel (encoder + decoder weights in LDS, request block written by the host).
        `scope`: "xcd" = 32 workgroups on one XCD (the rest of the chip stays free), "chip" = 256 workgroups over all CUs
        (stacks too big for one XCD, e.g. 4x1024: kernels that need more LDS than is left wait for the server to leave),
        "auto" = one XCD when the stacks fit.  Raises RuntimeError when nothing fits: keep using `infer` / `infer_host`.
        While it is resident, device-wide synchronisations wait for it (at most `idle_ms` after the last request)."""
        self._need_gpu()
        env = os.environ.get("PVAE_SERVER_SCOPE", "")[:1]         # (A/B: overrides the caller's choice, as the C getenv did)
        scope = {"x": "xcd", "c": "chip"}.get(env, scope)
        mbx = os.environ.get("PVAE_SERVER_MAILBOX", "")[:1]
        _lib.check(self.lib.pvae_set_option(self.ctx, b"server_mailbox", {"h": 1, "d": 2}.get(mbx, 0)), "pvae_set_option")
        _lib.check(self.lib.pvae_rollout_server_start(self.ctx, float(idle_ms), float(lifetime_s),
                                                      {"auto": 0, "xcd": 1, "chip": 2}[scope]), "pvae_rollout_server_start")
        if self._srv_io is None:
            import numpy as _np
            Da, Db, Z = self.arch.Da, self.arch.Db, self.arch.Z
            bufs = (_np.zeros(2 * Db, _np.float32), _np.zeros(Da, _np.float32), _np.zeros(2 * Z, _np.float32), _np.zeros(Z, _np.float32))
            self._srv_io = bufs + tuple(C.c_void_p(b.ctypes.data) for b in bufs)       # (argument objects built once)
            self._srv_fn = self.lib.pvae_rollout_server_infer

    def rollout_server_infer(self, obs, noise=True, seed=0, offset=0, reload=False, timeout_ms=1000.0):
        """obs [2 Db] (CPU array / tensor) -> (a_hat [Da], mu_logvar [2 Z], z [Z]) as numpy views that stay valid until
        the next call; the same values as `infer(obs[None], noise=noise, seed=seed, offset=offset)`, bit for bit.
        A plain host call: no launch, no stream operation.  `reload`: copy the weights from the arena into LDS first."""
        io = self._srv_io
        if io is None:
            raise RuntimeError("rollout server not started (rollout_server_start)")
        n_obs, n_a, n_ml, n_z, p_obs, p_a, p_ml, p_z = io
        if isinstance(obs, torch.Tensor):
            obs = obs.detach().cpu().numpy()
        n_obs[:] = obs.reshape(-1)
        rc = self._srv_fn(self.ctx, p_obs, 1 if noise else 0, seed, offset, 1 if reload else 0, p_a, p_ml, p_z, timeout_ms)
        if rc:
            _lib.check(rc, "pvae_rollout_server_infer")
        return n_a, n_ml, n_z

    def rollout_server_infer_rows(self, obs, noise=True, seed=0, offset=0, reload=False, timeout_ms=1000.0):
        """obs [rows, 2 Db], 1 <= rows <= 4 -> (a_hat [rows, Da], mu_logvar [rows, 2 Z], z [rows, Z]) as fresh numpy arrays:
        ONE request to the resident kernel (every weight fragment read from LDS feeds all rows); the same values as
        `infer(obs, noise=noise, seed=seed, offset=offset)`, bit for bit."""
        import numpy as _np
        if self._srv_io is None:
            raise RuntimeError("rollout server not started (rollout_server_start)")
        if isinstance(obs, torch.Tensor):
            obs = obs.detach().cpu().numpy()
        x = _np.ascontiguousarray(obs, dtype=_np.float32).reshape(-1, 2 * self.arch.Db)
        rows = x.shape[0]
        a = _np.empty((rows, self.arch.Da), _np.float32)
        ml = _np.empty((rows, 2 * self.arch.Z), _np.float32)
        z = _np.empty((rows, self.arch.Z), _np.float32)
        _lib.check(self.lib.pvae_rollout_server_infer_rows(self.ctx, x.ctypes.data, rows, 1 if noise else 0, int(seed), int(offset),
                                                           1 if reload else 0, a.ctypes.data, ml.ctypes.data, z.ctypes.data,
                                                           float(timeout_ms)), "pvae_rollout_server_infer_rows")
        return a, ml, z

    def params_changed(self, stream=None):
        """The parameter arena was written outside the library (load_state_dict, a torch optimizer): holders of a copy -- the
        rollout server's LDS -- refresh before their next answer.  Optimizer steps through the library count by themselves."""
        if self.ctx is not None:
            _lib.check(self.lib.pvae_params_changed(self.ctx, stream), "pvae_params_changed")

    def rollout_server_decode(self, s1_z, timeout_ms=1000.0):
        """forward_decoder at B = 1 through the resident kernel: s1_z = [s1 (Db) | z (Z)] (CPU array) -> a_hat [Da] (numpy view,
        valid until the next call); the same bits as `net_forward(NET_MD, s1_z[None])`."""
        import numpy as _np
        io = self._srv_io
        if io is None:
            raise RuntimeError("rollout server not started (rollout_server_start)")
        x = _np.ascontiguousarray(_np.asarray(s1_z, dtype=_np.float32).reshape(-1))
        assert x.size == self.arch.Db + self.arch.Z, "s1_z must hold [s1 (Db) | z (Z)]"
        rc = self.lib.pvae_rollout_server_decode(self.ctx, x.ctypes.data, io[5], float(timeout_ms))
        if rc:
            _lib.check(rc, "pvae_rollout_server_decode")
        return io[1]

    def rollout_server_selfbench(self, obs, n=1000, noise=True):
        """us per request of n back-to-back requests, timed on the host clock INSIDE the library call (no Python per request)."""
        import numpy as _np
        o = _np.ascontiguousarray(_np.asarray(obs, dtype=_np.float32).reshape(-1))
        us = _np.zeros(int(n), _np.float64)
        _lib.check(self.lib.pvae_rollout_server_selfbench(self.ctx, o.ctypes.data, 1 if noise else 0, int(n), us.ctypes.data),
                   "pvae_rollout_server_selfbench")
        return us

    def rollout_server_timeline(self):
        """Microseconds after workgroup 0 saw the LAST request: [0, layer 0 inputs in LDS, layer 0 outputs published, ...,
        completion word issued], and the shader clock during that request in MHz -> (stamps, mhz)."""
        import numpy as _np
        us, n = _np.zeros(64, _np.float64), C.c_int32()
        _lib.check(self.lib.pvae_rollout_server_timeline(self.ctx, us.ctypes.data, 64, C.byref(n)), "pvae_rollout_server_timeline")
        return us[: n.value - 1], float(us[n.value - 1])

    def rollout_server_stop(self):
        if self.ctx is not None:
            _lib.check(self.lib.pvae_rollout_server_stop(self.ctx), "pvae_rollout_server_stop")

    def rollout_server_mailbox(self):
        """Where the request block lives while the server runs: "device" (the host writes device memory through the BAR,
        the kernel polls local memory), "host" (pinned host memory the kernel pulls from), None (not serving)."""
        a = C.c_int32()
        if self.ctx is None:
            return None
        _lib.check(self.lib.pvae_rollout_server_status(self.ctx, C.byref(a), None, None), "pvae_rollout_server_status")
        return {0: None, 1: "host", 2: "device"}[a.value]

    def rollout_server_status(self):
        """(serving, requests served so far, LDS bytes per workgroup)."""
        a, b, c = C.c_int32(), C.c_uint32(), C.c_int32()
        if self.ctx is None:
            return False, 0, 0
        _lib.check(self.lib.pvae_rollout_server_status(self.ctx, C.byref(a), C.byref(b), C.byref(c)), "pvae_rollout_server_status")
        return bool(a.value), b.value, abs(c.value)

    def rollout_server_scope(self):
        """"xcd" / "chip": over how much of the GPU the resident kernel's workgroups are dealt (None: never planned)."""
        c = C.c_int32()
        if self.ctx is None:
            return None
        _lib.check(self.lib.pvae_rollout_server_status(self.ctx, None, None, C.byref(c)), "pvae_rollout_server_status")
        return None if c.value == 0 else ("xcd" if c.value > 0 else "chip")

    def infer_logits(self, obs, log_std, eps=None, noise=True, seed=0, offset=0, want_s2=True):
        """`infer` with the module's output layout: returns (logits [rows, 2 Da] = [a_hat | log_std], s2_hat|None, z)
        -- the decoder's log-std vector (device tensor [Da]) is appended by the launch that writes the action
        (AppendLogStd, rmt:160-206), so `PhysicsVAE.forward` needs no concatenation of its own."""
        self._need_gpu()
        if not (obs.dtype == torch.float32 and obs.device == self.device and obs.dim() == 2 and obs.is_contiguous()):
            obs = obs.reshape(obs.shape[0], -1).to(self.device, torch.float32).contiguous()
        rows, Da = obs.shape[0], self.arch.Da
        logits = torch.empty(rows, 2 * Da, dtype=torch.float32, device=self.device)
        s2 = torch.empty(rows, self.arch.Db, dtype=torch.float32, device=self.device) if want_s2 else None
        z = torch.empty(rows, self.arch.Z, dtype=torch.float32, device=self.device)
        if eps is not None:
            eps = eps.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.pvae_infer_logits(
            self.ctx, obs.data_ptr(), rows, eps.data_ptr() if eps is not None else None,
            1 if noise else 0, int(seed), int(offset), logits.data_ptr(), 2 * Da, log_std.data_ptr(),
            s2.data_ptr() if s2 is not None else None, z.data_ptr(), self._stream()), "pvae_infer_logits")
        return logits, s2, z

    def mlp_forward(self, x, layers, act="relu", out_act=None):
        """A stack of Linear layers on caller-owned dense weights: `layers` = [(weight [n_out, n_in], bias)], hidden
        activation `act` (one name, or one per hidden layer), output layer linear or `out_act` (`pvae_mlp_forward`; the
        rollout model's value branch, rmt:846-853, and the motor decoder's helper, rmt:833-835)."""
        self._need_gpu()
        x = x.reshape(x.shape[0], -1).to(self.device, torch.float32)
        if x.stride(-1) != 1:
            x = x.contiguous()
        n = len(layers)
        acts = [act] * (n - 1) if isinstance(act, str) else ["linear" if a is None else a for a in act]
        assert len(acts) == n - 1, "one activation per hidden layer"
        key = tuple((w.data_ptr(), b.data_ptr(), w.stride(0)) for w, b in layers) + (tuple(acts),)
        plans = self.__dict__.setdefault("_mlp_plans", {})
        plan = plans.get(key)
        if plan is None:                                # pointer tables are rebuilt only when a tensor moved
            Pf = C.c_void_p * n
            Ii = C.c_int32 * n
            for w, b in layers:
                assert w.dtype == torch.float32 and w.device == self.device and w.stride(1) == 1 and b.is_contiguous()
            plan = (key, Pf(*[w.data_ptr() for w, _ in layers]), Pf(*[b.data_ptr() for _, b in layers]),
                    Ii(*[w.shape[1] for w, _ in layers]), Ii(*[w.shape[0] for w, _ in layers]),
                    Ii(*[w.stride(0) for w, _ in layers]), max([w.shape[0] for w, _ in layers[:-1]] or [1]),
                    (C.c_int32 * max(n - 1, 1))(*[_lib.LAYER_ACTS[a] for a in acts]))
            if len(plans) > 8:
                plans.clear()
            plans[key] = plan
        rows = x.shape[0]
        scratch = torch.empty(2 * rows * plan[6], dtype=torch.float32, device=self.device)
        out = torch.empty(rows, layers[-1][0].shape[0], dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pvae_mlp_forward(x.data_ptr(), rows, x.stride(0), n, plan[1], plan[2], plan[3], plan[4], plan[5],
                                             0 if out_act in (None, "linear") else (1 + _lib.LAYER_ACTS[out_act]) << 8,
                                             plan[7], scratch.data_ptr(), out.data_ptr(), out.shape[1],
                                             self._stream()), "pvae_mlp_forward")
        return out

    def graphed_infer(self, rows, want_s2=True, noise=False):
        """The rollout forward for a fixed row count as ONE replayable HIP graph (see GraphedInfer)."""
        return GraphedInfer(self, rows, want_s2=want_s2, noise=noise)

    def panel(self, kind, net=0, layer=0):
        """Workspace panel as a [Bp, width] view (inspection / tests).  kind: 'in', 'd_in',
        'act', 'dz', 's2', 'act_t', 'eps'."""
        kinds = {"in": 0, "d_in": 1, "act": 2, "dz": 3, "s2": 4, "act_t": 5, "eps": 6}
        off = _lib.check(int(self.lib.pvae_workspace_offset(C.byref(self.cfg), kinds[kind], net, layer)))
        bp = (self.max_batch + 31) // 32 * 32
        lays = [l for l in self.layers if l["net"] == net]
        pad64 = lambda v: (v + 63) // 64 * 64
        width = {"in": lays[0]["ld"], "d_in": lays[0]["ld"], "act": lays[layer]["n_out_pad"],
                 "dz": lays[layer]["n_out_pad"], "s2": pad64(self.arch.Db), "act_t": pad64(self.arch.Da),
                 "eps": self.arch.Z}[kind]
        return self.workspace[off: off + bp * width].view(bp, width)

    def kept_obs(self, rows):
        """The library's own copy of the observation rows of the last rollout call with <= 4 rows (workspace kind 7):
        a [rows, 2 Db] view that stays valid until the next such call, whatever the caller does with its buffer."""
        off = getattr(self, "_obs_keep_off", None)
        if off is None:
            off = self._obs_keep_off = _lib.check(int(self.lib.pvae_workspace_offset(C.byref(self.cfg), 7, 0, 0)))
        w = 2 * self.arch.Db
        return self.workspace[off: off + rows * w].view(rows, w)

    def net_forward(self, net, x):
        self._need_gpu()
        x = x.reshape(x.shape[0], -1).to(self.device, torch.float32).contiguous()
        n_out = [l for l in self.layers if l["net"] == net][-1]["n_out"]
        out = torch.empty(x.shape[0], n_out, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pvae_net_forward(self.ctx, net, x.data_ptr(), x.shape[0], out.data_ptr(),
                                             self._stream()), "pvae_net_forward")
        return out

    def reparam(self, mu_logvar, eps=None, noise=True, seed=0, offset=0):
        self._need_gpu()
        ml = mu_logvar.to(self.device, torch.float32).contiguous()
        z = torch.empty(ml.shape[0], self.arch.Z, dtype=torch.float32, device=self.device)
        if eps is not None:
            eps = eps.to(self.device, torch.float32).contiguous()
        _lib.check(self.lib.pvae_reparam(self.ctx, ml.data_ptr(), ml.shape[0],
                                         eps.data_ptr() if eps is not None else None,
                                         1 if noise else 0, int(seed), int(offset), z.data_ptr(),
                                         self._stream()), "pvae_reparam")
        return z


def gemm_probe(kind, a, b, c, bias_or_mask=None, relu=False, m=0, n=0, k=0):
    """Kernel-level entry (tests / roofline probes).  Tensors are dense fp32 on the GPU."""
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.pvae_gemm_probe(
        kind, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(), c.stride(0),
        bias_or_mask.data_ptr() if bias_or_mask is not None else None,
        bias_or_mask.stride(0) if (bias_or_mask is not None and bias_or_mask.dim() == 2) else 0,
        m, n, k, 1 if relu else 0, st), "pvae_gemm_probe")
    return c
