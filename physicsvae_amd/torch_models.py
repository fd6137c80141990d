"""Module API of the reference's torch_models.py, re-implemented for the HIP hot path.

Same names and call contracts (tm:21-216): `get_lr_scheduler`, `DatasetBase`, `get_loss_fn`,
`TrainModel` with the Trainable hooks `setup / step / save_checkpoint / load_checkpoint` and
the overridable `load_dataset / get_data_loader / prepare_data / create_model /
compute_model / compute_loss / compute_test_loss`.

Differences by design:
  * `step()` does not iterate a DataLoader: the demonstration set is resident in HBM and each
    minibatch is one `pvae_train_step` (gather -> forward/backward -> Adam, all HIP).  The
    per-batch `loss.item()` host sync of tm:144 is gone: losses land in a device array and
    are read once per epoch; the returned dict is the same.
  * the optimizer is `HipAdam` (fused into the weight-gradient kernels on one GPU, a flat
    multi-tensor kernel after the RCCL all-reduce on several); torch's lr schedulers drive it
    unchanged through `param_groups`.
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

from . import parallel, tune
from ._lib import NET_MD, NET_MH, NET_PR, NET_TE, NET_WM, PHASE_JOINT, PHASE_WORLD

EPSILON = np.finfo(np.float32).eps


def get_lr_scheduler(optimizer, name, params):
    """tm:21-37: 'cosine' | 'cosine_restart' | 'step' | anything else -> None."""
    table = {
        "cosine": lambda: optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=params["T_max"]),
        "cosine_restart": lambda: optim.lr_scheduler.CosineAnnealingWarmRestarts(
            optimizer, T_0=params["T_0"], T_mult=params["T_mult"]),
        "step": lambda: optim.lr_scheduler.StepLR(
            optimizer, step_size=params["step_size"], gamma=params["gamma"]),
    }
    return table[name]() if name in table else None


def get_loss_fn(loss):
    """tm:97-107.  Only MSE runs on the HIP path; the others are returned for API parity."""
    if loss == "MSE":
        return nn.MSELoss()
    if loss in ("MAE", "L1"):
        return nn.L1Loss()
    if loss == "CrossEntropy":
        return nn.CrossEntropyLoss()
    if loss == "NLLLoss":
        return nn.NLLLoss()
    raise NotImplementedError(loss)


class DatasetBase(torch.utils.data.Dataset):
    """tm:39-95: (X, Y) arrays with optional per-feature standardisation."""

    def __init__(self, X, Y, normalize_x=True, normalize_y=True):
        self.X, self.Y = X, Y
        self.normalize_x, self.normalize_y = normalize_x, normalize_y
        if normalize_x:
            self.X_mean, self.X_std = np.mean(X, axis=0), np.std(X, axis=0)
        if normalize_y:
            self.Y_mean, self.Y_std = np.mean(Y, axis=0), np.std(Y, axis=0)

    def __len__(self):
        return len(self.X)

    def __getitem__(self, index):
        return self.preprocess_x(self.X[index]), self.preprocess_y(self.Y[index])

    @staticmethod
    def _wrap(a, return_tensor):
        return torch.Tensor(a) if return_tensor else a

    def preprocess_x(self, x, return_tensor=True):
        if self.normalize_x:
            x = (x - self.X_mean) / (self.X_std + EPSILON)
        return self._wrap(x, return_tensor)

    def postprocess_x(self, x, return_tensor=True):
        if self.normalize_x:
            x = self.X_mean + np.multiply(x, self.X_std)
        return self._wrap(x, return_tensor)

    def preprocess_y(self, y, return_tensor=True):
        if self.normalize_y:
            y = (y - self.Y_mean) / (self.Y_std + EPSILON)
        return self._wrap(y, return_tensor)

    def postprocess_y(self, y, return_tensor=True):
        if self.normalize_y:
            y = self.Y_mean + np.multiply(y, self.Y_std)
        return self._wrap(y, return_tensor)


class WindowLoader:
    """Keep-last-partial minibatch schedule over a window dataset -- what `DataLoader(dataset, batch_size,
    shuffle=shuffle)` yields in the reference (tm:166-175).  Upstream's config never shuffles (the "suffle_data" typo of
    tpv:260 leaves `shuffle_data` None at tm:181), but the loader honours the key, and so does this one:

    * `shuffle` false: windows in order (SequentialSampler).
    * `shuffle` true: every pass over the loader draws a fresh permutation exactly as torch's loader does --
      `DataLoader.__iter__` takes ONE int64 from the default generator (`_base_seed`), `RandomSampler.__iter__`
      (generator=None) takes ONE more, seeds a private `torch.Generator` with it and yields
      `torch.randperm(n, generator=g)` -- so under the same `torch.manual_seed` the sample order equals the reference's
      (pinned by tests/golden/shuffle_tiny.npz).  The HIP path consumes the permutation as a permuted window -> row
      TABLE bound for the epoch (`epoch_table`): minibatches stay runs of consecutive table entries, so the gather, its
      prefetch and the data-parallel sharding (ranks slice the permuted order) are the sequential path's.

    `len()` = number of minibatches; iterating yields the same (x, y) float32 tensors as the reference's loader (host
    side, for inspection); `spans()` yields (first, rows) positions in the epoch's order for the HIP path."""

    def __init__(self, dataset, batch_size, shuffle=None):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), bool(shuffle)
        self.order = None               # the current pass's permutation (host int64), None when sequential

    def __len__(self):
        return math.ceil(len(self.dataset) / self.batch_size)

    def new_epoch(self):
        """Start a pass: draws the permutation of a shuffled loader (see the class comment for the draws) and returns it
        (None: sequential)."""
        if not self.shuffle:
            self.order = None
            return None
        torch.empty((), dtype=torch.int64).random_()                       # DataLoader iterator's _base_seed
        seed = int(torch.empty((), dtype=torch.int64).random_().item())    # RandomSampler's private generator
        g = torch.Generator()
        g.manual_seed(seed)
        self.order = torch.randperm(len(self.dataset), generator=g)
        return self.order

    def spans(self):
        n = len(self.dataset)
        for first in range(0, n, self.batch_size):
            yield first, min(self.batch_size, n - first)

    def __iter__(self):
        order = self.new_epoch()
        for first, rows in self.spans():
            idx = range(first, first + rows) if order is None else order[first: first + rows].tolist()
            xs, ys = zip(*(self.dataset[i] for i in idx))
            yield torch.stack(xs), torch.stack(ys)


class HipAdam(optim.Optimizer):
    """Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay) as tm:119-122 constructs it, with
    the update executed by the HIP kernels.  Holds what torch's schedulers and callers look
    at (`param_groups[0]['lr']`) plus the per-net 1-based step counters: state is created
    lazily per parameter in torch, so a net's counter starts when it first receives a
    gradient (TE/MD start at 1 at the phase switch; SURVEY.md section 7 hard part 7)."""

    def __init__(self, params, engine, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.engine = engine
        self.net_steps = {NET_TE: 0, NET_MD: 0, NET_WM: 0, NET_PR: 0, NET_MH: 0}

    @property
    def lr(self):
        return self.param_groups[0]["lr"]

    def next_counts(self, nets):
        """Advance and return adam_t for the nets updated by the coming step."""
        for n in nets:
            self.net_steps[n] += 1
        # (the helper's entry doubles as its switch: 0 = frozen for this step, include/pvae.h PVAE_NET_MH)
        return [max(self.net_steps[n], 1) for n in (NET_TE, NET_MD, NET_WM, NET_PR)] + \
               [max(self.net_steps[NET_MH], 1) if NET_MH in nets else 0]

    def step(self, closure=None):
        # the update itself is issued by TrainModel (fused or after the all-reduce)
        return None

    def zero_grad(self, set_to_none=True):
        return None

    def moments(self):
        """{key: (exp_avg view, exp_avg_sq view)} in checkpoint key naming."""
        m, v = self.engine.named_views(self.engine.exp_avg), self.engine.named_views(self.engine.exp_avg_sq)
        return {k: (m[k], v[k]) for k in m}


def dp_buckets(plan, limit=None):
    """Gradient buckets of a data-parallel step on the torch.distributed transport.  `plan`: [(net, offset, count)] per
    backward stage (HipEngine.backward_plan).  A bucket grows while the next finished slice belongs to the same stack and
    lies directly below it in the arena (layers finish last to first), and closes at a stack boundary or once it holds
    `limit` floats.  Returns [(net, lo, hi, last_stage)]: the bucket is final after stage `last_stage`."""
    out, cur = [], None
    for k, (net, off, cnt) in enumerate(plan):
        if not cnt:
            continue
        if cur is not None and (cur[0] != net or off + cnt != cur[1]):
            out.append(tuple(cur)); cur = None
        cur = [net, off, cur[2] if cur is not None else off + cnt, k]
        if limit is not None and cur[2] - cur[1] >= limit:
            out.append(tuple(cur)); cur = None
    if cur is not None:
        out.append(tuple(cur))
    return out


class TrainModel(tune.Trainable):
    """Epoch-per-step supervised trainer (tm:109-216)."""

    # -- Trainable hooks ------------------------------------------------------------------
    def setup(self, config):
        self.model = self.create_model(config)
        self.engine = self.model.engine
        self.device = self.engine.device
        self.dp = parallel.DataParallel.from_env()
        self.dp_bucket_mb = float(config.get("dp_bucket_mb", os.environ.get("PVAE_DP_BUCKET_MB", 0)))
        # How the gradient is exchanged (every rank must choose the same): `dp_exchange` / PVAE_DP_EXCHANGE =
        #   "inline"   RCCL all-reduce per stack on the compute stream + replicated Adam
        #   "bucketed" the same in 6 MiB buckets (or dp_bucket_mb) on the library's exchange stream, overlapped
        #   "sharded"  RCCL reduce-scatter -> Adam on the owned 1/N slice -> all-gather of the parameters
        #   "p2p"      the sharded shape as ONE launch per stack over peer-mapped arenas, no RCCL (include/pvae.h);
        #   "p2p_push" the same with remote writes only (contributions pushed into the slice owners' staging buffers)
        #   "default"  the library's own schedule, chosen by arithmetic
        #   "auto"     measured: the first training epoch of EACH phase times every available form on its first minibatch
        #              (state snapshotted and restored, parallel.DataParallel.autotune_exchange) and keeps the fastest
        #              whose replicas stay bit-identical.  What an unset key means with more than one rank (round 4;
        #              before: "default").
        self.dp_exchange = config.get("dp_exchange", os.environ.get("PVAE_DP_EXCHANGE")) or None
        if self.dp_exchange is None and self.dp.world > 1:
            self.dp_exchange = "auto"
        if self.dp_exchange == "default":
            self.dp_exchange = None
        if self.dp_exchange not in (None, "inline", "bucketed", "sharded", "p2p", "p2p_push", "auto"):
            raise ValueError("dp_exchange %r: expected default / inline / bucketed / sharded / p2p / p2p_push / auto" % (self.dp_exchange,))
        self.dp_exchange_report = None            # the calibration of the phase being trained (None: not run yet)
        self.dp_exchange_reports = {}             # {phase: report}
        self._replicas_checked = set()            # phases whose first epoch has been followed by the replica check
        if self.dp_exchange == "auto":
            self.dp.attach(self.engine)               # RCCL when the backend offers it; the peer-mapped forms join at calibration
        elif self.dp_exchange in ("p2p", "p2p_push"):
            if self.dp.collective and not self.dp.attach_p2p(self.engine, self.dp_exchange):
                raise RuntimeError("dp_exchange = %s: the peer-mapped exchange could not be set up (see stderr)" % self.dp_exchange)
        else:
            self.dp.attach(self.engine)
        if self.dp_exchange == "bucketed" and self.dp_bucket_mb <= 0:
            self.dp_bucket_mb = 6.0
        if self.engine.in_library_exchange and (self.dp_exchange in ("inline", "bucketed", "sharded") or
                                                "dp_bucket_mb" in config or "PVAE_DP_BUCKET_MB" in os.environ):
            self.engine.comm_config(self.dp_bucket_mb if self.dp_exchange != "inline" else 0.0)
        # sharded exchange (default off): reduce-scatter -> Adam on the owned 1/N slice -> all-gather of the
        # parameters (include/pvae.h PVAE_EXCHANGE_SHARDED); every rank must choose the same
        self.dp_sharded = bool(config.get("dp_sharded", os.environ.get("PVAE_DP_SHARDED", "0") == "1")) or \
            self.dp_exchange in ("sharded", "p2p", "p2p_push")
        if self.engine.has_comm and not self.engine.has_p2p:
            self.engine.comm_mode(self.dp_sharded)
        self.prefetch_gather = bool(config.get("prefetch_gather", os.environ.get("PVAE_PREFETCH", "1") != "0"))
        self.prepare_data(config)
        self.optimizer = HipAdam(self.model.parameters(), self.engine,
                                 lr=config.get("lr", 1e-3),
                                 weight_decay=config.get("weight_decay", 0.0))
        self.lr_scheduler = get_lr_scheduler(self.optimizer, config.get("lr_schedule", None),
                                             config.get("lr_schedule_params", None))
        self.loss_name = config.get("loss", "MSE")
        if self.loss_name not in ("MSE", "L1", "MAE"):
            raise NotImplementedError("loss %r: the HIP path implements MSE (the trainer's setting, "
                                      "tpv:257-258) and L1/MAE for the reconstruction terms" % self.loss_name)
        self.loss_fn = get_loss_fn(self.loss_name)
        self.loss_fn_test = get_loss_fn(config.get("loss_test", "MSE"))   # built, never used (tm:127, as upstream)
        self.iter = 0
        self.global_batch = 0                     # minibatches consumed so far (eps / Philox key)
        self.eps_fn = config.get("eps_fn")        # callable(forward_call, (rows, Z)) -> eps, or None
        self.rng_seed = int(config.get("seed", 0))
        self.last_loss_terms = None

    def step(self):
        self.iter += 1
        if not self.model.training:               # tm:133 calls it every epoch; the recursion over ~80 submodules costs
            self.model.train()                    # 0.4 ms, during which the GPU sits idle between two epochs
        losses = self.run_epoch(self.train_loader, train=True)
        mean_train = float(losses[:, 0].mean()) if len(losses) else 0.0
        self.last_loss_terms = losses.mean(dim=0).tolist() if len(losses) else None
        mean_test = 0.0
        if self.test_loader:
            test = self.run_epoch(self.test_loader, train=False)
            mean_test = float(test[:, 0].mean()) if len(test) else 0.0
        if self.lr_scheduler:
            self.lr_scheduler.step()              # once per EPOCH (tm:158-159)
        return {"mean_train_loss": mean_train, "mean_test_loss": mean_test}

    # -- the hot loop ---------------------------------------------------------------------
    def phase(self):
        """World phase <=> only the world model is learnable (tpv:326-329 / 347-350)."""
        nets = self.model.learnable_nets()
        # The motor decoder's helper (rmt:670-680) is never frozen by the trainer (tpv:326-329, 347-350 switch encoder,
        # decoder and world model only): in the world phase with lookahead 1 it receives no gradient (the loss does not
        # depend on a_hat) and torch's Adam skips it; with lookahead > 1 the world phase reaches it -- the state the world
        # model continues from is its own prediction under the helped action (tpv:417-421) -- and Adam's counter for it runs
        # from the first epoch; in the joint phase it trains with the decoder.
        helper = [NET_MH] if NET_MH in nets else []
        nets = [n for n in nets if n != NET_MH]
        if nets == [NET_WM]:
            return PHASE_WORLD, nets + (helper if self.engine.lookahead > 1 else [])
        if nets in ([NET_TE, NET_MD], [NET_TE, NET_MD, NET_PR]):     # (+ the learned prior mean, when configured)
            return PHASE_JOINT, nets + helper
        raise NotImplementedError("learnable nets %s: the trainer only uses {WM} or {TE, MD}" % nets)

    def step_params(self, nets, global_rows, train):
        raise NotImplementedError

    def run_epoch(self, loader, train):
        """One pass over `loader` (tm:137-144 / 147-156).  Returns a host tensor
        [n_minibatches, 5] = {total, a, kl, s, cyc} per minibatch (unweighted: the epoch
        figure is the plain mean of minibatch means, tm:145)."""
        eng, dp = self.engine, self.dp
        phase, nets = self.phase()
        arrays = loader.dataset.device_arrays(eng.device)
        order = loader.new_epoch() if hasattr(loader, "new_epoch") else None
        if order is None:
            eng.bind_dataset(*arrays)
        else:
            # a shuffled pass (tm:166-175 with shuffle_data): the window -> row table in this pass's order, bound for the
            # pass; positions [first, first + rows) of it are the minibatches.  Every rank slices rank 0's permutation.
            order = dp.broadcast(order, eng.device)
            eng.bind_dataset(arrays[0], arrays[1], arrays[2][order.to(eng.device)], *arrays[3:], check=False)
        n_glob = dp.global_steps(len(loader.dataset), loader.batch_size)
        if train and dp.collective and self.dp_exchange == "auto" and phase not in self.dp_exchange_reports:
            # (per phase: the world phase trains one stack and has nothing to overlap an exchange with, the joint phase
            #  two or three -- the form and bucketing that win need not be the same)
            self._autotune_exchange(loader, phase, nets)
        out = torch.zeros(max(n_glob, 1), 5, dtype=torch.float32, device=eng.device)
        for g in range(n_glob):
            first, rows, global_rows = dp.shard(g, len(loader.dataset), loader.batch_size)
            sp = self.step_params(nets, global_rows, train)
            sp.rng_seed = self.rng_seed
            sp.rng_offset = self.global_batch * 65536 + dp.rank * 64     # + t per unrolled step
            eps = None
            L = eng.lookahead
            if self.eps_fn is not None and (phase == PHASE_JOINT or L > 1):
                # one draw per model forward, i.e. per unrolled step (call = minibatch * L + t)
                lo = first - dp.global_first(g, loader.batch_size)
                eps = torch.stack([self.eps_fn(self.global_batch * L + t, (global_rows, eng.arch.Z))[lo: lo + rows]
                                   for t in range(L)])
            if not train:
                if rows:
                    eng.gather(first, rows)
                    eng.forward_backward(phase, rows, sp, eps=eps, backward=False, loss_out=out[g])
            elif not dp.collective:
                nxt = (first + rows, min(loader.batch_size, len(loader.dataset) - first - rows))
                if nxt[1] <= 0:                   # epoch boundary: the next epoch starts at window 0
                    nxt = (0, min(loader.batch_size, len(loader.dataset)))
                eng.train_step(phase, first, rows, sp, eps=eps, loss_out=out[g],
                               next_span=nxt if self.prefetch_gather else None)
            else:
                g2 = g + 1 if g + 1 < n_glob else 0           # this rank's shard of the following step
                nfirst, nrows, _ = dp.shard(g2, len(loader.dataset), loader.batch_size)
                self.dp_step(phase, nets, first, rows, sp, eps, out[g], next_span=(nfirst, nrows))
            if train and g == 0:
                # bookkeeping only (torch's schedulers want an optimizer.step() before their own); once per
                # epoch: the hooks and profiler ranges torch wraps around Optimizer.step cost ~10 us of host
                # time per call, which small models cannot hide behind a 30-40 us step
                self.optimizer.step()
            self.global_batch += 1
        if dp.collective:
            dp.all_reduce(out)
        host = out[:n_glob].cpu()                 # the single host sync of the epoch
        if dp.collective and eng.p2p_active:
            # only while a peer-mapped form is what runs (a rejected calibration candidate leaves nothing behind:
            # autotune_exchange clears the word), and collectively: a wait that gave up on ONE rank aborted that rank's
            # part of the exchange, so every rank's parameters are suspect and every rank must stop
            bad = dp.p2p_timeouts(eng)
            if bad:
                raise RuntimeError("peer-mapped exchange: %d wait(s) for a peer gave up (time-out) on some rank; that "
                                   "rank's update was skipped and the replicas are no longer consistent" % bad)
        if train and dp.collective and dp.world > 1 and eng.in_library_exchange and phase not in self._replicas_checked:
            # once per phase, after its first epoch, whatever chose the exchange form (an explicit dp_exchange = p2p
            # never went through the calibration's check): two checksum all-reduces
            self._replicas_checked.add(phase)
            if not dp.replicas_identical(eng):
                raise RuntimeError("data-parallel replicas hold different parameters after the first epoch of phase %d "
                                   "(exchange %s)" % (phase, self.dp_exchange or "default"))
        return host

    def _autotune_exchange(self, loader, phase, nets):
        """`dp_exchange = "auto"`: time every exchange form this build and this machine offer on the first global
        minibatch (parameters and moments restored afterwards) and keep the fastest whose replicas stay bit-identical."""
        eng, dp = self.engine, self.dp
        if getattr(self, "dp_sharded", False) and eng.in_library_exchange:
            # the form about to be replaced left every rank with valid Adam moments for its own slices only; the next
            # form may own differently (or replicate): assemble them on every rank while the ownership is still known
            self.gather_moments()
        if not eng.in_library_exchange:
            dp.attach(eng)
        first, rows, global_rows = dp.shard(0, len(loader.dataset), loader.batch_size)
        scratch = torch.zeros(5, dtype=torch.float32, device=eng.device)

        def run(n):
            for _ in range(n):
                sp = self.step_params(nets, global_rows, True)
                sp.rng_seed, sp.rng_offset = self.rng_seed, dp.rank * 64
                eng.dp_train_step(phase, first, rows, sp, eps=None, loss_out=scratch, next_span=None)

        counts = dict(self.optimizer.net_steps)            # (step_params advances Adam's per-stack step counters)
        chosen, self.dp_exchange_report = dp.autotune_exchange(eng, run)
        self.dp_exchange_reports[phase] = self.dp_exchange_report
        self.optimizer.net_steps.clear()
        self.optimizer.net_steps.update(counts)
        self.dp_exchange_chosen = chosen
        if chosen is None:                         # nothing in-library qualified: the torch.distributed transport carries on
            # -- for real: the engine is left in whatever form the last rejected candidate set, so detach it (dp_step
            # routes on eng.in_library_exchange) and forget the previous phase's sharding
            self.dp_sharded = False
            if eng.has_p2p:
                eng.p2p_close()
            if eng.has_comm:
                eng.comm_destroy()
            if dp.rank == 0:
                print("[physicsvae_amd] dp_exchange auto: no in-library exchange form available (%s); using torch.distributed"
                      % self.dp_exchange_report, file=sys.stderr)
            return
        self.dp_sharded = chosen in ("sharded", "p2p", "p2p_push")
        if dp.rank == 0:
            print("[physicsvae_amd] dp_exchange auto -> %s  %s" % (chosen, {k: round(v["us_per_step"], 1) for k, v in
                                                                      self.dp_exchange_report.items() if "us_per_step" in v}))

    def dp_step(self, phase, nets, first, rows, sp, eps, loss_out, next_span=None):
        """One data-parallel optimizer step.  The backward pass is issued launch by launch; the
        slices of the gradient arena it finishes (one per layer, last layer first, adjacent in the
        arena) are merged into buckets, and each bucket is SUM-all-reduced asynchronously as soon as
        its last slice is final (RCCL runs on its own stream, ordered behind the launch that
        produced the data) while the remaining backward launches keep the GPU busy; Adam runs
        bucket by bucket once the reductions complete.

        Bucket size (`dp_bucket_mb`, env PVAE_DP_BUCKET_MB): a bucket closes at a net boundary or
        when it reaches that many MB.  Default = one bucket per net: a collective costs tens of
        microseconds of latency on xGMI and of host time in torch.distributed regardless of size,
        which at ~120 us per step outweighs what finer-grained overlap could hide (measured with
        one rank through RCCL: 5 collectives per step 206 us, 1 per step see DESIGN.md)."""
        eng, dp = self.engine, self.dp
        if eng.in_library_exchange:   # whole step inside the library (RCCL or peer-mapped; in line, or bucketed + overlapped)
            eng.dp_train_step(phase, first, rows, sp, eps=eps, loss_out=loss_out,
                              next_span=next_span if self.prefetch_gather else None)
            return
        # Buckets = runs of slices the backward pass finishes, merged while they stay inside one stack and adjacent in the
        # arena, closed at `dp_bucket_mb`: computed from the PLAN (pvae_backward_plan: no launch, independent of the
        # minibatch's rows), so that every rank -- also one whose shard of a ragged last global batch is empty and launches
        # nothing -- issues the same collectives, of the same sizes, in the same order.
        limit = int(self.dp_bucket_mb * (1 << 20) / 4) if self.dp_bucket_mb > 0 else None
        buckets = dp_buckets(eng.backward_plan(phase, sp), limit)      # [(net, lo, hi, last_stage)]
        if not rows:                              # empty shard: zero gradient through the same collectives, then Adam
            for net, lo, hi, _ in buckets:        # (all reductions first, then the updates: the order its peers use)
                eng.grads[lo:hi].zero_()
                dp.all_reduce(eng.grads[lo:hi])
            for net, lo, hi, _ in buckets:
                self._apply_update(net, lo, hi - lo, sp)
            return
        eng.gather(first, rows)
        eng.forward_seed(phase, rows, sp, eps=eps)
        pending, k, n, b = [], 0, 1, 0
        while k < n:
            _, _, n = eng.backward_stage(phase, rows, sp, k, loss_out=loss_out)
            while b < len(buckets) and buckets[b][3] == k:          # every slice of the bucket is final: reduce it
                net, lo, hi, _ = buckets[b]
                pending.append((net, lo, hi - lo, dp.all_reduce_async(eng.grads[lo:hi])))
                b += 1
            k += 1
        assert b == len(buckets), "backward plan and backward stages disagree"
        for net, off, cnt, work in pending:
            if work is not None:
                work.wait()
            self._apply_update(net, off, cnt, sp)

    def _apply_update(self, net, off, cnt, sp):
        """Adam on a reduced slice [off, off+cnt) of the gradient arena.  Sharded exchange over
        torch.distributed: this rank updates only its 1/N of the slice and the updated parameters are
        all-gathered (the gradient itself was all-reduced: this transport has no stream-ordered
        reduce-scatter on every backend; the in-library RCCL path does the true reduce-scatter)."""
        eng, dp = self.engine, self.dp
        if not (self.dp_sharded and dp.world > 1 and cnt % (dp.world * 4) == 0):
            eng.adam_segment(net, off, cnt, sp)
            return
        import torch.distributed as dist
        sl = cnt // dp.world
        mine = off + dp.rank * sl
        eng.adam_segment(net, mine, sl, sp)
        outs = [eng.params[off + r * sl: off + (r + 1) * sl] for r in range(dp.world)]
        dist.all_gather(outs, outs[dp.rank].clone(), group=dp.group)

    # -- overridables, reference names ----------------------------------------------------
    def load_dataset(self, file):
        raise NotImplementedError

    def get_data_loader(self, dataset, batch_size, shuffle):
        return WindowLoader(dataset, batch_size, shuffle)

    def prepare_data(self, config):
        train = self.load_dataset(config.get("dataset_train"))
        test = config.get("dataset_test")
        if test is not None:
            test = self.load_dataset(test)
        bs, shuffle = config.get("batch_size"), config.get("shuffle_data")
        self.train_loader = self.get_data_loader(train, bs, shuffle)
        self.test_loader = self.get_data_loader(test, bs, shuffle) if test is not None else None

    def create_model(self, config):
        return config.get("model")

    def compute_model(self, x):                   # tm:198-199
        return self.model(x)

    def compute_loss(self, y, x):                 # tm:201-203
        y_recon = self.compute_model(x)
        return self.loss_fn(y_recon, y)

    def compute_test_loss(self, y, x):            # tm:205-207
        y_recon = self.compute_model(x)
        return self.loss_fn(y_recon, y)

    def save_checkpoint(self, checkpoint_dir):
        print(checkpoint_dir)
        path = os.path.join(checkpoint_dir, "model.pth")
        torch.save(self.model.portable_state_dict(), path)
        if self.config_flag("save_trainer_state"):        # ours, opt-in: the directory otherwise holds exactly
            self.save_trainer_state(os.path.join(checkpoint_dir, "trainer_state.pt"))   # the reference's files
        return path

    def load_checkpoint(self, checkpoint_path):
        # weights only -- optimizer, scheduler, iter are not restored (tm:215-216, App. C-7), unless the
        # config asks for it with "resume_trainer_state" (ours) and the checkpoint directory has the file
        self.model.load_state_dict(torch.load(checkpoint_path, map_location="cpu"))
        extra = os.path.join(os.path.dirname(checkpoint_path), "trainer_state.pt")
        if self.config_flag("resume_trainer_state") and os.path.exists(extra):
            self.load_trainer_state(extra)

    def config_flag(self, name):
        return bool(getattr(self, "config", {}).get(name, False))

    # -- ours: everything a bit-exact resume needs beyond the weights -----------------
    def gather_moments(self):
        """COLLECTIVE (every rank calls it).  Under the sharded and peer-mapped exchanges a rank keeps valid Adam
        moments only for the slices it owns; this assembles the full moments on every rank, in place: each rank
        zeroes a copy outside what it owns (replicated ranges count for rank 0), the copies are SUM-all-reduced
        (x + 0 + ... + 0 is exact) and written back.  After it `save_trainer_state` may run on any one rank."""
        eng, dp = self.engine, self.dp
        if not (dp.collective and dp.world > 1 and getattr(self, "dp_sharded", False) and eng.in_library_exchange):
            return False
        import torch.distributed as dist
        mask = torch.zeros_like(eng.exp_avg, dtype=torch.bool)
        for net, phase in ((NET_WM, PHASE_WORLD), (NET_TE, PHASE_JOINT), (NET_MD, PHASE_JOINT), (NET_PR, PHASE_JOINT),
                           (NET_MH, PHASE_JOINT)):
            for off, cnt, rep in eng.owned_slices(phase, net):
                if cnt > 0 and (not rep or dp.rank == 0):
                    mask[off: off + cnt] = True
        for t in (eng.exp_avg, eng.exp_avg_sq):
            full = torch.where(mask, t, torch.zeros_like(t))
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=dp.group)
            t.copy_(full)
        self._moments_gathered_at = (self.iter, self.global_batch)
        return True

    def save_trainer_state(self, path):
        """Adam moments (checkpoint key naming, CPU), per-stack Adam step counts, epoch counter, minibatch
        counter (keys the eps / Philox stream) and the lr scheduler.  Under the sharded / peer-mapped exchanges
        every rank holds valid moments for its own slices only: the file is written only if `gather_moments`
        (collective) has assembled them at this very point of the run."""
        if getattr(self, "dp_sharded", False) and self.dp.world > 1 and \
                getattr(self, "_moments_gathered_at", None) != (self.iter, self.global_batch):
            return None
        mom = self.optimizer.moments()
        state = {
            "exp_avg": {k: m.detach().cpu().contiguous().clone() for k, (m, _) in mom.items()},
            "exp_avg_sq": {k: v.detach().cpu().contiguous().clone() for k, (_, v) in mom.items()},
            "net_steps": dict(self.optimizer.net_steps), "iter": self.iter, "global_batch": self.global_batch,
            "training_iteration": self.training_iteration, "lr": self.optimizer.lr,
            "lr_scheduler": self.lr_scheduler.state_dict() if self.lr_scheduler else None,
            "learnable_nets": self.model.learnable_nets(),
        }
        torch.save(state, path)
        return path

    def load_trainer_state(self, path):
        state = torch.load(path, map_location="cpu")
        mom = self.optimizer.moments()
        with torch.no_grad():
            for k, (m, v) in mom.items():
                m.copy_(state["exp_avg"][k])
                v.copy_(state["exp_avg_sq"][k])
        self.optimizer.net_steps.update(state["net_steps"])
        self.iter, self.global_batch = int(state["iter"]), int(state["global_batch"])
        self._iteration = int(state.get("training_iteration", self._iteration))
        if self.lr_scheduler and state.get("lr_scheduler"):
            self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        self.optimizer.param_groups[0]["lr"] = state["lr"]
        return state
