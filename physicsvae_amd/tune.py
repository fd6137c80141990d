"""A small stand-in for the slice of Ray Tune the reference uses (tpv:484-502): the
Trainable protocol (setup / step / save_checkpoint / load_checkpoint, train() bookkeeping)
and a `run` loop with periodic checkpoints that executes the trials of a `grid_search` config
one after the other (the cartesian product of the grid leaves, as Ray expands it; the reference's
`--vae_kl_coeff 0.1` means the two trials [1.0, 0.1]).  Actors, trial
schedulers and search algorithms are out of scope (control plane).

Under a multi-rank launch every rank runs the same trials (the data-parallel step needs all of
them); logs and checkpoints are written by rank 0 only and the other ranks wait for them."""
import itertools
import json
import os
import time


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:                                              # noqa: BLE001
        pass
    return 0, 1


def _barrier():
    rank, world = _rank_world()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def expand_grid(config):
    """All trial configs of `config`: one per element of the cartesian product of its
    `{"grid_search": [...]}` leaves (top level, as tpv:263-285 uses them), first key slowest."""
    keys = [k for k, v in config.items() if isinstance(v, dict) and set(v.keys()) == {"grid_search"}]
    if not keys:
        return [dict(config)]
    out = []
    for combo in itertools.product(*[config[k]["grid_search"] for k in keys]):
        c = dict(config)
        c.update(dict(zip(keys, combo)))
        out.append(c)
    return out


def grid_search(values):
    return {"grid_search": list(values)}


def resolve_grid(config):
    """A config with only single-valued grid_search leaves -> plain dict (tpv:263-285)."""
    out = {}
    for k, v in config.items():
        if isinstance(v, dict) and set(v.keys()) == {"grid_search"}:
            if len(v["grid_search"]) != 1:
                raise NotImplementedError(
                    "grid_search over %d values for %r: run one trial per value" % (len(v["grid_search"]), k))
            v = v["grid_search"][0]
        out[k] = v
    return out


class Trainable:
    def __init__(self, config=None, logdir=None):
        self.config = resolve_grid(config or {})
        self.logdir = logdir
        self._iteration = 0
        self._time_total = 0.0
        self.setup(self.config)

    @property
    def training_iteration(self):
        return self._iteration

    def setup(self, config):
        pass

    def step(self):
        raise NotImplementedError

    def train(self):
        t0 = time.time()
        result = dict(self.step())
        self._iteration += 1
        dt = time.time() - t0
        self._time_total += dt
        result.update(training_iteration=self._iteration, time_this_iter_s=dt,
                      time_total_s=self._time_total)
        return result

    def save_checkpoint(self, checkpoint_dir):
        raise NotImplementedError

    def load_checkpoint(self, checkpoint_path):
        raise NotImplementedError

    def save(self, checkpoint_dir=None):
        base = checkpoint_dir or os.path.join(self.logdir or ".", "checkpoint_%06d" % self._iteration)
        os.makedirs(base, exist_ok=True)
        return self.save_checkpoint(base)

    def restore(self, checkpoint_path):
        self.load_checkpoint(checkpoint_path)

    def stop(self):
        pass


class Trial:
    def __init__(self, logdir, config):
        self.logdir, self.config = logdir, config
        self.checkpoints, self.results = [], []


class Analysis:
    """What the reference reads back from `tune.run` (tpv:503-509): the best trial's logdir and
    its last checkpoint.  Single-trial attributes (`logdir`, `checkpoints`, `results`) refer to the
    last trial run."""

    def __init__(self, trials):
        self.trials = trials
        last = trials[-1]
        self.logdir, self.checkpoints, self.results = last.logdir, last.checkpoints, last.results

    def _best(self, metric, mode):
        if metric is None or len(self.trials) == 1:
            return self.trials[-1]
        scored = [(t.results[-1].get(metric), t) for t in self.trials if t.results and metric in t.results[-1]]
        if not scored:
            return self.trials[-1]
        pick = min if (mode or "min") == "min" else max
        return pick(scored, key=lambda kv: kv[0])[1]

    def get_best_logdir(self, metric=None, mode=None):
        return self._best(metric, mode).logdir

    def get_best_checkpoint(self, logdir=None, metric=None, mode=None):
        t = next((t for t in self.trials if t.logdir == logdir), None) or self._best(metric, mode)
        return t.checkpoints[-1] if t.checkpoints else None


def _latest_checkpoint(logdir):
    """(iteration, path of model.pth) of the newest checkpoint_<iter> directory under `logdir`, or None."""
    best = None
    if os.path.isdir(logdir):
        for d in os.listdir(logdir):
            path = os.path.join(logdir, d, "model.pth")
            if d.startswith("checkpoint_") and d[len("checkpoint_"):].isdigit() and os.path.exists(path):
                it = int(d[len("checkpoint_"):])
                if best is None or it > best[0]:
                    best = (it, path)
    return best


def _newest_run(exp_dir):
    """Trial directories of the NEWEST run of an experiment: `trial_<stamp>` or `trial_<stamp>_<index>`, grouped by
    stamp (an experiment directory accumulates one group per run, e.g. with the default --name)."""
    groups = {}
    for d in os.listdir(exp_dir):
        parts = d.split("_")
        if d.startswith("trial_") and len(parts) >= 3:
            groups.setdefault("_".join(parts[1:3]), []).append(d)
    return sorted(groups[max(groups)]) if groups else []


def run(trainable_cls, config=None, stop=None, checkpoint_freq=0, checkpoint_at_end=False,
        local_dir="~/ray_results", name=None, verbose=1, resume=False, **_ignored):
    """`resume=True` (tpv:500, `--resume`): continue the experiment `name` under `local_dir` -- trial i picks
    up the i-th trial directory of the experiment's NEWEST run at its newest checkpoint (weights through
    `restore`, the iteration counter from the directory name, as Ray does) and runs on to the stop criterion.
    As upstream (tm:215-216) that restores weights only: the trainer's epoch counter restarts at 0, so the
    world-model phase is replayed and Adam / the lr schedule start afresh -- unless the config carries
    "save_trainer_state" / "resume_trainer_state" (ours), which make the continuation bit-exact."""
    max_iter = int((stop or {}).get("training_iteration", 1))
    rank, world = _rank_world()
    stamp = time.strftime("%Y%m%d_%H%M%S")
    if world > 1:                                  # one directory name for the whole job
        import torch.distributed as dist
        box = [stamp]
        dist.broadcast_object_list(box, src=0)
        stamp = box[0]
    variants = expand_grid(config or {})
    trials = []
    exp_dir = os.path.join(os.path.expanduser(local_dir), name or trainable_cls.__name__)
    old = _newest_run(exp_dir) if (resume and os.path.isdir(exp_dir)) else []
    for ti, cfg in enumerate(variants):
        tag = "trial_%s" % stamp if len(variants) == 1 else "trial_%s_%02d" % (stamp, ti)
        if ti < len(old):
            tag = old[ti]
        logdir = os.path.join(exp_dir, tag)
        if rank == 0:
            os.makedirs(logdir, exist_ok=True)
        _barrier()
        trial = trainable_cls(cfg, logdir=logdir)
        rec = Trial(logdir, cfg)
        last = _latest_checkpoint(logdir) if ti < len(old) else None
        if last is not None:
            trial.restore(last[1])
            trial._iteration = max(trial._iteration, last[0])
            rec.checkpoints.append(last[1])
            if verbose and rank == 0:
                print("resumed %s at iteration %d" % (logdir, trial._iteration))
        log = open(os.path.join(logdir, "result.json"), "a" if last is not None else "w") if rank == 0 else None
        try:
            while trial.training_iteration < max_iter:
                res = trial.train()
                rec.results.append(res)
                if log:
                    log.write(json.dumps({k: v for k, v in res.items() if isinstance(v, (int, float, str))}) + "\n")
                    log.flush()
                if verbose and rank == 0:
                    print("iter %d  train %.6f  test %.6f  (%.2fs)" % (
                        res["training_iteration"], res.get("mean_train_loss", float("nan")),
                        res.get("mean_test_loss", float("nan")), res["time_this_iter_s"]))
                it = trial.training_iteration
                if (checkpoint_freq and it % checkpoint_freq == 0) or (checkpoint_at_end and it == max_iter):
                    base = os.path.join(logdir, "checkpoint_%06d" % it)
                    if world > 1 and getattr(trial, "config_flag", lambda k: False)("save_trainer_state"):
                        trial.gather_moments()     # collective: sharded exchanges keep moments per owner (torch_models.py)
                    if rank == 0:                  # replicas are bit-identical: one writer is enough
                        rec.checkpoints.append(trial.save(base))
                    else:
                        rec.checkpoints.append(os.path.join(base, "model.pth"))
                    _barrier()                     # the files exist before any rank may read them
        finally:
            if log:
                log.close()
        trial.stop()
        trials.append(rec)
    return Analysis(trials)
