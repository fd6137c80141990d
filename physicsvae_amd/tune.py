"""A small stand-in for the slice of Ray Tune the reference uses (tpv:484-502): the
Trainable protocol (setup / step / save_checkpoint / load_checkpoint, train() bookkeeping)
and a single-trial `run` loop with periodic checkpoints.  Trial scheduling, actors and
search are out of scope (control plane)."""
import json
import os
import time


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


class Analysis:
    def __init__(self, logdir, checkpoints, results):
        self.logdir, self.checkpoints, self.results = logdir, checkpoints, results

    def get_best_logdir(self, metric=None, mode=None):
        return self.logdir

    def get_best_checkpoint(self, logdir=None, metric=None, mode=None):
        return self.checkpoints[-1] if self.checkpoints else None


def run(trainable_cls, config=None, stop=None, checkpoint_freq=0, checkpoint_at_end=False,
        local_dir="~/ray_results", name=None, verbose=1, **_ignored):
    max_iter = int((stop or {}).get("training_iteration", 1))
    logdir = os.path.join(os.path.expanduser(local_dir), name or trainable_cls.__name__,
                          time.strftime("trial_%Y%m%d_%H%M%S"))
    os.makedirs(logdir, exist_ok=True)
    trial = trainable_cls(config, logdir=logdir)
    ckpts, results = [], []
    with open(os.path.join(logdir, "result.json"), "w") as log:
        while trial.training_iteration < max_iter:
            res = trial.train()
            results.append(res)
            log.write(json.dumps({k: v for k, v in res.items() if isinstance(v, (int, float, str))}) + "\n")
            log.flush()
            if verbose:
                print("iter %d  train %.6f  test %.6f  (%.2fs)" % (
                    res["training_iteration"], res.get("mean_train_loss", float("nan")),
                    res.get("mean_test_loss", float("nan")), res["time_this_iter_s"]))
            it = trial.training_iteration
            if (checkpoint_freq and it % checkpoint_freq == 0) or (checkpoint_at_end and it == max_iter):
                ckpts.append(trial.save())
    trial.stop()
    return Analysis(logdir, ckpts, results)
