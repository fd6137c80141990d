"""Minimal Trainable protocol: construct -> setup(config); train() -> step()."""


class Trainable:
    def __init__(self, config=None, logger_creator=None):
        self.config = config or {}
        self._iteration = 0
        self.setup(self.config)

    def setup(self, config):
        pass

    def step(self):
        raise NotImplementedError

    def train(self):
        result = self.step()
        self._iteration += 1
        result = dict(result)
        result["training_iteration"] = self._iteration
        return result

    def save_checkpoint(self, checkpoint_dir):
        raise NotImplementedError

    def load_checkpoint(self, checkpoint_path):
        raise NotImplementedError

    def restore(self, checkpoint_path):
        self.load_checkpoint(checkpoint_path)


def grid_search(values):
    return {"grid_search": list(values)}


def run(*args, **kwargs):
    raise RuntimeError("ray.tune.run is not available in the oracle stub")
