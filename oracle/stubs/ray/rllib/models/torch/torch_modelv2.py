import torch.nn as nn

from ray.rllib.models.modelv2 import ModelV2


class TorchModelV2(ModelV2):
    def __init__(self, obs_space, action_space, num_outputs, model_config, name):
        assert isinstance(self, nn.Module)
        ModelV2.__init__(self, obs_space, action_space, num_outputs, model_config, name,
                         framework="torch")
