"""SlimFC / normc_initializer semantics as published for ray 1.11.0 (SURVEY.md
Appendix B): Linear -> initializer(weight) -> bias = 0 -> optional activation,
held in `self._model = nn.Sequential(...)`; normc = N(0,1) then every output row
rescaled to L2 norm `std`."""
import torch
import torch.nn as nn


def normc_initializer(std=1.0):
    def initializer(tensor):
        tensor.data.normal_(0, 1)
        tensor.data *= std / torch.sqrt(tensor.data.pow(2).sum(1, keepdim=True))
    return initializer


class SlimFC(nn.Module):
    def __init__(self, in_size, out_size, initializer=None, activation_fn=None,
                 use_bias=True, bias_init=0.0):
        super().__init__()
        layers = []
        linear = nn.Linear(in_size, out_size, bias=use_bias)
        if initializer is None:
            initializer = nn.init.xavier_uniform_
        initializer(linear.weight)
        if use_bias:
            nn.init.constant_(linear.bias, bias_init)
        layers.append(linear)
        if activation_fn is not None:
            layers.append(activation_fn())
        self._model = nn.Sequential(*layers)

    def forward(self, x):
        return self._model(x)
