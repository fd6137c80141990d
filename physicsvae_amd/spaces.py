"""Minimal Box space (shape/low/high/dtype holder).  The reference builds gym.spaces.Box
objects only to carry dimensions into the model constructor (tpv:216-233); gym is not a
dependency of this package."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low, dtype=dtype)
        self.high = np.asarray(high, dtype=dtype)
        self.shape = tuple(self.low.shape) if shape is None else tuple(shape)
        self.dtype = np.dtype(dtype)

    def __repr__(self):
        return "Box(shape=%s, dtype=%s)" % (self.shape, self.dtype)
